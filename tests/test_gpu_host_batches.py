"""Decoder calls on LLRs in host memory (POLAR_HOST): the batch crosses the link in chunks of 16384 frames through two
staging buffers, chunk i+1 on a second stream while chunk i is decoded.  Batches of several chunks, from pageable and from
page-locked memory, must give what the same frames give when they are resident on the GPU."""
import numpy as np
import pytest
import torch

import _native
import functions as pcf
from channel import Channel

pytestmark = pytest.mark.gpu

B = 3 * 16384 + 777  # four chunks, the last one short


def _frames(code, N, K, dtype, snr=1.5):
    tdt = torch.float32 if dtype is np.float32 else torch.float64
    dev = torch.empty((B, N), dtype=tdt, device="cuda")
    code.gen_frames_dev(None, dev.data_ptr(), dtype, Channel("BPSK", snr, "SNRb", K / N).noisePower, 11, 0, B)
    _native.default_ctx(0).sync()
    pinned = torch.empty((B, N), dtype=tdt, pin_memory=True)
    pinned.copy_(dev)
    return dev, pinned, pinned.numpy().copy()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_batches_of_several_chunks_equal_resident_batches(dtype):
    N, K = 128, 64
    ctx = _native.default_ctx(0)
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
    lib = _native.lib()
    dev, pinned, pageable = _frames(code, N, K, dtype)
    dt = _native._dt(dtype)

    def run(call):
        want = torch.zeros((B, code.Kw), dtype=torch.int32, device="cuda")
        _native._check(call(dev.data_ptr(), want.data_ptr(), _native.DEVICE))
        ctx.sync()
        want = want.cpu().numpy()
        assert want.any()
        for src in (pinned.data_ptr(), pageable.ctypes.data):
            for _ in range(2):  # the second call finds the staging buffers and events as the first left them
                got = np.zeros((B, code.Kw), dtype=np.int32)
                _native._check(call(src, got.ctypes.data, _native.HOST))
                assert np.array_equal(got, want)
        return want

    sc = run(lambda l, h, sp: lib.polar_sc_decode_packed(code._h, l, dt, h, None, None, B, _native.SC_FAST, sp))
    scl = run(lambda l, h, sp: lib.polar_scl_decode_packed(code._h, l, dt, 4, h, None, None, B, 0, sp))
    assert (sc != scl).any()  # at 1.5 dB the list decoder repairs some frames: the two legs did run different decoders

    # byte outputs with the decision LLRs coming back as well (a second large copy per chunk on the decode stream)
    info, _, dl = code.sc_decode(pageable, want_llrs=True)
    info_d = np.zeros((B, K), dtype=np.uint8)
    dl_d = torch.zeros((B, N), dtype=dev.dtype, device="cuda")
    hat = torch.zeros((B, code.Kw), dtype=torch.int32, device="cuda")
    code.sc_decode_dev(dev.data_ptr(), dtype, hat.data_ptr(), B, fast=False, dec_llr_ptr=dl_d.data_ptr())
    ctx.sync()
    words = hat.cpu().numpy().view(np.uint32)
    info_d = ((words[:, np.arange(K) >> 5] >> (np.arange(K) & 31).astype(np.uint32)) & 1).astype(np.uint8)
    assert np.array_equal(info, info_d) and np.array_equal(dl, dl_d.cpu().numpy())


def test_host_flip_search_of_several_chunks_equals_resident_batch():
    N, K, crc = 128, 72, 0x107
    ctx = _native.default_ctx(0)
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(ctx, mask)
    code.set_crc(crc)
    code.set_reliability(pcf.GA(N, K, 2.0))
    dev, pinned, pageable = _frames(code, N, K, np.float32, snr=2.5)
    hat = torch.zeros((B, code.Kw), dtype=torch.int32, device="cuda")
    runs = torch.zeros(B, dtype=torch.int32, device="cuda")
    code.pbf_decode_dev(dev.data_ptr(), np.float32, 1, hat.data_ptr(), B, runs_ptr=runs.data_ptr())
    ctx.sync()
    info, r, ok, last = code.pbf_decode(pageable, 1)
    words = hat.cpu().numpy().view(np.uint32)
    want = ((words[:, np.arange(K) >> 5] >> (np.arange(K) & 31).astype(np.uint32)) & 1).astype(np.uint8)
    assert np.array_equal(info, want) and np.array_equal(r, runs.cpu().numpy()) and r.max() > 1


def test_host_calls_interleaved_with_the_monte_carlo_pipeline(monkeypatch):
    """Host-space decoder calls and the staged Monte-Carlo loop share the context's second stream, its events and the two
    LLR staging buffers: interleaved in any order, on the context's own stream or on one the caller set, each gives what it
    gives alone."""
    N, K = 256, 128
    ctx = _native.default_ctx(0)
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
    lib = _native.lib()
    dev, pinned, pageable = _frames(code, N, K, np.float32)
    n0 = Channel("BPSK", 1.5, "SNRb", K / N).noisePower
    monkeypatch.setenv("POLAR_MC_KERNEL", "staged")
    monkeypatch.setenv("POLAR_MC_CHUNK", "20000")  # several chunks: the loop's two-buffer path
    dt = _native._dt(np.float32)

    def host_scl(src):
        got = np.zeros((B, code.Kw), dtype=np.int32)
        _native._check(lib.polar_scl_decode_packed(code._h, src, dt, 4, got.ctypes.data, None, None, B, 0, _native.HOST))
        return got

    def host_sc(src):
        got = np.zeros((B, code.Kw), dtype=np.int32)
        _native._check(lib.polar_sc_decode_packed(code._h, src, dt, got.ctypes.data, None, None, B, _native.SC_FAST, _native.HOST))
        return got

    want_scl, want_sc = host_scl(pinned.data_ptr()), host_sc(pageable.ctypes.data)
    want_mc = code.mc_run(_native.DEC_SCL, 4, np.float32, n0, 3, 0, 70000).tolist()
    side = torch.cuda.Stream()
    for stream in (None, side):
        ctx.set_stream(stream.cuda_stream if stream is not None else None)
        try:
            for _ in range(2):
                assert np.array_equal(host_scl(pageable.ctypes.data), want_scl)
                assert code.mc_run(_native.DEC_SCL, 4, np.float32, n0, 3, 0, 70000).tolist() == want_mc
                assert np.array_equal(host_sc(pinned.data_ptr()), want_sc)
                assert code.mc_run(_native.DEC_SC, 1, np.float32, n0, 3, 0, 70000)[0] == 70000
                assert np.array_equal(host_scl(pinned.data_ptr()), want_scl)
        finally:
            ctx.set_stream(None)
