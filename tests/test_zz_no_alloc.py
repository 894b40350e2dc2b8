"""The entry point that allocates nothing (polar_pbf_decode_noalloc) against polar_pbf_decode_packed: same decoded
words, SC-run counts, CRC flags and last flip sequences, on two calls in a row, the second with fewer frames (the
scratch array keeps the first call's contents)."""
import numpy as np
import pytest

import _native
import functions as pcf
from channel import Channel

pytestmark = pytest.mark.gpu


def _setup(N, K, poly):
    import torch

    ctx = _native.default_ctx()
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
    code.set_crc(poly)
    code.set_reliability(pcf.GA(N, K, 2.0))
    return torch, ctx, code


def _both(torch, ctx, code, llr, B, T, dtype):
    dev = llr.device
    outs = []
    for fn in (code.pbf_decode_dev, code.pbf_decode_noalloc_dev):
        hat = torch.full((B, code.Kw), -1, dtype=torch.int32, device=dev)
        runs = torch.full((B,), -7, dtype=torch.int32, device=dev)
        ok = torch.full((B,), 9, dtype=torch.uint8, device=dev)
        last = torch.full((B, 3), -5, dtype=torch.int32, device=dev)
        fn(llr.data_ptr(), dtype, T, hat.data_ptr(), B, runs_ptr=runs.data_ptr(), ok_ptr=ok.data_ptr(), last_ptr=last.data_ptr())
        ctx.sync()
        outs.append([t.cpu().numpy() for t in (hat, runs, ok, last)])
    return outs


@pytest.mark.parametrize("N,K,poly,dtype", [(1024, 536, 0x1800063, np.float32), (256, 136, 0x107, np.float64)])
def test_noalloc_entry_point_equals_allocating_one(N, K, poly, dtype):
    torch, ctx, code = _setup(N, K, poly)
    w = code.crc_width
    ch = Channel("BPSK", 1.5, "SNRb", (K - w) / N)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    dev = torch.device("cuda", ctx.device)
    for B, seed, T in ((6000, 11, 1), (1500, 12, 2 if N == 256 else 1)):  # the second call is smaller than the first
        llr = torch.empty((B, N), dtype=tdt, device=dev)
        code.gen_frames_dev(None, llr.data_ptr(), dtype, ch.noisePower, seed, 0, B, crc=True)
        ctx.sync()
        ref, got = _both(torch, ctx, code, llr, B, T, dtype)
        assert (ref[1] > 1).sum() > 20, "the sample must hold frames that enter the flip search"
        for a, b, name in zip(ref, got, ("info_hat", "runs", "crc_ok", "last_seq")):
            assert np.array_equal(a, b), name


def test_noalloc_entry_point_rejects_what_it_cannot_hold():
    torch, ctx, code = _setup(128, 72, 0x107)
    with pytest.raises(_native.PolarError, match="exceed the scratch array"):
        code.pbf_decode_noalloc_dev(1, np.float32, 1, 1, (1 << 20) + 1)
    big = _native.Code(ctx, pcf.degaBuild(2048, 1032, 2.0))
    big.set_crc(0x107)
    big.set_reliability(pcf.GA(2048, 1032, 2.0))
    with pytest.raises(_native.PolarError, match="longer than 1024"):
        code_ptr = torch.zeros(8, dtype=torch.int32, device="cuda")
        big.pbf_decode_noalloc_dev(code_ptr.data_ptr(), np.float32, 1, code_ptr.data_ptr(), 1)
    # the timing bracket synchronises after every launch, which a caller under graph capture must not do
    ctx.enable_timing(True)
    try:
        with pytest.raises(_native.PolarError, match="enable_timing"):
            buf = torch.zeros(1024, dtype=torch.int32, device="cuda")
            code.pbf_decode_noalloc_dev(buf.data_ptr(), np.float32, 1, buf.data_ptr(), 1)
    finally:
        ctx.enable_timing(False)
