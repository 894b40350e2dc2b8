"""Edge cases of the C ABI on the GPU: empty batches, batch sizes that fill no whole wavefront or workgroup (with guard rows
behind every output: a kernel that writes past its batch is caught), a batch decoded as a whole against the same frames decoded
in ragged pieces, and the error codes for arguments no kernel is written for."""
import numpy as np
import pytest
import torch

import _native
import functions as pcf
from channel import Channel

pytestmark = pytest.mark.gpu

GUARD = 0x5A5A5A5A  # pattern of the rows behind the batch
RAGGED = [1, 2, 3, 5, 63, 65, 257, 1023]


def _code(N, K, cons="dega"):
    ctx = _native.default_ctx(0)
    mask = pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)
    return ctx, _native.Code(ctx, mask)


def _llrs(code, N, K, B, dtype=np.float32, snr=1.5):
    tdt = torch.float32 if dtype is np.float32 else torch.float64
    llr = torch.empty((B, N), dtype=tdt, device="cuda")
    info = torch.empty((B, code.Kw), dtype=torch.int32, device="cuda")
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), dtype, Channel("BPSK", snr, "SNRb", K / N).noisePower, 5, 0, B)
    return llr, info


def _pieces_equal_whole(ctx, code, llr, decode):
    """decode(llr_ptr, hat_ptr, B) on the whole batch, then on ragged pieces of it into buffers with guard rows."""
    Btot = llr.shape[0]
    whole = torch.zeros((Btot, code.Kw), dtype=torch.int32, device="cuda")
    decode(llr.data_ptr(), whole.data_ptr(), Btot)
    ctx.sync()
    whole = whole.cpu().numpy()
    b0 = 0
    for B in RAGGED:
        assert b0 + B <= Btot
        out = torch.full((B + 4, code.Kw), GUARD, dtype=torch.int32, device="cuda")
        decode(llr[b0:].data_ptr(), out.data_ptr(), B)
        ctx.sync()
        out = out.cpu().numpy()
        assert np.array_equal(out[:B], whole[b0:b0 + B]), f"piece of {B} frames at {b0}"
        assert (out[B:] == GUARD).all(), f"piece of {B} frames wrote behind its batch"
        b0 += B
    return whole


@pytest.mark.parametrize("N,K,specialised", [(64, 32, False), (256, 128, False), (1024, 512, False), (1024, 512, True), (2048, 1024, False)])
def test_sc_ragged_batches(N, K, specialised):
    ctx, code = _code(N, K)
    if specialised:
        code.specialize()
    llr, info = _llrs(code, N, K, sum(RAGGED))
    for fast in (True, False):
        _pieces_equal_whole(ctx, code, llr, lambda l, h, B: code.sc_decode_dev(l, np.float32, h, B, fast=fast))
    llr64, _ = _llrs(code, N, K, sum(RAGGED), np.float64)
    if N <= 1024:
        _pieces_equal_whole(ctx, code, llr64, lambda l, h, B: code.sc_decode_dev(l, np.float64, h, B, fast=False))


@pytest.mark.parametrize("N,K,L,dtype", [(128, 64, 4, np.float32), (512, 256, 8, np.float32), (1024, 512, 8, np.float32),
                                         (256, 128, 32, np.float32), (1024, 512, 16, np.float32), (512, 256, 8, np.float64),
                                         (2048, 1024, 4, np.float32)])
def test_scl_ragged_batches(N, K, L, dtype):
    ctx, code = _code(N, K)
    llr, info = _llrs(code, N, K, sum(RAGGED), dtype)
    _pieces_equal_whole(ctx, code, llr, lambda l, h, B: code.scl_decode_dev(l, dtype, L, h, B))


def test_fused_count_ragged_batches():
    """polar_sc_decode_count on ragged batches: counters equal a separate comparison of the decoded words."""
    N, K = 1024, 512
    ctx, code = _code(N, K)
    code.specialize()
    llr, info = _llrs(code, N, K, sum(RAGGED))
    hat = torch.zeros((sum(RAGGED), code.Kw), dtype=torch.int32, device="cuda")
    code.sc_decode_dev(llr.data_ptr(), np.float32, hat.data_ptr(), sum(RAGGED), fast=True)
    ctx.sync()
    diff = (hat ^ info).cpu().numpy().view(np.uint32)
    bit_err = np.array([sum(bin(int(w)).count("1") for w in row) for row in diff])
    b0 = 0
    for B in RAGGED:
        cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
        code.sc_decode_count_dev(llr[b0:].data_ptr(), np.float32, info[b0:].data_ptr(), B, cnt.data_ptr(), fast=True)
        ctx.sync()
        be = bit_err[b0:b0 + B]
        assert cnt.cpu().tolist() == [B, int((be > 0).sum()), int(be.sum())]
        b0 += B


def test_empty_batches_touch_nothing():
    N, K = 128, 64
    ctx, code = _code(N, K)
    code.set_crc(0x107)
    code.set_reliability(pcf.GA(N, K, 2.0))
    llr = torch.zeros((4, N), dtype=torch.float32, device="cuda")
    out = torch.full((4, code.Kw), GUARD, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
    code.sc_decode_dev(llr.data_ptr(), np.float32, out.data_ptr(), 0, fast=True)
    code.sc_decode_dev(llr.data_ptr(), np.float32, out.data_ptr(), 0, fast=False)
    code.scl_decode_dev(llr.data_ptr(), np.float32, 8, out.data_ptr(), 0)
    code.scl_decode_dev(llr.data_ptr(), np.float32, 32, out.data_ptr(), 0)
    code.pbf_decode_dev(llr.data_ptr(), np.float32, 1, out.data_ptr(), 0)
    code.sc_decode_count_dev(llr.data_ptr(), np.float32, out.data_ptr(), 0, cnt.data_ptr(), fast=True)
    code.gen_frames_dev(out.data_ptr(), llr.data_ptr(), np.float32, 1.0, 1, 0, 0)
    code.count_errors_dev(out.data_ptr(), out.data_ptr(), 0, cnt.data_ptr())
    ctx.sync()
    assert (out.cpu().numpy() == GUARD).all() and cnt.cpu().tolist() == [0, 0, 0] and not llr.cpu().numpy().any()
    # host space, numpy in and out
    e = np.zeros((0, N), dtype=np.float32)
    assert code.sc_decode(e)[0].shape == (0, K) and code.scl_decode(e, 4)[0].shape == (0, K)
    assert code.encode(np.zeros((0, K), dtype=np.uint8)).shape == (0, N)
    assert code.mc_run(_native.DEC_SC, 1, np.float32, 1.0, 1, 0, 0).tolist() == [0, 0, 0]


def test_arguments_no_kernel_is_written_for_are_refused():
    ctx = _native.default_ctx(0)
    for bad_n in (0, 1, 3, 96, 1 << 17):
        with pytest.raises(_native.PolarError):
            _native.Code(ctx, np.ones(max(bad_n, 1), dtype=np.uint8))
    N, K = 128, 64
    ctx, code = _code(N, K)
    llr = torch.zeros((4, N), dtype=torch.float32, device="cuda")
    out = torch.zeros((4, code.Kw), dtype=torch.int32, device="cuda")
    for L in (0, 3, 6, 64):
        with pytest.raises(_native.PolarError, match="L must be"):
            code.scl_decode_dev(llr.data_ptr(), np.float32, L, out.data_ptr(), 4)
    with pytest.raises(_native.PolarError):
        code.sc_decode_dev(None, np.float32, out.data_ptr(), 4)
    with pytest.raises(_native.PolarError):
        code.sc_decode_dev(llr.data_ptr(), np.float32, out.data_ptr(), -1)
    with pytest.raises(_native.PolarError):  # no CRC attached
        code.scl_decode_dev(llr.data_ptr(), np.float32, 4, out.data_ptr(), 4, crc=True)
    with pytest.raises(_native.PolarError):  # flip search without CRC and reliabilities
        code.pbf_decode_dev(llr.data_ptr(), np.float32, 1, out.data_ptr(), 4)
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SC, 1, np.float32, -1.0, 1, 0, 16)
    # the library still works after the refusals
    code.sc_decode_dev(llr.data_ptr(), np.float32, out.data_ptr(), 4)
    ctx.sync()
