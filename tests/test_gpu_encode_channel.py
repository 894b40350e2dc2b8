"""GPU parity: encoder (bit-exact vs reference vectors and oracle), channel / Philox stream vs oracle."""
import numpy as np
import pytest

import _native
import functions as pcf
from channel import Channel
from frames import pack, unpack
from oracle import oracle
from polarCode import PolarCode

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K", [(8, 4), (128, 64), (512, 256), (1024, 512), (16384, 8192)])
def test_encoder_matches_reference_vectors(golden, N, K):
    g = golden.encoder
    pc = PolarCode(N, K, "dega", 2.0)
    info = unpack(g[f"info_{N}_{K}"], K)
    x = pc.polarEncode(info)
    assert x.dtype.kind == "i" and x.shape == (info.shape[0], N)
    assert np.array_equal(pack(x), g[f"x_{N}_{K}"])
    u = unpack(g[f"u_{N}"], N).astype(int)
    keep = u.copy()
    xu = pc.mul_matrix(u)
    assert xu is u, "mul_matrix works in place like the reference"
    assert np.array_equal(pack(xu), g[f"xu_{N}"])
    # involution: F^{(x)n} is its own inverse over GF(2)
    assert np.array_equal(pc.mul_matrix(xu.copy()), keep)


def test_encoder_kat_and_single_frame_shapes():
    pc = PolarCode(8, 4, "dega", 2.0)
    x = pc.polarEncode(np.array([1, 0, 1, 1]))
    assert x.shape == (8,) and x.tolist() == [1, 0, 1, 0, 0, 1, 0, 1]
    assert pc.rateProfile([1, 0, 1, 1]).tolist() == [0, 0, 0, 1, 0, 0, 1, 1]
    assert pc.extract(np.array([0, 0, 0, 1, 0, 1, 1, 1])).tolist() == [1, 1, 1, 1]


@pytest.mark.parametrize("N", [2, 4, 16, 32, 64, 2048, 32768])
@pytest.mark.parametrize("rate", [0.0, 0.3, 1.0])
def test_encoder_sizes_and_rates_vs_oracle(N, rate):
    rs = np.random.RandomState(N + int(rate * 10))
    K = int(round(N * rate))
    mask = np.zeros(N, dtype=int)
    mask[rs.permutation(N)[:K]] = 1
    pc = PolarCode.from_mask(mask)
    B = 37 if N <= 2048 else 3
    info = rs.randint(0, 2, size=(B, K))
    x = pc.polarEncode(info)
    assert np.array_equal(x, oracle.polar_encode(info, mask).reshape(B, N))


def test_encoder_linearity_large_batch():
    pc = PolarCode(1024, 512, "dega", 2.0)
    rs = np.random.RandomState(11)
    a = rs.randint(0, 2, size=(20000, 512))
    b = rs.randint(0, 2, size=(20000, 512))
    assert np.array_equal(pc.polarEncode(a ^ b), pc.polarEncode(a) ^ pc.polarEncode(b))
    assert not pc.polarEncode(np.zeros((3, 512), dtype=int)).any()
    assert pc.polarEncode(np.zeros((0, 512), dtype=int)).shape == (0, 1024)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 3e-5), (np.float64, 1e-12)])
def test_channel_matches_oracle_stream(dtype, tol):
    """Same Philox counters, same Box-Muller: device LLRs equal the oracle's within rounding of the
    float math library (tolerance: 3e-5 absolute in fp32, 1e-12 in fp64, on LLRs of magnitude ~3)."""
    N, K = 1024, 512
    mask = pcf.degaBuild(N, K, 2.0)
    ch = Channel("BPSK", 2.0, "SNRb", 0.5)
    code = _native.Code(_native.default_ctx(), mask)
    seed, frame0, B = 99, 123456789012, 9
    xs, ref = [], []
    for b in range(B):
        info, x, llr = oracle.gen_frame(seed, frame0 + b, mask, ch.noisePower)
        xs.append(x)
        ref.append(llr)
    got = code.channel_llr(np.stack(xs), ch.noisePower, seed, frame0, dtype)
    assert got.dtype == dtype
    np.testing.assert_allclose(got, np.stack(ref), rtol=0, atol=tol)
    got2 = Channel("BPSK", 2.0, "SNRb", 0.5).llrBatch(np.stack(xs), seed, frame0, dtype)
    assert np.array_equal(got, got2)


def test_channel_statistics():
    N = 1024
    code = _native.scratch_code(N)
    x = np.zeros((2000, N), dtype=np.uint8)
    n0 = 1.2619146889603865
    llr = code.channel_llr(x, n0, 5, 0, np.float32).astype(np.float64)
    z = (llr * n0 / 4 - 1.0) / np.sqrt(n0 / 2)
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 3e-3
    from scipy import stats
    assert stats.kstest(z.ravel()[::7], "norm").pvalue > 1e-3
    # frames are keyed by their global index, not by their position in the batch
    part = code.channel_llr(x[:10], n0, 5, 100, np.float32)
    assert np.array_equal(part, code.channel_llr(x[:200], n0, 5, 0, np.float32)[100:110])
