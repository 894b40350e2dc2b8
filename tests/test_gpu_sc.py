"""GPU parity of the SC decoder: bit-exact uHat and decision LLRs against the reference's vectors and the
oracle in fp64; bit-exact against the fp32 oracle in fp32; fast schedule == plain schedule."""
import hashlib

import numpy as np
import pytest

import _native
import functions as pcf
from channel import Channel
from frames import make_frame, pack, unpack
from oracle import oracle
from polarCode import PolarCode

pytestmark = pytest.mark.gpu

SC_TAGS = [("dega", 128, 64, 1.0), ("dega", 128, 64, 2.0), ("dega", 128, 64, 4.0), ("dega", 1024, 512, 1.0),
           ("dega", 1024, 512, 2.0), ("dega", 1024, 512, 3.0), ("rmpolar", 512, 256, 2.0), ("dega", 16384, 8192, 2.0)]


def test_sc_kat_8_4(golden):
    g = golden.sc
    pc = PolarCode(8, 4, "dega", 2.0)
    pc.decoderType = "SC"
    dec = pc.decode(g["kat8_llr"])
    assert dec.tolist() == [1, 1, 1, 1] and dec.dtype.kind == "i"
    assert pc.uHat.tolist() == [0, 0, 0, 1, 0, 1, 1, 1]
    assert pc.llrs.tolist() == [1, -3, 1.5, -3.5, -2.5, -2, -6, -31.1]
    assert pc.ANV == 8


@pytest.mark.parametrize("cons,N,K,snr", SC_TAGS)
def test_sc_fp64_matches_reference_vectors(golden, cons, N, K, snr):
    g = golden.sc
    tag = f"{cons}_{N}_{K}_{snr}"
    pc = PolarCode(N, K, cons, 2.0)
    seeds = g[f"{tag}_seeds"]
    frames = [make_frame(pc.polarMask, snr, int(s), oracle.polar_encode) for s in seeds]
    llr = np.stack([f[2] for f in frames])
    dec = pc.scDecoder(llr)  # one batched call
    assert np.array_equal(pack(pc.uHat), g[f"{tag}_uhat"])
    for f in range(len(seeds)):
        assert hashlib.sha1(pc.llrs[f].tobytes()).digest() == g[f"{tag}_decllr_sha1"][f].tobytes(), (tag, f)
        assert int(np.sum(dec[f] != frames[f][0])) == int(g[f"{tag}_biterr"][f])
    # fast schedule (frozen subtrees skipped, all-information subtrees hard-decided): same bits
    info_fast, uh_fast, _ = pc.native.sc_decode(llr, want_uhat=True, fast=True)
    assert np.array_equal(uh_fast, pc.uHat)
    assert np.array_equal(info_fast, dec)


@pytest.mark.parametrize("lanes", [8, 16, 32, 64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sc_every_lane_width_vs_oracle(monkeypatch, lanes, dtype):
    """All lane-group widths of the LDS-staged kernel, ragged batch sizes, both arithmetic types."""
    monkeypatch.setenv("POLAR_SC_LANES", str(lanes))
    monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
    rs = np.random.RandomState(lanes)
    for N, K in [(64, 20), (256, 128), (1024, 512)]:
        mask = pcf.degaBuild(N, K, 2.0)
        code = _native.Code(_native.default_ctx(), mask)
        B = 1 + 64 // lanes * 3 + 1  # not a multiple of the frames-per-wavefront
        frames = [make_frame(mask, 1.5, int(rs.randint(1 << 30)), oracle.polar_encode) for _ in range(B)]
        llr = np.stack([f[2] for f in frames]).astype(dtype)
        info, uh, dl = code.sc_decode(llr, want_uhat=True, want_llrs=True)
        assert code.last_launch()["lanes_per_frame"] == lanes
        for b in range(B):
            ruh, rdl = oracle.sc_decode(llr[b], mask, want_llrs=True)
            assert np.array_equal(uh[b], ruh), (N, b)
            assert np.array_equal(dl[b], rdl), (N, b)
            assert np.array_equal(info[b], oracle.extract(ruh, mask))
        info_f, uh_f, _ = code.sc_decode(llr, want_uhat=True, fast=True)
        assert np.array_equal(uh_f, uh)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sc_register_kernel_with_side_outputs_vs_oracle(monkeypatch, dtype):
    """The register-resident every-leaf kernel (u_hat and decision LLRs, fp32 and fp64; what PolarCode.scDecoder runs for
    N = 32 .. 1024): bits and decision LLRs equal the oracle's exactly, and the LDS-staged kernel's, for every length,
    extreme rates and batch sizes that do not fill the last wavefront."""
    rs = np.random.RandomState(5)
    for N, K in [(32, 16), (64, 0), (64, 64), (128, 100), (256, 3), (512, 256), (1024, 512), (1024, 1023)]:
        mask = np.zeros(N, dtype=int)
        mask[np.sort(rs.permutation(N)[:K])] = 1
        if (N, K) in ((512, 256), (1024, 512)):
            mask = pcf.degaBuild(N, K, 2.0)
        code = _native.Code(_native.default_ctx(), mask)
        B = 11
        llr = (rs.standard_normal((B, N)) * 3 + 1).astype(dtype)
        llr[0, :8] = 0.0  # exact zeros decide 0
        monkeypatch.delenv("POLAR_SC_KERNEL", raising=False)
        info, uh, dl = code.sc_decode(llr, want_uhat=True, want_llrs=True)
        assert code.last_launch()["kernel"].startswith("sc_reg_full_kernel"), code.last_launch()
        for b in range(B):
            ruh, rdl = oracle.sc_decode(llr[b], mask, want_llrs=True)
            assert np.array_equal(uh[b], ruh), (N, K, b)
            assert np.array_equal(dl[b], rdl), (N, K, b)
            assert np.array_equal(info[b], oracle.extract(ruh, mask))
        monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
        info2, uh2, dl2 = code.sc_decode(llr, want_uhat=True, want_llrs=True)
        assert np.array_equal(uh, uh2) and np.array_equal(dl, dl2) and np.array_equal(info, info2)


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32, 64])
def test_sc_small_lengths_and_extreme_rates(N):
    rs = np.random.RandomState(N)
    for K in sorted({0, 1, N // 2, N - 1, N}):
        mask = np.zeros(N, dtype=int)
        mask[rs.permutation(N)[:K]] = 1
        code = _native.Code(_native.default_ctx(), mask)
        llr = rs.standard_normal((5, N)) * 4
        for fast in (False, True):
            info, uh, _ = code.sc_decode(llr, want_uhat=True, fast=fast)
            assert info.shape == (5, K)
            for b in range(5):
                assert np.array_equal(uh[b], oracle.sc_decode(llr[b], mask)), (N, K, fast, b)


def test_sc_zero_llr_and_errors():
    """A zero decision LLR decodes to 0 (the reference blocks on input() and then keeps 0)."""
    mask = np.ones(8, dtype=int)
    code = _native.Code(_native.default_ctx(), mask)
    info, uh, dl = code.sc_decode(np.zeros((1, 8)), want_uhat=True, want_llrs=True)
    assert not uh.any() and not dl.any()
    with pytest.raises(_native.PolarError):
        _native.Code(_native.default_ctx(), np.ones(12))
    rc = _native.lib().polar_sc_decode(code._h, None, 0, None, None, None, 1, 0, 0)
    assert rc == -1 and b"llr" in _native.lib().polar_last_error()
    assert code.sc_decode(np.zeros((0, 8)))[0].shape == (0, 8)
    pc = PolarCode(8, 4, "dega", 2.0)
    pc.decoderType = "fano"
    with pytest.raises(NotImplementedError):
        pc.decode(np.ones(8))


def test_sc_full_size_round_trip():
    """(1024,512) at the benchmark's batch scale: noiseless and high-SNR frames must decode to the
    transmitted message; the fused generator + decoder + counter agree with a host-side count."""
    pc = PolarCode(1024, 512, "dega", 2.0)
    rs = np.random.RandomState(1)
    B = 30000
    msg = rs.randint(0, 2, size=(B, 512))
    x = pc.polarEncode(msg)
    llr = (4.0 * (1 - 2 * x)).astype(np.float32)  # noiseless
    info, _, _ = pc.native.sc_decode(llr, fast=True)
    assert np.array_equal(info, msg)
    ch = Channel("BPSK", 6.0, "SNRb", 0.5)
    llr = pc.native.channel_llr(x, ch.noisePower, 7, 0, np.float32)
    info, _, _ = pc.native.sc_decode(llr, fast=True)
    assert np.mean(np.any(info != msg, axis=1)) < 1e-3


def test_mc_counters_match_oracle_and_are_chunk_invariant():
    """Fused Monte-Carlo run in fp64: identical frames (Philox) on both sides; decisions can only differ
    where device and host libm round a noise sample differently near a decision boundary, so the frame
    error counts agree to within a few frames in 3000."""
    mask = pcf.degaBuild(128, 64, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    n0 = Channel("BPSK", 2.0, "SNRb", 0.5).noisePower
    ref = oracle.mc_run(mask, 0, 1, 1, n0, 2024, 1000, 3000)
    got = code.mc_run(_native.DEC_SC, 1, np.float64, n0, 2024, 1000, 3000, fast=False)
    assert int(got[0]) == 3000
    assert abs(int(got[1]) - int(ref[1])) <= 3 and abs(int(got[2]) - int(ref[2])) <= 60, (got, ref)
    a = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 2024, 0, 5000)
    b = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 2024, 0, 2000) + code.mc_run(_native.DEC_SC, 1, np.float32, n0, 2024, 2000, 3000)
    assert np.array_equal(a, b), "counters must not depend on how the frame range is split"


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024])
def test_sc_register_kernel_vs_oracle_fp32(N):
    """The register-resident kernel (fp32, info bits only) against the fp32 oracle, bit-exact, for every
    supported length, ragged batches, plain and fast schedules, several rates."""
    rs = np.random.RandomState(N)
    for K in sorted({1, N // 4, N // 2, (3 * N) // 4, N - 1}):
        mask = pcf.degaBuild(N, K, 2.0)
        code = _native.Code(_native.default_ctx(), mask)
        B = 23
        frames = [make_frame(mask, 1.0 + 2.0 * rs.rand(), int(rs.randint(1 << 30)), oracle.polar_encode) for _ in range(B)]
        llr = np.stack([f[2] for f in frames]).astype(np.float32)
        ref = np.stack([oracle.extract(oracle.sc_decode(llr[b], mask), mask) for b in range(B)])
        for fast in (True, False):
            info, _, _ = code.sc_decode(llr, fast=fast)
            assert code.last_launch()["kernel"].startswith("sc_reg_kernel"), code.last_launch()
            assert np.array_equal(info, ref), (N, K, fast)
    # random (non-nested) rate profiles exercise every 16-leaf subtree pattern
    for trial in range(4):
        mask = (rs.rand(N) < 0.5).astype(int)
        mask[0], mask[-1] = 0, 1
        code = _native.Code(_native.default_ctx(), mask)
        llr = (rs.standard_normal((9, N)) * 3).astype(np.float32)
        ref = np.stack([oracle.extract(oracle.sc_decode(llr[b], mask), mask) for b in range(9)])
        for fast in (True, False):
            info, _, _ = code.sc_decode(llr, fast=fast)
            assert np.array_equal(info, ref), (N, trial, fast)


def test_sc_register_and_lds_kernels_agree(monkeypatch):
    pc = PolarCode(1024, 512, "dega", 2.0)
    ch = Channel("BPSK", 1.5, "SNRb", 0.5)
    rs = np.random.RandomState(2)
    x = pc.polarEncode(rs.randint(0, 2, size=(4001, 512)))
    llr = pc.native.channel_llr(x, ch.noisePower, 3, 0, np.float32)
    a, _, _ = pc.native.sc_decode(llr, fast=True)
    assert pc.native.last_launch()["kernel"].startswith("sc_reg_kernel")
    monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
    b, _, _ = pc.native.sc_decode(llr, fast=True)
    assert pc.native.last_launch()["kernel"].startswith("sc_decode_kernel")
    assert np.array_equal(a, b)


PROFILES16 = [0x8000, 0xC000, 0xE000, 0xE800, 0xE880, 0xF000, 0xF800, 0xF880, 0xFC00, 0xFC80, 0xFCC0, 0xFE00, 0xFE80, 0xFEC0,
              0xFEE0, 0xFEE8, 0xFF00, 0xFF80, 0xFFC0, 0xFFE0, 0xFFE8, 0xFFF0, 0xFFF8, 0xFFFC, 0xFFFE]


def test_sc_every_compiled_16_leaf_profile(monkeypatch):
    """A rate profile whose 16-leaf nodes run through all 25 profiles the register kernel has compiled code for,
    the all-frozen / all-information ones and some it has not: register kernel (fp32), and the LDS-staged kernel
    with its in-register 16-leaf operation (fp32 and fp64), against the oracle."""
    rs = np.random.RandomState(16)
    blocks = PROFILES16 + [0x0000, 0xFFFF, 0x0001, 0x5555, 0x8001, 0x00FF, 0x7FFE] + [int(v) for v in rs.randint(0, 1 << 16, size=32)]
    blocks = [blocks[i] for i in rs.permutation(len(blocks))]
    mask = np.array([(m >> i) & 1 for m in blocks for i in range(16)], dtype=int)
    N = mask.size
    assert N == 1024
    code = _native.Code(_native.default_ctx(), mask)
    llr64 = rs.standard_normal((37, N)) * 2.5 + 0.7
    for dtype in (np.float32, np.float64):
        llr = llr64.astype(dtype)
        ref_uh = np.stack([oracle.sc_decode(llr[b], mask) for b in range(llr.shape[0])])
        ref = ref_uh[:, mask == 1]
        for fast in (True, False):
            monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
            info, uh, _ = code.sc_decode(llr, want_uhat=True, fast=fast)      # LDS-staged kernel, no decision LLRs
            monkeypatch.delenv("POLAR_SC_KERNEL")
            assert code.last_launch()["kernel"].startswith("sc_decode_kernel")
            assert np.array_equal(uh, ref_uh) and np.array_equal(info, ref), (dtype, fast)
            info, uh, _ = code.sc_decode(llr, want_uhat=True, fast=False)     # the register kernel with side outputs
            assert code.last_launch()["kernel"].startswith("sc_reg_full_kernel")
            assert np.array_equal(uh, ref_uh) and np.array_equal(info, ref), dtype
            if dtype == np.float32:
                info, _, _ = code.sc_decode(llr, fast=fast)
                assert code.last_launch()["kernel"].startswith("sc_reg_kernel")
                assert np.array_equal(info, ref), fast


@pytest.mark.parametrize("N,K", [(32, 16), (64, 20), (128, 64), (256, 200), (512, 100), (1024, 512), (1024, 800), (1024, 1023)])
def test_sc_specialised_kernel_equals_generic(monkeypatch, N, K):
    """The decoder compiled at run time for one rate profile against the generic register kernel and the oracle."""
    rs = np.random.RandomState(N + K)
    for mask in (pcf.degaBuild(N, K, 2.0), np.r_[0, (rs.rand(N - 2) < 0.5).astype(int), 1]):
        code = _native.Code(_native.default_ctx(), mask)
        Kc = int(mask.sum())
        x = oracle.polar_encode(rs.randint(0, 2, size=(203, Kc)), mask)
        llr = code.channel_llr(x, n0_of(1.5, Kc / N), 9, 0, np.float32)
        a, _, _ = code.sc_decode(llr, fast=True)
        assert code.last_launch()["kernel"].startswith("sc_reg_kernel")
        code.specialize()
        b, _, _ = code.sc_decode(llr, fast=True)
        assert code.last_launch()["kernel"].startswith("sc_static_kernel"), code.last_launch()
        assert np.array_equal(a, b)
        ref = np.stack([oracle.extract(oracle.sc_decode(llr[f], mask), mask) for f in range(0, 203, 7)])
        assert np.array_equal(b[::7], ref)
        monkeypatch.setenv("POLAR_SC_KERNEL", "generic")
        c, _, _ = code.sc_decode(llr, fast=True)
        assert code.last_launch()["kernel"].startswith("sc_reg_kernel")
        monkeypatch.delenv("POLAR_SC_KERNEL")
        # the fused Monte-Carlo loop picks the specialised decoder up as well
        plain = _native.Code(_native.default_ctx(), mask)
        n0 = n0_of(2.0, Kc / N)
        assert code.mc_run(_native.DEC_SC, 1, np.float32, n0, 5, 0, 20000).tolist() == plain.mc_run(_native.DEC_SC, 1, np.float32, n0, 5, 0, 20000).tolist()


@pytest.mark.parametrize("N,K", [(64, 32), (1024, 512), (1024, 700)])
def test_sc_signed_zeros_and_extremes(monkeypatch, N, K):
    """Decisions are `LLR < 0`: +0 and -0 both decide 0, also where they only arise inside the tree (f of a zero and a
    negative value is -0; equal magnitudes cancel to +0).  Zeros, signed zeros, repeated magnitudes, denormals and huge
    values: the every-leaf schedule equals the oracle on all of them; the schedule with short-cuts takes the signs of
    an all-information subtree's LLRs, which is the leaf-by-leaf result unless one of those LLRs is exactly zero, so
    it is held to the oracle on the all-zero rows and otherwise to itself: generic register kernel, specialised
    kernel and LDS-staged kernel must agree bit for bit."""
    rs = np.random.RandomState(N * 3 + K)
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    B = 64
    llr = np.zeros((B, N), dtype=np.float32)
    llr[1] = -0.0
    llr[2] = np.where(rs.rand(N) < 0.5, 0.0, -0.0)
    llr[3] = rs.choice([-1.0, 1.0], size=N)                         # every g cancels or doubles
    llr[4] = rs.choice([-2.0, -1.0, -0.0, 0.0, 1.0, 2.0], size=N)
    llr[5] = rs.choice([-1e-42, 1e-42, 0.0, -0.0], size=N)          # denormals
    llr[6] = rs.choice([-3e37, 3e37, 1.0, -1.0], size=N)            # sums that overflow to infinity
    for b in range(7, B):
        v = rs.choice([-2.0, -1.0, -0.5, -0.0, 0.0, 0.5, 1.0, 2.0], size=N) * rs.choice([1.0, 1.0, 3.0], size=N)
        llr[b] = np.where(rs.rand(N) < 0.3, 0.0, v)
    ref = np.stack([oracle.extract(oracle.sc_decode(llr[b], mask), mask) for b in range(B)])
    plain, _, _ = code.sc_decode(llr, fast=False)
    assert code.last_launch()["kernel"].startswith("sc_reg_kernel")
    assert np.array_equal(plain, ref)
    gen, _, _ = code.sc_decode(llr, fast=True)
    assert code.last_launch()["kernel"].startswith("sc_reg_kernel")
    assert np.array_equal(gen[:3], ref[:3])
    monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
    lds, _, _ = code.sc_decode(llr, fast=True)
    assert code.last_launch()["kernel"].startswith("sc_decode_kernel")
    monkeypatch.delenv("POLAR_SC_KERNEL")
    assert np.array_equal(lds, gen)
    code.specialize()
    spec, _, _ = code.sc_decode(llr, fast=True)
    assert code.last_launch()["kernel"].startswith("sc_static_kernel")
    assert np.array_equal(spec, gen)


@pytest.mark.parametrize("N,K", [(128, 64), (256, 31), (512, 300), (1024, 512), (1024, 800)])
def test_mc_fused_kernel_equals_staged_pipeline(monkeypatch, N, K):
    """The single kernel that draws the bits, encodes, adds the noise, decodes and counts (specialised codes, fp32 SC)
    against generator -> LLR buffer -> decoder (whose parts are held to the oracle by the tests above), for frame counts
    that do not fill the last wavefront and frame offsets beyond 32 bits.  The counters must be identical."""
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    code.specialize()
    n0 = n0_of(1.5, K / N)
    for seed, frame0, nframes in ((3, 0, 1), (3, 5, 37), (11, 1 << 33, 20011)):
        fused = code.mc_run(_native.DEC_SC, 1, np.float32, n0, seed, frame0, nframes).tolist()
        assert code.last_launch()["kernel"].startswith("sc_static_mc_kernel"), code.last_launch()
        monkeypatch.setenv("POLAR_MC_KERNEL", "staged")
        staged = code.mc_run(_native.DEC_SC, 1, np.float32, n0, seed, frame0, nframes).tolist()
        assert code.last_launch()["kernel"].startswith("sc_static_kernel"), code.last_launch()
        monkeypatch.delenv("POLAR_MC_KERNEL")
        assert fused == staged and fused[0] == nframes, (seed, frame0, nframes, fused, staged)
    # frames split over calls add up (what ranks and rounds of the simulator rely on)
    a = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 5, 100, 3000)
    b = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 5, 100, 1234) + code.mc_run(_native.DEC_SC, 1, np.float32, n0, 5, 1334, 1766)
    assert a.tolist() == b.tolist()


@pytest.mark.parametrize("N,K", [(2048, 1000), (2048, 37), (4096, 3000), (8192, 4096), (16384, 8192), (32768, 20000)])
def test_sc_long_codes_slab_kernel_vs_oracle_and_lds_kernel(monkeypatch, N, K):
    """Codes longer than 1024 in fp32: the kernel that keeps the wide levels in a global slab and decodes 1024-leaf
    subtrees in registers, against the oracle (bit-exact) and the LDS-staged kernel, ragged batch sizes included."""
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + K)
    B = 21
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask)
    llr = code.channel_llr(x, n0_of(1.0, K / N), 3, 0, np.float32)
    got, _, _ = code.sc_decode(llr, fast=True)
    assert code.last_launch()["kernel"].startswith("sc_big_kernel"), code.last_launch()
    nref = 4 if N <= 8192 else 2
    ref = np.stack([oracle.extract(oracle.sc_decode(llr[b], mask), mask) for b in range(nref)])
    assert np.array_equal(got[:nref], ref)
    monkeypatch.setenv("POLAR_SC_KERNEL", "lds")
    lds, _, _ = code.sc_decode(llr, fast=True)
    assert code.last_launch()["kernel"].startswith("sc_decode_kernel")
    monkeypatch.delenv("POLAR_SC_KERNEL")
    assert np.array_equal(got, lds)
    one, _, _ = code.sc_decode(llr[:1], fast=True)
    assert np.array_equal(one, got[:1])


def n0_of(snr_db, rate):
    return 1.0 / (rate * 10 ** (snr_db / 10))


@pytest.mark.parametrize("N,K,dtype", [(1024, 512, np.float32), (256, 77, np.float32), (64, 33, np.float32), (1024, 512, np.float64), (2048, 1000, np.float32)])
def test_sc_decode_count_fused_equals_two_kernels(N, K, dtype):
    """polar_sc_decode_count (comparison inside the decoder's output stage where the register kernels apply)
    against decode followed by polar_count_errors, generic and specialised kernels, with and without storing."""
    import torch

    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    B = 30001
    dev = torch.device("cuda:0")
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    llr = torch.empty((B, N), dtype=tdt, device=dev)
    info = torch.zeros((B, code.Kw), dtype=torch.int32, device=dev)
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), dtype, n0_of(1.5, K / N), 3, 0, B)
    hat = torch.zeros_like(info)
    ref = torch.zeros(3, dtype=torch.int64, device=dev)
    code.sc_decode_dev(llr.data_ptr(), dtype, hat.data_ptr(), B, fast=True)
    code.count_errors_dev(info.data_ptr(), hat.data_ptr(), B, ref.data_ptr())
    code.ctx.sync()
    assert 0 < ref[1] < B
    for spec in ([False, True] if (dtype == np.float32 and N <= 1024) else [False]):
        if spec:
            code.specialize()
        for store in (True, False):
            got = torch.zeros(3, dtype=torch.int64, device=dev)
            hat2 = torch.zeros_like(info)
            code.sc_decode_count_dev(llr.data_ptr(), dtype, info.data_ptr(), B, got.data_ptr(), fast=True,
                                     info_hat_ptr=hat2.data_ptr() if store else None)
            code.ctx.sync()
            assert got.cpu().tolist() == ref.cpu().tolist(), (spec, store)
            if store:
                assert torch.equal(hat2, hat)


@pytest.mark.parametrize("decoder", ["sc", "sc_staged", "scl", "pbf"])
def test_mc_run_chunked_pipeline_counts_like_one_chunk(monkeypatch, decoder):
    """polar_mc_run in several chunks (POLAR_MC_CHUNK caps the chunk: generator and decoder then overlap on two streams and
    two buffers, ordered by events) counts exactly what one chunk counts."""
    N, K = 256, 128
    if decoder == "pbf":
        K += 8
    code = _native.Code(_native.default_ctx(), pcf.degaBuild(N, K, 2.0))
    n0 = 1.0 / ((128 / N) * 10 ** (1.5 / 10))
    dec, L, kw = _native.DEC_SC, 1, {}
    if decoder == "scl":
        dec, L = _native.DEC_SCL, 4
    elif decoder == "pbf":
        code.set_crc(0x107)
        code.set_reliability(pcf.GA(N, K, 2.0))
        dec, L, kw = _native.DEC_PBF, 1, {"crc": True}
    elif decoder == "sc":
        code.specialize()
    else:
        monkeypatch.setenv("POLAR_MC_KERNEL", "staged")
    nfr = 5 * 4096 + 777
    monkeypatch.delenv("POLAR_MC_CHUNK", raising=False)
    one = code.mc_run(dec, L, np.float32, n0, 99, 12345, nfr, **kw)
    monkeypatch.setenv("POLAR_MC_CHUNK", "4096")
    many = code.mc_run(dec, L, np.float32, n0, 99, 12345, nfr, **kw)
    assert int(one[0]) == nfr and one[1] > 0
    assert one.tolist() == many.tolist()
