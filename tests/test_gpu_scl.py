"""GPU parity of the SCL decoder in fp64 against the reference's vectors (whole sorted list, in order,
path metrics to 1e-12) and the oracle; fp32 mode against the fp32 oracle."""
import numpy as np
import pytest

import _native
import functions as pcf
from frames import make_frame, pack, unpack
from oracle import oracle
from polarCode import PolarCode

pytestmark = pytest.mark.gpu

SCL_TAGS = [("dega", 128, 64, 2, 0.0), ("dega", 128, 64, 2, 2.0), ("dega", 128, 64, 4, 0.0), ("dega", 128, 64, 4, 2.0),
            ("dega", 128, 64, 8, 0.0), ("dega", 128, 64, 16, 0.0), ("dega", 128, 64, 32, 0.0),
            ("dega", 1024, 512, 8, 1.0), ("dega", 1024, 512, 8, 2.0),
            ("rmpolar", 512, 256, 32, 2.0), ("dega", 512, 256, 32, 1.0), ("dega", 512, 256, 32, 2.0)]


def test_scl_kat_8_4(golden):
    g = golden.scl
    pc = PolarCode(8, 4, "dega", 2.0, listSize=4)
    pc.decoderType = "SCL"
    dec = pc.decode(g["kat8_llr"])
    assert dec.tolist() == [1, 1, 1, 1]
    pms = [p.pathMetric for p in pc.pathList]
    np.testing.assert_allclose(pms, [-1.0908631565524596, -3.9727056749214777, -7.787037005675798, -9.229732048891263], rtol=1e-13)
    assert np.array_equal(pack(np.stack([p.uHat for p in pc.pathList])), g["kat8_uhat"])


@pytest.mark.parametrize("cons,N,K,L,snr", SCL_TAGS)
def test_scl_fp64_matches_reference_vectors(golden, cons, N, K, L, snr):
    g = golden.scl
    tag = f"{cons}_{N}_{K}_L{L}_{snr}"
    pc = PolarCode(N, K, cons, 2.0, listSize=L)
    seeds = g[f"{tag}_seeds"]
    frames = [make_frame(pc.polarMask, snr, int(s), oracle.polar_encode) for s in seeds]
    llr = np.stack([f[2] for f in frames])
    dec = pc.sclDecoder(llr)
    if 64 <= N <= 1024 and L > 1 and K >= 5:
        # the reference's arithmetic (fp64) with the whole list as output runs on the register kernels: eight lanes per
        # path up to L = 8, two lanes per path for L = 16 / 32
        want = "scl_reg_kernel" if L <= 8 else "scl_wide_kernel"
        kern = pc.native.last_launch()["kernel"]
        assert kern.startswith(want) and "double" in kern, kern
    assert (g[f"{tag}_gap"] > 1e-9).all()  # no stored frame is sensitive to last-ulp metric differences
    assert np.array_equal(pack(dec), g[f"{tag}_dec"])
    for f in range(len(seeds)):
        got_uh = np.stack([p.uHat for p in pc.pathList[f]])
        assert np.array_equal(pack(got_uh), g[f"{tag}_uhat"][f]), (tag, f)  # every survivor, in list order
        ref = g[f"{tag}_pm"][f]
        pm = pc.pathMetrics[f]
        assert np.array_equal(np.isinf(pm), np.isinf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(pm[fin], ref[fin], rtol=1e-12)
        assert int(np.sum(dec[f] != frames[f][0])) == int(g[f"{tag}_biterr"][f])
    assert [pc.ANV, pc.sortNum, pc.unitCal] == g[f"{tag}_counts"][0].tolist()


def test_scl_seeded_kat_1024():
    pc = PolarCode(1024, 512, "dega", 2.0, listSize=8)
    msg, x, llr = make_frame(pc.polarMask, 2.0, 2024, oracle.polar_encode)
    dec = pc.sclDecoder(llr)
    assert abs(pc.pathList[0].pathMetric - 600.3603000267871) < 1e-9
    assert (pc.ANV, pc.sortNum, pc.unitCal) == (6788, 510, 64900)
    assert np.array_equal(dec, msg)


@pytest.mark.parametrize("L", [1, 2, 4, 8, 16, 32])
def test_scl_small_lengths_vs_oracle(L):
    rs = np.random.RandomState(L)
    for N in (2, 4, 8, 16, 32, 64, 256):
        for K in sorted({0, 1, 3 if N > 4 else 1, N // 2, N}):
            mask = np.zeros(N, dtype=int)
            mask[rs.permutation(N)[:K]] = 1
            code = _native.Code(_native.default_ctx(), mask)
            llr = rs.standard_normal((3, N)) * 2.5
            info, pm, uh = code.scl_decode(llr, L, want_list=True)
            for b in range(3):
                ruh, rpm, _ = oracle.scl_decode(llr[b], mask, L)
                ns = len(rpm)
                assert np.array_equal(uh[b, :ns], ruh), (N, K, L, b)
                np.testing.assert_allclose(pm[b, :ns], rpm, rtol=1e-12, atol=1e-12)
                assert np.all(np.isneginf(pm[b, ns:])) and not uh[b, ns:].any()
                assert np.array_equal(info[b], oracle.extract(ruh[0], mask))


def test_scl_metric_overflow_gives_minus_inf():
    """LLR magnitudes beyond 1024 ln 2 make the wrong-decision metric -inf, as in the reference."""
    mask = np.ones(8, dtype=int)
    code = _native.Code(_native.default_ctx(), mask)
    llr = np.array([[900.0, -900.0, 800.0, 1000.0, -750.0, 720.0, 5.0, -3.0]])
    info, pm, uh = code.scl_decode(llr, 4, want_list=True)
    ruh, rpm, _ = oracle.scl_decode(llr[0], mask, 4)
    assert np.array_equal(uh[0], ruh)
    assert np.array_equal(np.isinf(pm[0]), np.isinf(rpm))
    np.testing.assert_allclose(pm[0][np.isfinite(rpm)], rpm[np.isfinite(rpm)], rtol=1e-12)


def test_scl_fp32_mode_vs_fp32_oracle(monkeypatch):
    """fp32 arithmetic at a short length, (256,128) L = 8, held to the fp32 oracle frame by frame with the margin argument of
    test_scl8_headline_kernel_vs_fp32_oracle: device and host differ only in exp2 / log2 of the branch metric; that
    difference is measured on the lists the LDS-staged kernel returns, and every frame whose closest pair of competing
    metrics (reported by the oracle) lies further apart than twice that difference must decode to the oracle's bits, on
    the register kernel and on the LDS-staged kernel alike."""
    N, K, L = 256, 128, 8
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    frames = [make_frame(mask, 1.5, 4000 + s, oracle.polar_encode) for s in range(400)]
    llr = np.stack([f[2] for f in frames]).astype(np.float32)
    B = len(frames)
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    got, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel<8>/L8"), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    lds, pm, uh = code.scl_decode(llr, L, want_list=True)
    assert np.array_equal(got, lds)
    ref = np.empty_like(got)
    gap = np.empty(B)
    err = 0.0
    for b in range(B):
        ruh, rpm, _ = oracle.scl_decode(llr[b], mask, L)
        gap[b] = oracle.scl_last_min_gap()
        ref[b] = oracle.extract(ruh[0], mask)
        if np.array_equal(uh[b], ruh):
            fin = np.isfinite(rpm)
            assert np.array_equal(np.isfinite(pm[b]), fin)
            err = max(err, float(np.max(np.abs(pm[b][fin] - rpm[fin]) / np.maximum(np.abs(rpm[fin]), 1.0), initial=0.0)))
    margin = 2.0 * err
    same = np.all(got == ref, axis=1)
    inside = gap <= margin
    assert err < 1e-5
    assert same[~inside].all(), np.flatnonzero(~same & ~inside)[:10]
    assert (~same).sum() <= 0.02 * B


def test_scl_list_1_equals_sc_and_L_improves_fer():
    N, K = 256, 128
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    frames = [make_frame(mask, 1.0, 7000 + s, oracle.polar_encode) for s in range(300)]
    llr = np.stack([f[2] for f in frames])
    msg = np.stack([f[0] for f in frames])
    sc, _, _ = code.sc_decode(llr)
    l1, _, _ = code.scl_decode(llr, 1)
    assert np.array_equal(sc, l1)
    l8, _, _ = code.scl_decode(llr, 8)
    assert np.any(l8 != msg, axis=1).sum() <= np.any(sc != msg, axis=1).sum()


@pytest.mark.parametrize("L", [8, 4, 2])
@pytest.mark.parametrize("N,K", [(64, 32), (128, 64), (256, 128), (512, 256), (1024, 512), (1024, 300), (256, 200)])
def test_scl_register_kernel_equals_lds_kernel(monkeypatch, N, K, L):
    """The register-resident kernel (list sizes 8, 4 and 2; the smaller ones leave list positions empty) and the
    LDS-staged kernel run the same fp32 arithmetic in the same order (f, g, metric, candidate ranking): decoded bits
    must be identical, frame by frame."""
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + K)
    B = 1501
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.0 / 10))
    llr = code.channel_llr(x, n0, 11, 0, np.float32)  # Eb/N0 = 1 dB: plenty of close decisions
    a, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel"), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), int(np.any(a != b, axis=1).sum())


@pytest.mark.parametrize("snr", [1.0, 2.0])
def test_scl8_headline_kernel_vs_fp32_oracle(monkeypatch, snr):
    """(1024,512), L = 8, fp32 -- the configuration the list-decoding throughput is quoted on -- straight against the fp32
    oracle, frame by frame.  The two differ only in exp2 / log2 of the branch metric (hardware v_exp_f32 / v_log_f32 against
    the host's exp2f / log2f).  That difference is measured here (final path metrics of the LDS-staged kernel, which runs
    the same arithmetic and also returns them, against the oracle's), and every frame whose closest pair of competing
    metrics is further apart than twice the largest difference seen (a decoder whose metrics are off by less than half
    a gap makes the same choices) must decode to the oracle's bits; the frames inside that margin are counted and may
    differ, but few of them do."""
    N, K, L, B = 1024, 512, 8, 2048
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(int(snr * 10))
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (snr / 10))
    llr = code.channel_llr(x, n0, 2024 + int(snr), 0, np.float32)
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    got, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel<10>/L8"), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    lds, pm, uh = code.scl_decode(llr, L, want_list=True)
    assert np.array_equal(got, lds)
    ref = np.empty_like(got)
    gap = np.empty(B)
    err = 0.0
    for b in range(B):
        ruh, rpm, _ = oracle.scl_decode(llr[b], mask, L)
        gap[b] = oracle.scl_last_min_gap()
        ref[b] = oracle.extract(ruh[0], mask)
        if np.array_equal(uh[b], ruh):  # same list in the same order: the metrics are comparable entry by entry
            fin = np.isfinite(rpm)
            assert np.array_equal(np.isfinite(pm[b]), fin)
            err = max(err, float(np.max(np.abs(pm[b][fin] - rpm[fin]) / np.maximum(np.abs(rpm[fin]), 1.0), initial=0.0)))
    margin = 2.0 * err
    same = np.all(got == ref, axis=1)
    inside = gap <= margin
    print(f"\nSCL-8 (1024,512) {snr} dB: device-vs-host metric difference {err:.3g} (relative), margin {margin:.3g}; "
          f"{int(inside.sum())} of {B} frames inside it, {int((~same).sum())} frames decoded differently")
    assert err < 1e-5
    assert same[~inside].all(), np.flatnonzero(~same & ~inside)[:10]
    assert (~same).sum() <= 0.01 * B


@pytest.mark.parametrize("snr", [1.0, 2.0])
def test_scl32_rate_profiled_512_kernel_vs_fp32_oracle(monkeypatch, snr):
    """The rate-profiled (512,256) code with L = 32 in fp32 -- the configuration that stresses sort and prune -- straight
    against the fp32 oracle, frame by frame, with the margin argument of the test above: the device-vs-host difference of the
    metrics is measured on the lists the LDS-staged kernel returns, every frame whose closest pair of competing metrics is
    further apart than twice that must decode to the oracle's bits on the two-lanes-per-path register kernel."""
    N, K, L, B = 512, 256, 32, 768
    mask = pcf.rmPolarBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(100 + int(snr * 10))
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (snr / 10))
    llr = code.channel_llr(x, n0, 77 + int(snr), 0, np.float32)
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    got, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel<9>/L32"), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    lds, pm, uh = code.scl_decode(llr, L, want_list=True)
    assert np.array_equal(got, lds)
    ref = np.empty_like(got)
    gap = np.empty(B)
    err = 0.0
    for b in range(B):
        ruh, rpm, _ = oracle.scl_decode(llr[b], mask, L)
        gap[b] = oracle.scl_last_min_gap()
        ref[b] = oracle.extract(ruh[0], mask)
        if np.array_equal(uh[b], ruh):
            fin = np.isfinite(rpm)
            assert np.array_equal(np.isfinite(pm[b]), fin)
            err = max(err, float(np.max(np.abs(pm[b][fin] - rpm[fin]) / np.maximum(np.abs(rpm[fin]), 1.0), initial=0.0)))
    margin = 2.0 * err
    same = np.all(got == ref, axis=1)
    inside = gap <= margin
    print(f"\nSCL-32 rmpolar (512,256) {snr} dB: device-vs-host metric difference {err:.3g} (relative), margin {margin:.3g}; "
          f"{int(inside.sum())} of {B} frames inside it, {int((~same).sum())} frames decoded differently")
    assert err < 1e-5
    assert same[~inside].all(), np.flatnonzero(~same & ~inside)[:10]
    assert (~same).sum() <= 0.02 * B


@pytest.mark.parametrize("L", [32, 16])
@pytest.mark.parametrize("cons,N,K", [("dega", 64, 32), ("dega", 128, 64), ("dega", 256, 128), ("dega", 512, 256), ("rmpolar", 512, 256),
                                      ("dega", 512, 150), ("dega", 256, 200), ("dega", 1024, 512), ("dega", 1024, 820)])
def test_scl_wide_kernel_equals_lds_kernel(monkeypatch, cons, N, K, L):
    """The two-lanes-per-path register kernel for lists of 16 and 32 paths against the LDS-staged kernel, which runs the
    same fp32 arithmetic in the same order: identical decoded bits, frame by frame, at an SNR with plenty of close
    decisions (and, at the short lengths, exactly tied metrics)."""
    mask = pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + K + L)
    B = 1201
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.0 / 10))
    llr = code.channel_llr(x, n0, 13, 0, np.float32)
    llr[:40] = np.round(llr[:40])  # integer LLRs: equal candidate metrics happen
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    a, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel"), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, _ = code.scl_decode(llr, L)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), np.flatnonzero(np.any(a != b, axis=1))[:20]


@pytest.mark.parametrize("L", [8, 4, 2])
@pytest.mark.parametrize("N,K", [(64, 32), (128, 64), (256, 128), (512, 256), (1024, 512), (1024, 300), (256, 200)])
def test_scl_fp64_register_kernel_equals_lds_kernel(monkeypatch, N, K, L):
    """fp64, the reference's arithmetic: the register-resident kernel against the LDS-staged kernel, whole sorted list.
    Both evaluate f, g and the metric expression in the same order, so decoded bits, every survivor and every path metric
    must be identical to the last bit -- including frames with integer LLRs, where candidate metrics tie exactly."""
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + K + L)
    B = 701
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.0 / 10))
    llr = code.channel_llr(x, n0, 17, 0, np.float64)
    llr[:40] = np.round(llr[:40])
    llr[40:44] *= 200.0  # metrics of wrong decisions overflow to -inf
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    a, apm, auh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel") and "double" in code.last_launch()["kernel"], code.last_launch()
    a2, _, _ = code.scl_decode(llr, L)  # without the list outputs
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, bpm, buh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), np.flatnonzero(np.any(a != b, axis=1))[:20]
    assert np.array_equal(a2, b)
    assert np.array_equal(auh, buh), np.flatnonzero(np.any(auh != buh, axis=(1, 2)))[:20]
    assert np.array_equal(apm, bpm), np.flatnonzero(np.any(apm != bpm, axis=1))[:20]


@pytest.mark.parametrize("L", [32, 16])
@pytest.mark.parametrize("cons,N,K", [("dega", 64, 32), ("dega", 128, 64), ("dega", 256, 128), ("rmpolar", 512, 256), ("dega", 512, 150),
                                      ("dega", 1024, 512)])
def test_scl_fp64_wide_kernel_equals_lds_kernel(monkeypatch, cons, N, K, L):
    """fp64 with lists of 16 and 32 paths (what PolarCode.sclDecoder runs by default at those list sizes): the
    two-lanes-per-path register kernel against the LDS-staged kernel, whole sorted list.  Same operations in the same
    order: decoded bits, every survivor and every metric identical to the last bit, including frames with integer LLRs
    (exactly tied candidate metrics) and frames whose wrong decisions overflow to -inf."""
    mask = pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + K + L + 1)
    B = 301
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.0 / 10))
    llr = code.channel_llr(x, n0, 19, 0, np.float64)
    llr[:40] = np.round(llr[:40])
    llr[40:44] *= 200.0
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    a, apm, auh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel") and "double" in code.last_launch()["kernel"], code.last_launch()
    a2, _, _ = code.scl_decode(llr, L)  # without the list outputs
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, bpm, buh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), np.flatnonzero(np.any(a != b, axis=1))[:20]
    assert np.array_equal(a2, b)
    assert np.array_equal(auh, buh), np.flatnonzero(np.any(auh != buh, axis=(1, 2)))[:20]
    assert np.array_equal(apm, bpm), np.flatnonzero(np.any(apm != bpm, axis=1))[:20]


@pytest.mark.parametrize("N,L,dtype", [(2048, 8, np.float32), (2048, 16, np.float32), (2048, 8, np.float64), (2048, 16, np.float64),
                                       (4096, 8, np.float32), (4096, 8, np.float64), (2048, 4, np.float32), (2048, 2, np.float64),
                                       (4096, 4, np.float64), (4096, 2, np.float32)])
def test_scl_wide_kernel_beyond_1024_equals_lds_kernel(monkeypatch, N, L, dtype):
    """N = 2048 and 4096, beyond the eight-lanes-per-path kernel: lists of 2 to 16 on the two-lanes-per-path kernel against
    the LDS-staged kernel (same operations in the same order): identical bits in fp32, identical bits, survivors and metrics
    in fp64, frames with exactly tied metrics and with overflowing metrics included."""
    K = N // 2
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + L)
    B = (401 if dtype is np.float32 else 151) * 2048 // N
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.0 / 10))
    llr = code.channel_llr(x, n0, 23, 0, dtype)
    llr[:30] = np.round(llr[:30])
    llr[30:33] *= 200.0
    lists = dtype is np.float64
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    a, apm, auh = code.scl_decode(llr, L, want_list=lists)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel<%d" % (11 if N == 2048 else 12)), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, bpm, buh = code.scl_decode(llr, L, want_list=lists)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), np.flatnonzero(np.any(a != b, axis=1))[:20]
    if lists:
        assert np.array_equal(auh, buh), np.flatnonzero(np.any(auh != buh, axis=(1, 2)))[:20]
        assert np.array_equal(apm, bpm), np.flatnonzero(np.any(apm != bpm, axis=1))[:20]


@pytest.mark.parametrize("N,L", [(2048, 32), (4096, 16), (4096, 32)])
def test_scl_wide_kernel_beyond_1024_vs_oracle(N, L):
    """List sizes the LDS-staged kernel has no room for at N = 2048 and 4096: fp64 against the CPU oracle (the reference's
    arithmetic): decoded bits, the whole sorted list and its metrics, frame by frame; fp32 decodes the same information bits
    on these frames."""
    K = N // 2
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(7)
    B = 6
    nl = 11 if N == 2048 else 12
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    n0 = 1.0 / (K / N * 10 ** (1.5 / 10))
    llr = code.channel_llr(x, n0, 29, 0, np.float64)
    info, pm, uh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel<%d,double>/L%d" % (nl, L)), code.last_launch()
    info32, _, _ = code.scl_decode(llr.astype(np.float32), L)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel<%d>/L%d" % (nl, L)), code.last_launch()
    for f in range(B):
        ouh, opm, _ = oracle.scl_decode(llr[f], mask, L)
        assert np.array_equal(uh[f][:len(ouh)], ouh), f
        # the oracle takes exp2 / log2 from the host's libm, the GPU from its device library: after 1024 information leaves a
        # last-bit difference in one of them is allowed (the shorter codes' tests compare the GPU kernels with each other bit by bit)
        np.testing.assert_allclose(pm[f][:len(opm)], opm, rtol=1e-13)
        assert np.array_equal(info[f], ouh[0][np.asarray(mask) == 1]), f
    assert np.array_equal(info32, info)


@pytest.mark.parametrize("N,K,L", [(64, 32, 8), (256, 128, 4), (512, 256, 8), (1024, 512, 8), (1024, 512, 2), (128, 64, 32), (512, 256, 32),
                                   (1024, 512, 16), (2048, 1024, 8), (2048, 1024, 16)])
def test_scl_fp32_lists_of_the_register_kernels_equal_lds_kernel(monkeypatch, N, K, L):
    """fp32 with the whole sorted list and its metrics as outputs: both register kernels against the LDS-staged kernel --
    bits, every survivor and every fp32 metric identical, frames with exactly tied and with overflowing metrics included."""
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    rs = np.random.RandomState(N + L)
    B = 400
    x = oracle.polar_encode(rs.randint(0, 2, size=(B, K)), mask).reshape(B, N)
    llr = code.channel_llr(x, 1.0 / (K / N * 10 ** 0.1), 31, 0, np.float32)
    llr[:40] = np.round(llr[:40])
    llr[40:44] *= 200.0
    monkeypatch.delenv("POLAR_SCL_KERNEL", raising=False)
    a, apm, auh = code.scl_decode(llr, L, want_list=True)
    want = "scl_reg_kernel" if (L <= 8 and N <= 1024) else "scl_wide_kernel"
    assert code.last_launch()["kernel"].startswith(want), code.last_launch()
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, bpm, buh = code.scl_decode(llr, L, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), np.flatnonzero(np.any(a != b, axis=1))[:20]
    assert np.array_equal(auh, buh), np.flatnonzero(np.any(auh != buh, axis=(1, 2)))[:20]
    assert np.array_equal(apm, bpm), np.flatnonzero(np.any(apm != bpm, axis=1))[:20]
