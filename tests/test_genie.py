"""Genie-aided list decoding: sclDecoderOracle + polarSCLFork2, the decoder cn/simulator.py:156 calls.

tests/golden/genie.npz (tools/gen_golden.py, section `genie`) holds frames decoded by the reference's
sclDecoderOracle with pcode.message set as in cn/simulator.py:140-141: decoded bits, the whole sorted list with
its metrics, the list index of the transmitted word (-1: not in the list) and the complexity counters.
"""
import numpy as np
import pytest

import functions as pcf
from frames import make_frame, n0_ebn0, pack
from oracle import oracle

CFGS = [("dega", 128, 64, 2, 0.0), ("dega", 128, 64, 4, 0.0), ("dega", 128, 64, 8, 1.0), ("dega", 256, 128, 16, 0.5),
        ("dega", 1024, 512, 8, 1.0), ("rmpolar", 512, 256, 32, 1.5)]


def build_mask(cons, N, K):
    return pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)


@pytest.mark.parametrize("cons,N,K,L,snr", CFGS)
def test_oracle_fork2_and_genie_match_reference_vectors(golden, cons, N, K, L, snr):
    g = golden.genie
    tag = f"{cons}_{N}_{K}_L{L}_{snr}"
    mask = build_mask(cons, N, K)
    deeper = 0
    for f, seed in enumerate(g[f"{tag}_seeds"]):
        msg, x, llr = make_frame(mask, snr, int(seed), oracle.polar_encode)
        uh, pm, cnt = oracle.scl2_decode(llr, mask, L)
        assert np.array_equal(pack(uh), g[f"{tag}_uhat"][f]), (tag, f)
        np.testing.assert_allclose(pm, g[f"{tag}_pm"][f], rtol=1e-12)
        assert cnt.tolist() == g[f"{tag}_counts"][f].tolist()
        ut = np.zeros(N, dtype=np.uint8)
        ut[mask == 1] = msg
        j = oracle.genie_select(uh, ut)
        at = int(g[f"{tag}_truth_at"][f])
        assert j == max(at, 0)
        dec = uh[j][mask == 1]
        assert np.array_equal(pack(dec), g[f"{tag}_dec"][f])
        assert int(np.sum(dec != msg)) == int(g[f"{tag}_biterr"][f])
        deeper += at > 0
        # the in-place fork and the sorting fork produce the same sorted list when no two metrics are equal
        uh1, pm1, _ = oracle.scl_decode(llr, mask, L)
        assert np.array_equal(uh, uh1) and np.array_equal(pm, pm1)
    assert deeper > 0 or L == 32


def test_oracle_fork2_tie_rule():
    """Exact ties separate the two forks: all-equal channel LLRs give equal metrics to many candidates.
    polarSCLFork2 prefers u = 0 and the lower slot; polarSCLFork prefers the copies (u = 1)."""
    mask = np.array([0, 1, 1, 1], dtype=np.uint8)
    llr = np.zeros(4) + 1e-300  # decision LLRs of +-1e-300 or 0: every branch metric is exactly 0
    uh2, pm2, _ = oracle.scl2_decode(llr, mask, 2)
    uh1, pm1, _ = oracle.scl_decode(llr, mask, 2)
    assert pm2.tolist() == pm1.tolist() == [0.0, 0.0]
    assert uh2.tolist() == [[0, 0, 0, 0], [0, 1, 0, 0]]
    assert uh1.tolist() != uh2.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("cons,N,K,L,snr", CFGS)
def test_gpu_genie_fp64_matches_reference_vectors(golden, cons, N, K, L, snr):
    from polarCode import PolarCode

    g = golden.genie
    tag = f"{cons}_{N}_{K}_L{L}_{snr}"
    pc = PolarCode(N, K, cons, 2.0, listSize=L)
    seeds = g[f"{tag}_seeds"]
    fr = [make_frame(pc.polarMask, snr, int(s), oracle.polar_encode) for s in seeds]
    msgs = np.stack([f[0] for f in fr])
    llr = np.stack([f[2] for f in fr])
    pc.message = np.zeros((len(seeds), N), dtype=int)
    pc.message[:, pc.polarMask == 1] = msgs
    dec = pc.sclDecoderOracle(llr)
    assert np.array_equal(pack(dec), g[f"{tag}_dec"])
    assert np.array_equal(np.sum(dec != msgs, axis=1), g[f"{tag}_biterr"])
    for f in range(len(seeds)):
        assert np.array_equal(pack(np.stack([p.uHat for p in pc.pathList[f]])), g[f"{tag}_uhat"][f])
        np.testing.assert_allclose(pc.pathMetrics[f], g[f"{tag}_pm"][f], rtol=1e-12)
    assert [pc.ANV, pc.sortNum, pc.unitCal] == g[f"{tag}_counts"][0].tolist()
    # one frame, reference-style call
    pc.message = pc.message[3]
    one = pc.sclDecoderOracle(llr[3])
    assert one.shape == (K,) and np.array_equal(one, dec[3])
    # a message that is not a codeword's u (ones on frozen positions) can never match: best path
    pc.message = np.ones(N, dtype=int)
    assert np.array_equal(pc.sclDecoderOracle(llr[3]), pc.sclDecoder(llr[3]))
    # without the genie the best path comes back
    plain = pc.sclDecoder(llr)
    assert np.array_equal(np.sum(plain != msgs, axis=1), g[f"{tag}_biterr_first_path"])


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,L,nfr", [(128, 64, 4, 3000), (256, 100, 8, 1500), (32, 3, 16, 500), (1024, 512, 8, 300), (512, 256, 32, 100)])
def test_gpu_mc_run_genie_fp64_equals_oracle(N, K, L, nfr):
    import _native

    mask = pcf.degaBuild(N, K, 2.0)
    n0 = n0_ebn0(1.0, K / N)
    code = _native.Code(_native.default_ctx(), mask)
    ref = oracle.mc_run_genie(mask, L, 1, n0, 31, 200, nfr)
    got = code.mc_run(_native.DEC_SCL, L, np.float64, n0, 31, 200, nfr, genie=True)
    plain = code.mc_run(_native.DEC_SCL, L, np.float64, n0, 31, 200, nfr)
    assert got.tolist() == ref.tolist()
    assert got[1] <= plain[1]
    if K > 8:
        assert got[1] < plain[1]


@pytest.mark.gpu
def test_gpu_genie_register_kernel_equals_lds_kernel(monkeypatch):
    import _native

    N, K = 1024, 512
    mask = pcf.degaBuild(N, K, 2.0)
    n0 = n0_ebn0(1.0, K / N)
    code = _native.Code(_native.default_ctx(), mask)
    B = 4000
    msg = np.random.RandomState(2).randint(0, 2, size=(B, K)).astype(np.uint8)
    llr = code.channel_llr(oracle.polar_encode(msg, mask), n0, 5, 0, np.float32)
    a, _, _ = code.scl_decode_genie(llr, 8, msg)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel"), code.last_launch()
    plain, _, _ = code.scl_decode(llr, 8)
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, uh = code.scl_decode_genie(llr, 8, msg, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b)
    words = uh[:, :, mask == 1]
    hit = (words == msg[:, None, :]).all(axis=2)
    pick = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
    assert np.array_equal(b, words[np.arange(B), pick])
    assert (pick > 0).any() and np.any(a != plain)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,L", [(512, 256, 32), (1024, 512, 16), (128, 64, 32)])
def test_gpu_genie_wide_kernel_equals_lds_kernel(monkeypatch, N, K, L):
    import _native

    mask = pcf.degaBuild(N, K, 2.0)
    n0 = n0_ebn0(0.5, K / N)
    code = _native.Code(_native.default_ctx(), mask)
    B = 1500
    msg = np.random.RandomState(2).randint(0, 2, size=(B, K)).astype(np.uint8)
    llr = code.channel_llr(oracle.polar_encode(msg, mask), n0, 5, 0, np.float32)
    a, _, _ = code.scl_decode_genie(llr, L, msg)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel"), code.last_launch()
    plain, _, _ = code.scl_decode(llr, L)
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, uh = code.scl_decode_genie(llr, L, msg, want_list=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b)
    words = uh[:, :, mask == 1]
    hit = (words == msg[:, None, :]).all(axis=2)
    pick = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
    assert np.array_equal(b, words[np.arange(B), pick])
    assert (pick > 0).any() and np.any(a != plain)


@pytest.mark.gpu
def test_gpu_genie_argument_errors():
    import _native

    code = _native.Code(_native.default_ctx(), pcf.degaBuild(64, 32, 2.0))
    code.set_crc(0x107)
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SCL, 4, np.float32, 1.0, 1, 0, 10, crc=True, genie=True)
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SC, 1, np.float32, 1.0, 1, 0, 10, genie=True)
    rc = _native.lib().polar_scl_decode_genie(code._h, None, 0, 4, None, None, None, None, 1, 0)
    assert rc == -1
