"""Pins the CPU oracle (oracle/polar_oracle.c) to vectors produced by running the reference."""
import hashlib

import numpy as np
import pytest

from frames import make_frame, pack, unpack
from oracle import oracle


def _mask(golden, tag, N):
    m = golden.masks[tag]
    return m if N <= 8 else unpack(m, N)


@pytest.mark.parametrize("N,K", [(8, 4), (128, 64), (512, 256), (1024, 512), (16384, 8192)])
def test_encoder_matches_reference(golden, N, K):
    g = golden.encoder
    mask = _mask(golden, f"dega_{N}_{K}_2.0", N)
    info = unpack(g[f"info_{N}_{K}"], K)
    x = oracle.polar_encode(info, mask)
    assert np.array_equal(pack(x), g[f"x_{N}_{K}"])
    u = unpack(g[f"u_{N}"], N)
    xu = np.stack([oracle.mul_matrix(r) for r in u])
    assert np.array_equal(pack(xu), g[f"xu_{N}"])


def test_encoder_kat_8_4():
    # message 1011 on the (8,4) dega mask 00010111 -> codeword 10100101
    x = oracle.polar_encode([1, 0, 1, 1], [0, 0, 0, 1, 0, 1, 1, 1])
    assert x.tolist() == [1, 0, 1, 0, 0, 1, 0, 1]


def test_sc_kat_8_4(golden):
    g = golden.sc
    uh, dl = oracle.sc_decode(g["kat8_llr"], [0, 0, 0, 1, 0, 1, 1, 1], want_llrs=True)
    assert np.array_equal(uh, g["kat8_uhat"])
    assert dl.tobytes() == g["kat8_decllr"].tobytes()
    assert dl.tolist() == [1, -3, 1.5, -3.5, -2.5, -2, -6, -31.1]


SC_TAGS = [("dega", 128, 64, 1.0), ("dega", 128, 64, 2.0), ("dega", 128, 64, 4.0), ("dega", 1024, 512, 1.0),
           ("dega", 1024, 512, 2.0), ("dega", 1024, 512, 3.0), ("rmpolar", 512, 256, 2.0), ("dega", 16384, 8192, 2.0)]


@pytest.mark.parametrize("cons,N,K,snr", SC_TAGS)
def test_sc_matches_reference(golden, cons, N, K, snr):
    g = golden.sc
    tag = f"{cons}_{N}_{K}_{snr}"
    mask = _mask(golden, f"{cons}_{N}_{K}_2.0", N)
    seeds = g[f"{tag}_seeds"]
    full_llr, full_dec = g[f"{tag}_llr"], g[f"{tag}_decllr"]
    for f, seed in enumerate(seeds):
        msg, x, llr = make_frame(mask, snr, int(seed), oracle.polar_encode)
        if f < len(full_llr):  # stored inputs guard the seeded regeneration itself
            assert llr.tobytes() == full_llr[f].tobytes()
        uh, dl = oracle.sc_decode(llr, mask, want_llrs=True)
        assert np.array_equal(pack(uh), g[f"{tag}_uhat"][f]), (tag, f)
        assert hashlib.sha1(dl.tobytes()).digest() == g[f"{tag}_decllr_sha1"][f].tobytes(), (tag, f)
        if f < len(full_dec):
            assert dl.tobytes() == full_dec[f].tobytes()
        assert int(np.sum(oracle.extract(uh, mask) != msg)) == int(g[f"{tag}_biterr"][f])


def test_scl_kat_8_4(golden):
    g = golden.scl
    uh, pm, _ = oracle.scl_decode(g["kat8_llr"], [0, 0, 0, 1, 0, 1, 1, 1], 4)
    assert np.array_equal(oracle.extract(uh[0], [0, 0, 0, 1, 0, 1, 1, 1]), g["kat8_dec"])
    np.testing.assert_allclose(pm, g["kat8_pm"], rtol=1e-13)
    np.testing.assert_allclose(pm, [-1.0908631565524596, -3.9727056749214777, -7.787037005675798, -9.229732048891263], rtol=1e-13)


SCL_TAGS = [("dega", 128, 64, 2, 0.0), ("dega", 128, 64, 2, 2.0), ("dega", 128, 64, 4, 0.0), ("dega", 128, 64, 4, 2.0),
            ("dega", 128, 64, 8, 0.0), ("dega", 128, 64, 16, 0.0), ("dega", 128, 64, 32, 0.0),
            ("dega", 1024, 512, 8, 1.0), ("dega", 1024, 512, 8, 2.0),
            ("rmpolar", 512, 256, 32, 2.0), ("dega", 512, 256, 32, 1.0), ("dega", 512, 256, 32, 2.0)]


@pytest.mark.parametrize("cons,N,K,L,snr", SCL_TAGS)
def test_scl_matches_reference(golden, cons, N, K, L, snr):
    g = golden.scl
    tag = f"{cons}_{N}_{K}_L{L}_{snr}"
    mask = _mask(golden, f"{cons}_{N}_{K}_2.0", N)
    for f, seed in enumerate(g[f"{tag}_seeds"]):
        msg, x, llr = make_frame(mask, snr, int(seed), oracle.polar_encode)
        uh, pm, cnt = oracle.scl_decode(llr, mask, L)
        assert g[f"{tag}_gap"][f] > 1e-9  # no fixture frame is tie-sensitive
        assert np.array_equal(pack(uh), g[f"{tag}_uhat"][f]), (tag, f)  # whole sorted list, in order
        assert np.array_equal(pack(oracle.extract(uh[0], mask)), g[f"{tag}_dec"][f])
        ref = g[f"{tag}_pm"][f]
        assert np.array_equal(np.isinf(pm), np.isinf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(pm[fin], ref[fin], rtol=1e-12)
        assert cnt.tolist() == g[f"{tag}_counts"][f].tolist()  # ANV, sortNum, unitCal


def test_scl_seeded_kat_1024(golden):
    """Reference-side numbers quoted for np.random.seed(2024), (1024,512), L=8, 2 dB."""
    mask = unpack(golden.masks["dega_1024_512_2.0"], 1024)
    msg, x, llr = make_frame(mask, 2.0, 2024, oracle.polar_encode)
    uh, pm, cnt = oracle.scl_decode(llr, mask, 8)
    assert abs(pm[0] - 600.3603000267871) < 1e-9
    assert cnt.tolist() == [6788, 510, 64900]


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2).tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_noise_is_standard_normal():
    z = np.concatenate([oracle.gen_noise(7, f, 1024) for f in range(200)])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    from scipy import stats
    assert stats.kstest(z, "norm").pvalue > 1e-3
