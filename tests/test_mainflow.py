"""The bit-reversed flow of the reference's stand-alone script cn/main.py (vectors captured by executing it
under a numpy seed).  CPU part: the flow equals natural-order SC with the bit-reversed rate profile (oracle).
GPU part: PolarCode(bitReversed=True) reproduces the script's codeword, uHat, decoded bits and decision LLRs."""
import numpy as np
import pytest

import functions as pcf
from frames import unpack
from oracle import oracle

N, K = 1024, 512


def _load(golden):
    g = golden.mainflow
    mask = unpack(g["mask"], N)
    rev = np.array([pcf.bitreversed(i, 10) for i in range(N)])
    return g, mask, rev


def test_main_script_mask_is_dega_2p5(golden):
    g, mask, rev = _load(golden)
    assert np.array_equal(mask, pcf.degaBuild(N, K, 2.5))


def test_oracle_with_reversed_profile_reproduces_main_script(golden):
    g, mask, rev = _load(golden)
    mask_v = mask[rev]
    for s in range(1, 9):
        msg = unpack(g[f"s{s}_message"], K)
        u = np.zeros(N, dtype=np.uint8)
        u[mask == 1] = msg
        assert np.array_equal(oracle.mul_matrix(u[rev]), unpack(g[f"s{s}_x"], N))
        vhat, dl = oracle.sc_decode(g[f"s{s}_llr"], mask_v, want_llrs=True)
        uhat = vhat[rev]
        assert np.array_equal(uhat, unpack(g[f"s{s}_uhat"], N))
        assert np.array_equal(uhat[mask == 1], unpack(g[f"s{s}_decoded"], K))
        assert dl.tobytes() == g[f"s{s}_decllr"].tobytes()


@pytest.mark.gpu
def test_gpu_bit_reversed_flow_matches_main_script(golden):
    from polarCode import PolarCode

    g, mask, rev = _load(golden)
    pc = PolarCode(N, K, "dega", 2.5)
    pc.bitReversed = True
    msgs = np.stack([unpack(g[f"s{s}_message"], K) for s in range(1, 9)])
    x = pc.polarEncode(msgs)
    assert np.array_equal(x, np.stack([unpack(g[f"s{s}_x"], N) for s in range(1, 9)]))
    llr = np.stack([g[f"s{s}_llr"] for s in range(1, 9)])
    dec = pc.scDecoder(llr)
    for i, s in enumerate(range(1, 9)):
        assert np.array_equal(pc.uHat[i], unpack(g[f"s{s}_uhat"], N))
        assert np.array_equal(dec[i], unpack(g[f"s{s}_decoded"], K))
        assert np.array_equal(pc.llrs[i], g[f"s{s}_decllr"])
    # natural ordering is untouched by the flag
    pc.bitReversed = False
    assert np.array_equal(pc.polarEncode(msgs[0]), oracle.polar_encode(msgs[0], mask))


@pytest.mark.gpu
def test_gpu_main_cli_runs(capsys):
    import main

    res = main.run(["--N", "128", "--K", "64", "--snr", "2", "4", "--frames", "2000", "--seed", "3", "--bitreversed"])
    assert len(res) == 2 and res[0][1] > res[1][1] >= 0
    main.run(["--N", "64", "--K", "32", "--snr", "3", "--seed", "1"])
    out = capsys.readouterr().out
    assert "decoded =" in out and "uHat =" in out
