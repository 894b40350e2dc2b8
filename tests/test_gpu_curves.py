"""fp32 Monte-Carlo BER/FER of the GPU path against the reference's own curve (reference Python run on
seeded frames, counts stored in tests/golden/curves.npz with frames, frame errors, bit errors and the sum
of squared per-frame bit errors).

Acceptance: every GPU point must lie inside the 95 % confidence interval of the reference point -- Wilson
interval for FER, normal interval from the per-frame bit-error variance for BER, each widened by the (much
smaller) sampling error of the GPU estimate.  With 17 independent points one excursion beyond 95 % is
expected by chance, so at most one point may sit between the 95 % and the 99.9 % interval; none beyond.
"""
import numpy as np
import pytest

import _native
import functions as pcf
from channel import Channel

pytestmark = pytest.mark.gpu

PLANS = [  # tag, N, K, decoder, L, GPU frames per point
    ("sc_128_64", 128, 64, _native.DEC_SC, 1, 2_000_000),
    ("sc_1024_512", 1024, 512, _native.DEC_SC, 1, 2_000_000),
    ("scl8_1024_512", 1024, 512, _native.DEC_SCL, 8, 200_000),
    ("scl4_128_64", 128, 64, _native.DEC_SCL, 4, 400_000),
]


def wilson(k, n, z):
    p = k / n
    d = 1 + z * z / n
    c = (p + z * z / (2 * n)) / d
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / d
    return c - h, c + h


def zscores(tag, N, K, dec, L, frames):
    curves = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "curves.npz"))
    ref = curves[tag]
    code = _native.Code(_native.default_ctx(), pcf.degaBuild(N, K, 2.0))
    out = []
    for i, (snr, nfr, fe, be, be2) in enumerate(ref):
        n0 = Channel("BPSK", float(snr), "SNRb", K / N).noisePower
        c = code.mc_run(dec, L, np.float32, n0, 777 + i, 0, frames)
        g_fer, g_ber = int(c[1]) / frames, int(c[2]) / (frames * K)
        r_fer, r_ber = fe / nfr, be / (nfr * K)
        # BER: mean of per-frame bit errors e/K; variance from sum e^2
        var_e = max(be2 / nfr - (be / nfr) ** 2, 0.0) / nfr
        sd_ber = np.sqrt(var_e) / K
        sd_ber_g = sd_ber * np.sqrt(nfr / frames)
        out.append(dict(snr=float(snr), g_fer=g_fer, r_fer=r_fer, g_ber=g_ber, r_ber=r_ber, nfr=int(nfr), fe=int(fe),
                        sd_fer_g=np.sqrt(max(g_fer * (1 - g_fer), 1e-12) / frames), sd_ber=sd_ber, sd_ber_g=sd_ber_g))
    return out


def inside(pt, z):
    lo, hi = wilson(pt["fe"], pt["nfr"], z)
    ok_fer = lo - z * pt["sd_fer_g"] <= pt["g_fer"] <= hi + z * pt["sd_fer_g"]
    ok_ber = abs(pt["g_ber"] - pt["r_ber"]) <= z * np.hypot(pt["sd_ber"], pt["sd_ber_g"]) + 1e-12
    return ok_fer, ok_ber


def test_fp32_curves_inside_reference_confidence_intervals():
    beyond95, report = 0, []
    for plan in PLANS:
        for pt in zscores(*plan):
            f95, b95 = inside(pt, 1.96)
            f999, b999 = inside(pt, 3.29)
            report.append((plan[0], pt["snr"], pt["g_fer"], pt["r_fer"], pt["g_ber"], pt["r_ber"], f95, b95))
            assert f999 and b999, (plan[0], pt)
            beyond95 += (not f95) + (not b95)
    for r in report:
        print("%-14s %.1f dB  FER gpu %.4e ref %.4e | BER gpu %.4e ref %.4e | in95 %s %s" % r)
    assert beyond95 <= 1, report


def test_simulate_sweep_on_the_gpu_matches_mc_run():
    """simulate() end to end on the GPU (single rank): fixed frame count, specialised decoder, same counters as
    direct polar_mc_run calls with the same seeds."""
    import simulator
    from polarCode import PolarCode

    pc = PolarCode(256, 128, "dega", 2.0)
    pc.decoderType, pc.snrType, pc.modu = "sc", "SNRb", "BPSK"
    res = simulator.simulate(pc, [1.0, 2.0], frames=50000, seed=11, verbose=False, batch=1 << 14)
    assert pc.native.last_launch()["kernel"].startswith("sc_static_mc_kernel")  # one kernel from the random bits to the counters
    ref = _native.Code(_native.default_ctx(), pc.polarMask)
    for i, snr in enumerate([1.0, 2.0]):
        n0 = Channel("BPSK", snr, "SNRb", 0.5).noisePower
        c = ref.mc_run(_native.DEC_SC, 1, np.float32, n0, 11 + i, 0, 50000)
        assert [res.frames[i], res.frameErrors[i], res.bitErrors[i]] == c.tolist()
    assert res.fer[0] > res.fer[1] > 0
