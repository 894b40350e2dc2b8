"""BER / FER curve of (1024,512) list decoding, L = 8, fp32, through polar_mc_run (generator -> buffer -> register list
decoder -> count), 10^8 frames per point, next to the reference's own points (tests/golden/curves.npz: counts from the
reference Python decoder) with their 95 % Wilson intervals."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np

import _native
import functions as pcf
from channel import Channel


def wilson(k, n, z=1.96):
    p = k / n
    c = (p + z * z / (2 * n)) / (1 + z * z / n)
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / (1 + z * z / n)
    return c - h, c + h


N, K, L = 1024, 512, 8
frames = int(float(os.environ.get("FRAMES", "1e8")))
ctx = _native.default_ctx(0)
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
ref = {round(float(r[0]), 2): r for r in np.load(os.path.join(ROOT, "tests", "golden", "curves.npz"))["scl8_1024_512"]}
print(f"Polar({N},{K}) dega dSNR 2.0, SCL L = {L}, fp32, {frames:.0e} frames per point")
print("Eb/N0 dB | frame errors | FER | BER | seconds | M frames/s | reference FER (frames) [95 % interval] | inside")
for i, snr in enumerate(np.arange(1.0, 3.01, 0.25)):
    n0 = Channel("BPSK", float(snr), "SNRb", K / N).noisePower
    t0 = time.perf_counter()
    c = code.mc_run(_native.DEC_SCL, L, np.float32, n0, 777 + i, 0, frames)
    dt = time.perf_counter() - t0
    fer = int(c[1]) / frames
    r = ref.get(round(float(snr), 2))
    if r is not None:
        lo, hi = wilson(r[2], r[1])
        rs = f"{r[2] / r[1]:.4e} ({int(r[1])}) [{lo:.3e}, {hi:.3e}] | {'yes' if lo <= fer <= hi else 'NO'}"
    else:
        rs = "- | -"
    print(f"{snr:.2f} | {int(c[1])} | {fer:.4e} | {int(c[2]) / (frames * K):.4e} | {dt:.2f} | {frames / dt / 1e6:.2f} | {rs}", flush=True)
print("kernel:", code.last_launch()["kernel"])
