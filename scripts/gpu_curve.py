"""BER / FER curve of (1024,512) SC in fp32 from the one-kernel Monte-Carlo loop, 10^9 frames per point, next to the
reference's own points (tests/golden/curves.npz: counts from the reference Python decoder) where they exist."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np
import _native, functions as pcf
from channel import Channel

N, K, frames = 1024, 512, 10 ** 9
ctx = _native.default_ctx(0)
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0)); code.specialize()
ref = {round(float(r[0]), 2): r for r in np.load(os.path.join(ROOT, "tests", "golden", "curves.npz"))["sc_1024_512"]}
print(f"Polar({N},{K}) dega dSNR 2.0, SC, fp32, {frames:.0e} frames per point, kernel sc_static_mc_kernel")
print("Eb/N0 dB | frame errors | FER | BER | seconds | reference FER (frames) | reference BER")
for i, snr in enumerate(np.arange(1.0, 4.01, 0.25)):
    n0 = Channel("BPSK", float(snr), "SNRb", K / N).noisePower
    t0 = time.perf_counter()
    c = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 4242 + i, 0, frames)
    dt = time.perf_counter() - t0
    r = ref.get(round(float(snr), 2))
    rs = f"{r[2] / r[1]:.4e} ({int(r[1])}) | {r[3] / (r[1] * K):.4e}" if r is not None else "- | -"
    print(f"{snr:.2f} | {int(c[1])} | {int(c[1]) / frames:.4e} | {int(c[2]) / (frames * K):.4e} | {dt:.2f} | {rs}", flush=True)
