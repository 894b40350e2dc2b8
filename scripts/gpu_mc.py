"""Monte-Carlo loop: the single fused kernel against generator -> buffer -> decoder (POLAR_MC_KERNEL=staged): counters
must be identical; frames per second of both."""
import sys, os, time
os.environ["POLAR_EXPERIMENT"] = "1"  # the library reads its kernel switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np
import _native, functions as pcf
ctx = _native.default_ctx(0)
for N, K, nf in [(128, 64, 1 << 20), (256, 100, 1 << 20), (512, 256, 1 << 20), (1024, 512, 1 << 22), (1024, 800, 1 << 20), (1024, 512, 1 << 24)]:
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0)); code.specialize()
    n0 = 1.0 / (K / N * 10 ** 0.2)
    out = {}
    for mode in ("fused", "staged"):
        if mode == "staged":
            os.environ["POLAR_MC_KERNEL"] = "staged"
        else:
            os.environ.pop("POLAR_MC_KERNEL", None)
        code.mc_run(_native.DEC_SC, 1, np.float32, n0, 7, 12345, 4096)
        t0 = time.perf_counter(); c = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 7, 12345, nf); dt = time.perf_counter() - t0
        out[mode] = (c.tolist(), nf / dt, code.last_launch()["kernel"])
    os.environ.pop("POLAR_MC_KERNEL", None)
    print(f"({N},{K}) {nf} frames: fused {out['fused'][0]} {out['fused'][1]/1e6:.1f} M frames/s [{out['fused'][2]}]; staged {out['staged'][0]} {out['staged'][1]/1e6:.1f} M frames/s; "
          + ("SAME" if out['fused'][0] == out['staged'][0] else "DIFFERENT"), flush=True)
