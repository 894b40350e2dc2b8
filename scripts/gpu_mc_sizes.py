"""One-kernel Monte-Carlo loop (polar_mc_run on a specialised code) at four code lengths: frames per second and counters."""
import sys, os, time
sys.path[:0] = ['.', 'polar-codes_amd']
import numpy as np
import _native, functions as pcf
ctx = _native.default_ctx(0)
for N, K, nf in [(128, 64, 1 << 28), (256, 128, 1 << 27), (512, 256, 1 << 27), (1024, 512, 1 << 26)]:
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0)); code.specialize()
    n0 = 1.0 / (K / N * 10 ** 0.2)
    code.mc_run(_native.DEC_SC, 1, np.float32, n0, 7, 0, 1 << 20)
    t0 = time.perf_counter(); c = code.mc_run(_native.DEC_SC, 1, np.float32, n0, 7, 0, nf); dt = time.perf_counter() - t0
    print(f"({N},{K}) {nf} frames: {nf/dt/1e6:.1f} M frames/s = {nf*K/dt/1e9:.1f} Gb/s {c.tolist()} {code.last_launch()['kernel']} grid {code.last_launch()['grid_blocks']}", flush=True)
