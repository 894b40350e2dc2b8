"""(1024,512) SC on 10^7 resident frames: one launch against the same frames in chunks (kernel time from the events)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np, torch
import _native, functions as pcf

ctx = _native.default_ctx(0)
dev = torch.device("cuda:0")
N, K = 1024, 512
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
B = 10_000_000
llr = torch.empty((B, N), dtype=torch.float32, device=dev)
info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
hat = torch.empty_like(info)
code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, 1.0 / (K / N * 10 ** 0.2), 1, 0, B)
code.specialize()
ctx.enable_timing(True)
for chunk in (B, 5_000_000, 2_500_000, 1_000_000, 500_000):
    best = 1e9
    for _ in range(3):
        tot = 0.0
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            code.sc_decode_dev(llr.data_ptr() + 4 * N * b0, np.float32, hat.data_ptr() + 4 * code.Kw * b0, nb, fast=True)
            tot += code.last_launch()["last_ms"]
        best = min(best, tot)
    print(f"chunk {chunk}: {best:.3f} ms for {B} frames = {B / best / 1e3:.0f} M frames/s", flush=True)
