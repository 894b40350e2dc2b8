"""SC throughput of the generic and the specialised register kernels for several code lengths and rates."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np, torch
import _native, functions as pcf

ctx = _native.default_ctx(0)
dev = torch.device("cuda:0")
for N, K in [(128, 64), (256, 128), (512, 256), (1024, 256), (1024, 512), (1024, 768)]:
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(ctx, mask)
    B = (1 << 32) // (4 * N)
    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
    hat = torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, 1.0 / (K / N * 10 ** 0.2), 1, 0, B)
    ctx.enable_timing(True)
    out = []
    for spec in (False, True):
        if spec:
            code.specialize()
        ms = []
        for _ in range(3):
            code.sc_decode_dev(llr.data_ptr(), np.float32, hat.data_ptr(), B, fast=True)
            ms.append(code.last_launch()["last_ms"])
        out.append((code.last_launch()["kernel"], B / (min(ms) * 1e-3)))
    ctx.enable_timing(False)
    print(f"({N},{K}) B={B}: " + "; ".join(f"{k}: {v/1e6:.0f} M frames/s = {v*K/1e9:.0f} Gb/s" for k, v in out), flush=True)
    del llr, info, hat
