import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np, torch
import _native, functions as pcf
ctx = _native.default_ctx(0); dev = torch.device("cuda:0")
N, K = 1024, 512
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0)); code.specialize()
B = 4000000
llr = torch.empty((B, N), dtype=torch.float32, device=dev)
info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev); hat = torch.empty_like(info)
code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, 1.0 / (K / N * 10 ** 0.2), 1, 0, B)
ctx.enable_timing(True)
for name, ptr in (("with output stage", hat.data_ptr()), ("tree walk only", None)):
    ms = []
    for _ in range(4):
        code.sc_decode_dev(llr.data_ptr(), np.float32, ptr, B, fast=True)
        ms.append(code.last_launch()["last_ms"])
    print(name, min(ms), "ms", B / min(ms) / 1e3, "M frames/s")
