"""How often the schedule with short-cuts differs from the every-leaf schedule in fp32 (exact zeros inside
all-information subtrees): frames with any differing information bit, (1024,512) at 2 dB."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np, torch
import _native, functions as pcf
ctx = _native.default_ctx(0); dev = torch.device("cuda:0")
N, K = 1024, 512
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0)); code.specialize()
B = 1 << 22
llr = torch.empty((B, N), dtype=torch.float32, device=dev)
info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
a = torch.empty_like(info); b = torch.empty_like(info)
tot = diff = 0
for chunk in range(4):
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, 1.0 / (K / N * 10 ** 0.2), 77, chunk * B, B)
    code.sc_decode_dev(llr.data_ptr(), np.float32, a.data_ptr(), B, fast=True)
    code.sc_decode_dev(llr.data_ptr(), np.float32, b.data_ptr(), B, fast=False)
    ctx.sync()
    d = (a != b).any(dim=1)
    tot += B; diff += int(d.sum().item())
    print("frames", tot, "differing", diff, flush=True)
