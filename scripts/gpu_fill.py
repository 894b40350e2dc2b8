"""HBM store rate reference: torch fill of the bench's LLR buffer (41 GB) next to the frame generator."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd")]
import numpy as np, torch
import _native, functions as pcf
ctx = _native.default_ctx(0); dev = torch.device("cuda:0")
N, K, B = 1024, 512, 10 ** 7
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
llr = torch.empty((B, N), dtype=torch.float32, device=dev)
info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
for name, fn in (("fill_", lambda: llr.fill_(1.0)), ("gen_frames", lambda: code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, 1.0, 1, 0, B)),
                 ("copy_", lambda: llr[: B // 2].copy_(llr[B // 2:]))):
    for _ in range(2):
        torch.cuda.synchronize(); ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt*1e3:.2f} ms, {llr.numel()*4/dt/1e12:.2f} TB/s of LLR bytes")
