"""SC for N > 1024 (fp32): the slab kernel against the oracle and the LDS-staged kernel; frames per second of both."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polar-codes_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import _native, functions as pcf
from oracle import oracle
ctx = _native.default_ctx(0); dev = torch.device("cuda:0")
for N, K, nb in [(2048, 1000, 1 << 17), (4096, 2048, 1 << 16), (4096, 3000, 1 << 16), (16384, 8192, 1 << 15), (65536, 30000, 1 << 12)]:
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(ctx, mask)
    n0 = 1.0 / (K / N * 10 ** 0.2)
    llr = torch.empty((nb, N), dtype=torch.float32, device=dev)
    info = torch.empty((nb, code.Kw), dtype=torch.int32, device=dev); a = torch.empty_like(info); b = torch.empty_like(info)
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, n0, 5, 0, nb)
    ctx.enable_timing(True)
    res = {}
    for mode, out in (("slab", a), ("lds", b)):
        if mode == "lds": os.environ["POLAR_SC_KERNEL"] = "lds"
        else: os.environ.pop("POLAR_SC_KERNEL", None)
        for _ in range(2):
            code.sc_decode_dev(llr.data_ptr(), np.float32, out.data_ptr(), nb, fast=True)
        ctx.sync()
        res[mode] = (code.last_launch()["kernel"], nb / (code.last_launch()["last_ms"] * 1e-3))
    os.environ.pop("POLAR_SC_KERNEL", None)
    same = bool((a == b).all().item())
    # a few frames against the oracle
    h = llr[:3].cpu().numpy()
    ref = np.stack([oracle.extract(oracle.sc_decode(h[i], mask), mask) for i in range(3)])
    got = np.unpackbits(a[:3].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :K]
    print(f"({N},{K}) {nb} frames: {res['slab'][0]} {res['slab'][1]/1e6:.2f} M frames/s = {res['slab'][1]*K/1e9:.1f} Gb/s; {res['lds'][0]} {res['lds'][1]/1e6:.2f} M frames/s; "
          f"equal {same}; oracle {bool((got == ref).all())}", flush=True)
