"""Rates of the decoders when the caller hands LLRs in host memory (POLAR_HOST): the batch crosses PCIe in chunks of
16384 frames, chunk i+1 on a second stream while chunk i is decoded.  Pageable numpy arrays and page-locked torch
tensors, packed outputs; the device-resident rate of the same call beside it."""
import ctypes
import os
import sys
import time

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "polar-codes_amd")]
import numpy as np
import torch

import _native
import functions as pcf
from channel import Channel

ctx = _native.default_ctx(0)
L_ = _native.lib()
N, K = 1024, 512
code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
code.specialize()
n0 = Channel("BPSK", 2.0, "SNRb", K / N).noisePower
nb = 1 << 18
for dt, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
    dev = torch.empty((nb, N), dtype=tdt, device="cuda")
    code.gen_frames_dev(None, dev.data_ptr(), dt, n0, 1, 0, nb)
    ctx.sync()
    pinned = torch.empty((nb, N), dtype=tdt, pin_memory=True)
    pinned.copy_(dev)
    pageable = pinned.numpy().copy()
    hat_dev = torch.empty((nb, code.Kw), dtype=torch.int32, device="cuda")
    hat_pin = torch.empty((nb, code.Kw), dtype=torch.int32, pin_memory=True)
    hat_pag = np.empty((nb, code.Kw), dtype=np.int32)
    legs = [("SC fast", lambda l, h, sp: L_.polar_sc_decode_packed(code._h, l, _native._dt(dt), h, None, None, nb, _native.SC_FAST, sp)),
            ("SCL L=8", lambda l, h, sp: L_.polar_scl_decode_packed(code._h, l, _native._dt(dt), 8, h, None, None, nb, 0, sp))]
    for name, call in legs:
        res = {}
        for kind, lp, hp, sp in (("device", dev.data_ptr(), hat_dev.data_ptr(), _native.DEVICE),
                                 ("pinned host", pinned.data_ptr(), hat_pin.data_ptr(), _native.HOST),
                                 ("pageable host", pageable.ctypes.data, hat_pag.ctypes.data, _native.HOST)):
            best = 1e9
            for _ in range(3):
                ctx.sync()
                t0 = time.perf_counter()
                _native._check(call(lp, hp, sp))
                ctx.sync()
                best = min(best, time.perf_counter() - t0)
            res[kind] = best
        assert np.array_equal(hat_pin.numpy(), hat_dev.cpu().numpy()) and np.array_equal(hat_pag, hat_dev.cpu().numpy())
        gb = nb * N * np.dtype(dt).itemsize / 1e9
        print("(%d,%d) %s %s, %d frames (%.2f GB of LLRs): " % (N, K, name, np.dtype(dt).name, nb, gb)
              + "; ".join("%s %.2f M frames/s (%.1f GB/s)" % (k, nb / v / 1e6, gb / v) for k, v in res.items()), flush=True)
    del dev, pinned, pageable
