"""ctypes binding of libpolar_hip.so (the C ABI declared in include/polar_hip.h).

There is no CPU fallback: if the library is missing or no GPU is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POLAR_HIP_LIB") or os.path.join(_HERE, "libpolar_hip.so")  # POLAR_HIP_LIB: an instrumented build

HOST, DEVICE = 0, 1
F32, F64 = 0, 1
DEC_SC, DEC_SCL, DEC_PBF = 0, 1, 2
SC_FAST = 1
CRC = 2
GENIE = 4

_lib = None
_default_ctx = None


class PolarError(RuntimeError):
    pass


class LaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("lanes_per_frame", C.c_int), ("lds_bytes_per_block", C.c_int),
                ("block_threads", C.c_int), ("grid_blocks", C.c_int), ("last_ms", C.c_float)]


# every symbol include/polar_hip.h declares, with its ctypes signature
_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
SIGNATURES = {
    "polar_last_error": (C.c_char_p, []),
    "polar_version": (C.c_char_p, []),
    "polar_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "polar_ctx_destroy": (_i, [_vp]),
    "polar_ctx_set_stream": (_i, [_vp, _vp]),
    "polar_ctx_sync": (_i, [_vp]),
    "polar_ctx_num_cus": (_i, [_vp]),
    "polar_ctx_enable_timing": (_i, [_vp, _i]),
    "polar_code_create": (_i, [_vp, _i, _vp, C.POINTER(_vp)]),
    "polar_code_destroy": (_i, [_vp]),
    "polar_code_n": (_i, [_vp]),
    "polar_code_k": (_i, [_vp]),
    "polar_code_schedule_len": (_i, [_vp, _i]),
    "polar_code_specialize": (_i, [_vp]),
    "polar_static_compile_check": (_i, [_vp, _i, C.c_char_p]),
    "polar_code_set_crc": (_i, [_vp, _u64]),
    "polar_code_crc_width": (_i, [_vp]),
    "polar_schedule_build": (_i, [_vp, _i, _i, _vp, _i]),
    "polar_output_tables": (_i, [_vp, _i, _vp]),
    "polar_encode": (_i, [_vp, _vp, _vp, _i64, _i]),
    "polar_encode_packed": (_i, [_vp, _vp, _vp, _i64, _i]),
    "polar_mul_matrix": (_i, [_vp, _vp, _vp, _i64, _i]),
    "polar_channel_llr": (_i, [_vp, _vp, _vp, _i, _d, _u64, _u64, _i64, _i]),
    "polar_gen_frames": (_i, [_vp, _vp, _vp, _i, _d, _u64, _u64, _i64, _i]),
    "polar_sc_decode": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _i]),
    "polar_sc_decode_packed": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _i]),
    "polar_scl_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i, _i]),
    "polar_scl_decode_packed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i, _i]),
    "polar_scl_decode_genie": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i]),
    "polar_scl_decode_genie_packed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i]),
    "polar_critical_set": (_i, [_vp, _i, _i, _vp, _i]),
    "polar_dispatch_select": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "polar_dispatch_table": (_i, [_vp, _i]),
    "polar_code_set_reliability": (_i, [_vp, _vp]),
    "polar_sc_decode_flip": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i]),
    "polar_pbf_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i]),
    "polar_pbf_decode_packed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i]),
    "polar_pbf_decode_noalloc": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64]),
    "polar_sc_decode_count": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _i, _vp]),
    "polar_count_errors": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "polar_mc_run": (_i, [_vp, _i, _i, _i, _i, _d, _u64, _u64, _i64, _vp]),
    "polar_comm_init_all": (_i, [_i, _vp, C.POINTER(_vp)]),
    "polar_comm_destroy": (_i, [_vp]),
    "polar_comm_size": (_i, [_vp]),
    "polar_allreduce_counters": (_i, [_vp, _vp, _i]),
    "polar_last_launch": (_i, [_vp, C.POINTER(LaunchInfo)]),
}


def lib():
    """Loads libpolar_hip.so; raises PolarError if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PolarError("libpolar_hip.so not found at %s: build it first (make -C polar-codes_amd/csrc)" % LIB_PATH)
        try:
            # PyTorch ships its own copy of the HIP/HSA runtime.  A process can only work with one: if this
            # library pulled in the system runtime first, torch.cuda would later find no GPU.  Loading torch
            # first makes both use the same runtime, in whichever order the caller touches the GPU.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("POLAR_HIP_LIB") and not hasattr(L, name):
                continue  # an older build loaded for a side-by-side measurement lacks the newest entry points
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise PolarError("polar_hip error %d: %s" % (rc, lib().polar_last_error().decode()))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)  # raw device pointer (e.g. torch tensor.data_ptr())


def _dt(dtype):
    return F64 if np.dtype(dtype) == np.float64 else F32


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().polar_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().polar_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().polar_ctx_sync(self._h))

    def set_stream(self, stream_ptr):
        _check(lib().polar_ctx_set_stream(self._h, stream_ptr))

    def enable_timing(self, on=True):
        _check(lib().polar_ctx_enable_timing(self._h, 1 if on else 0))

    @property
    def num_cus(self):
        return lib().polar_ctx_num_cus(self._h)


class Code:
    """Device-side code object (mask tables + decoder schedules)."""

    def __init__(self, ctx, mask):
        self.ctx = ctx
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        self.N = int(m.size)
        self._h = C.c_void_p()
        _check(lib().polar_code_create(ctx._h, self.N, m.ctypes.data, C.byref(self._h)))
        self.K = lib().polar_code_k(self._h)
        self.Nw = (self.N + 31) // 32
        self.Kw = max(1, (self.K + 31) // 32)
        self.crc_width = 0

    def close(self):
        if self._h:
            lib().polar_code_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def schedule_len(self, fast):
        return lib().polar_code_schedule_len(self._h, 1 if fast else 0)

    def specialize(self):
        """Compile (hiprtc, a few seconds) the SC decoder unrolled for this rate profile; fast fp32 SC decoding
        of this code uses it from then on."""
        _check(lib().polar_code_specialize(self._h))

    def set_crc(self, crc_poly):
        """Attach (or with 0 remove) the CRC generator polynomial, leading term included (0x107 = CRC-8)."""
        _check(lib().polar_code_set_crc(self._h, int(crc_poly)))
        self.crc_width = lib().polar_code_crc_width(self._h)

    def set_reliability(self, mu):
        """Mean LLRs of the bit channels (functions.GA) that rank the flip candidates."""
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        assert mu.size == self.N
        _check(lib().polar_code_set_reliability(self._h, mu.ctypes.data))

    def last_launch(self):
        li = LaunchInfo()
        _check(lib().polar_last_launch(self._h, C.byref(li)))
        return {"kernel": li.kernel.decode(), "lanes_per_frame": li.lanes_per_frame,
                "lds_bytes_per_block": li.lds_bytes_per_block, "block_threads": li.block_threads,
                "grid_blocks": li.grid_blocks, "last_ms": li.last_ms}

    # ---- host-array conveniences (numpy in, numpy out) -------------------------------------------
    def encode(self, info):
        info = np.ascontiguousarray(np.atleast_2d(info), dtype=np.uint8)
        assert info.shape[1] == self.K, "info must have K columns"
        x = np.empty((info.shape[0], self.N), dtype=np.uint8)
        _check(lib().polar_encode(self._h, info.ctypes.data, x.ctypes.data, info.shape[0], HOST))
        return x

    def mul_matrix(self, u):
        u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.uint8)
        assert u.shape[1] == self.N
        x = np.empty_like(u)
        _check(lib().polar_mul_matrix(self._h, u.ctypes.data, x.ctypes.data, u.shape[0], HOST))
        return x

    def channel_llr(self, x, N0, seed=2024, frame0=0, dtype=np.float32):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.uint8)
        assert x.shape[1] == self.N
        llr = np.empty(x.shape, dtype=np.float64 if _dt(dtype) == F64 else np.float32)
        _check(lib().polar_channel_llr(self._h, x.ctypes.data, llr.ctypes.data, _dt(dtype), float(N0), seed, frame0,
                                       x.shape[0], HOST))
        return llr

    def sc_decode(self, llr, want_uhat=False, want_llrs=False, fast=False):
        llr = np.ascontiguousarray(np.atleast_2d(llr))
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        B = llr.shape[0]
        assert llr.shape[1] == self.N
        info = np.empty((B, self.K), dtype=np.uint8)
        uh = np.empty((B, self.N), dtype=np.uint8) if want_uhat else None
        dl = np.empty((B, self.N), dtype=llr.dtype) if want_llrs else None
        _check(lib().polar_sc_decode(self._h, llr.ctypes.data, _dt(llr.dtype), info.ctypes.data, _ptr(uh), _ptr(dl), B,
                                     SC_FAST if fast else 0, HOST))
        return info, uh, dl

    def scl_decode(self, llr, L, want_list=False, crc=False):
        llr = np.ascontiguousarray(np.atleast_2d(llr))
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        B = llr.shape[0]
        assert llr.shape[1] == self.N
        info = np.empty((B, self.K), dtype=np.uint8)
        pm = np.empty((B, L), dtype=np.float64) if want_list else None
        uh = np.empty((B, L, self.N), dtype=np.uint8) if want_list else None
        _check(lib().polar_scl_decode(self._h, llr.ctypes.data, _dt(llr.dtype), L, info.ctypes.data, _ptr(pm), _ptr(uh), B,
                                      CRC if crc else 0, HOST))
        return info, pm, uh

    def scl_decode_genie(self, llr, L, info_true, want_list=False):
        """List decoding with the genie's choice: the list entry equal to info_true [B, K] if there is one."""
        llr = np.ascontiguousarray(np.atleast_2d(llr))
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        B = llr.shape[0]
        truth = np.ascontiguousarray(np.atleast_2d(info_true), dtype=np.uint8)
        assert llr.shape[1] == self.N and truth.shape == (B, self.K)
        info = np.empty((B, self.K), dtype=np.uint8)
        pm = np.empty((B, L), dtype=np.float64) if want_list else None
        uh = np.empty((B, L, self.N), dtype=np.uint8) if want_list else None
        _check(lib().polar_scl_decode_genie(self._h, llr.ctypes.data, _dt(llr.dtype), L, truth.ctypes.data, info.ctypes.data,
                                            _ptr(pm), _ptr(uh), B, HOST))
        return info, pm, uh

    def sc_decode_flip(self, llr, flips, want_uhat=True, want_llrs=True):
        """SC decoding with inverted decisions: flips [B, <=3] positions per frame, -1 = unused."""
        llr = np.ascontiguousarray(np.atleast_2d(llr))
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        B = llr.shape[0]
        fl = np.full((B, 3), -1, dtype=np.int32)
        flips = np.atleast_2d(np.asarray(flips, dtype=np.int32))
        if flips.size:
            assert flips.shape[1] <= 3, "at most three flips per frame"
            fl[:, :flips.shape[1]] = flips
        info = np.empty((B, self.K), dtype=np.uint8)
        uh = np.empty((B, self.N), dtype=np.uint8) if want_uhat else None
        dl = np.empty((B, self.N), dtype=llr.dtype) if want_llrs else None
        _check(lib().polar_sc_decode_flip(self._h, llr.ctypes.data, _dt(llr.dtype), fl.ctypes.data, info.ctypes.data, _ptr(uh),
                                          _ptr(dl), B, HOST))
        return info, uh, dl

    def pbf_decode(self, llr, max_level):
        """Progressive bit flipping.  Returns (info [B, K], runs [B], crc_ok [B], last_seq [B, 3])."""
        llr = np.ascontiguousarray(np.atleast_2d(llr))
        if llr.dtype not in (np.float32, np.float64):
            llr = llr.astype(np.float64)
        B = llr.shape[0]
        assert llr.shape[1] == self.N
        info = np.empty((B, self.K), dtype=np.uint8)
        runs = np.empty(B, dtype=np.int32)
        ok = np.empty(B, dtype=np.uint8)
        last = np.empty((B, 3), dtype=np.int32)
        _check(lib().polar_pbf_decode(self._h, llr.ctypes.data, _dt(llr.dtype), int(max_level), info.ctypes.data, runs.ctypes.data,
                                      ok.ctypes.data, last.ctypes.data, B, HOST))
        return info, runs, ok, last

    def pbf_decode_dev(self, llr_ptr, dtype, max_level, info_hat_ptr, B, runs_ptr=None, ok_ptr=None, last_ptr=None):
        _check(lib().polar_pbf_decode_packed(self._h, llr_ptr, _dt(dtype), int(max_level), info_hat_ptr, runs_ptr, ok_ptr, last_ptr,
                                             B, DEVICE))

    def pbf_decode_noalloc_dev(self, llr_ptr, dtype, max_level, info_hat_ptr, B, runs_ptr=None, ok_ptr=None, last_ptr=None):
        """pbf_decode_dev for callers that cannot allocate: at most 2^20 frames per call, N <= 1024."""
        _check(lib().polar_pbf_decode_noalloc(self._h, llr_ptr, _dt(dtype), int(max_level), info_hat_ptr, runs_ptr, ok_ptr, last_ptr, B))

    def mc_run(self, decoder, L, dtype, N0, seed, frame0, nframes, fast=True, crc=False, genie=False):
        cnt = np.zeros(3, dtype=np.uint64)
        flags = (SC_FAST if (fast and decoder == DEC_SC) else 0) | (CRC if crc else 0) | (GENIE if genie else 0)
        _check(lib().polar_mc_run(self._h, decoder, L, _dt(dtype), flags, float(N0),
                                  seed, frame0, nframes, cnt.ctypes.data))
        return cnt

    # ---- raw device-pointer entry points (used by bench.py / simulator with torch tensors) --------
    def gen_frames_dev(self, info_ptr, llr_ptr, dtype, N0, seed, frame0, B, crc=False):
        _check(lib().polar_gen_frames(self._h, info_ptr, llr_ptr, _dt(dtype), float(N0), seed, frame0, B, CRC if crc else 0))

    def sc_decode_dev(self, llr_ptr, dtype, info_hat_ptr, B, fast=True, u_hat_ptr=None, dec_llr_ptr=None):
        _check(lib().polar_sc_decode_packed(self._h, llr_ptr, _dt(dtype), info_hat_ptr, u_hat_ptr, dec_llr_ptr, B,
                                            SC_FAST if fast else 0, DEVICE))

    def sc_decode_count_dev(self, llr_ptr, dtype, info_ptr, B, counters_ptr, fast=True, info_hat_ptr=None):
        """SC decode and error count in one pass (device buffers); the decoded words are stored only if asked for."""
        _check(lib().polar_sc_decode_count(self._h, llr_ptr, _dt(dtype), info_ptr, info_hat_ptr, B, SC_FAST if fast else 0,
                                           counters_ptr))

    def scl_decode_dev(self, llr_ptr, dtype, L, info_hat_ptr, B, pm_ptr=None, u_hat_ptr=None, crc=False):
        _check(lib().polar_scl_decode_packed(self._h, llr_ptr, _dt(dtype), L, info_hat_ptr, pm_ptr, u_hat_ptr, B,
                                             CRC if crc else 0, DEVICE))

    def scl_decode_genie_dev(self, llr_ptr, dtype, L, info_true_ptr, info_hat_ptr, B, pm_ptr=None, u_hat_ptr=None):
        _check(lib().polar_scl_decode_genie_packed(self._h, llr_ptr, _dt(dtype), L, info_true_ptr, info_hat_ptr, pm_ptr,
                                                   u_hat_ptr, B, DEVICE))

    def count_errors_dev(self, info_ptr, hat_ptr, B, counters_ptr, crc=False):
        _check(lib().polar_count_errors(self._h, info_ptr, hat_ptr, B, CRC if crc else 0, counters_ptr))


class Comm:
    """RCCL communicators over GPUs driven from this one process (polar_comm_init_all)."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        _check(lib().polar_comm_init_all(len(devices), devs, C.byref(self._h)))
        self.devices = list(devices)

    def allreduce_counters(self, dev_ptrs, n=3):
        """Sum n uint64 counters over the devices, in place; dev_ptrs[i] = device pointer on devices[i]."""
        arr = (C.c_void_p * len(dev_ptrs))(*[int(p) for p in dev_ptrs])
        _check(lib().polar_allreduce_counters(self._h, arr, int(n)))

    def close(self):
        if self._h:
            lib().polar_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def schedule_build(mask, fast, sub16=False):
    """Decoder schedule for a rate profile as a uint32 array (host only, no GPU needed)."""
    m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    fast = (1 if fast else 0) | (2 if sub16 else 0)
    n = lib().polar_schedule_build(m.ctypes.data, m.size, fast, None, 0)
    if n < 0:
        _check(n)
    out = np.empty(n, dtype=np.uint32)
    _check(min(0, lib().polar_schedule_build(m.ctypes.data, m.size, fast, out.ctypes.data, n)))
    return out


def output_tables(mask):
    """The 256-word compress / place table of the register SC kernels' output stage (host only, no GPU needed)."""
    m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    out = np.empty(256, dtype=np.uint32)
    _check(lib().polar_output_tables(m.ctypes.data, m.size, out.ctypes.data))
    return out


def static_compile_check(mask, arch="gfx950"):
    """Host only: run the run-time compiler on the decoder specialised for `mask`; returns the code size."""
    m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    n = lib().polar_static_compile_check(m.ctypes.data, m.size, arch.encode())
    if n < 0:
        _check(n)
    return n


def critical_set(mask, after=-1):
    """First positions of the maximal all-information subtrees of `mask` restricted to indices > after (host only)."""
    m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    out = np.empty(m.size, dtype=np.int32)
    n = lib().polar_critical_set(m.ctypes.data, m.size, int(after), out.ctypes.data, out.size)
    if n < 0:
        _check(n)
    return out[:n].copy()


# ---- which kernel serves a call (host only: the library's static dispatch table) ------------------------------------
OP_SC, OP_SCL, OP_PBF, OP_MC_SC = 0, 1, 2, 3
OUT_INFO, OUT_UHAT, OUT_DECLLR, OUT_PM = 1, 2, 4, 8
DQ_FAST, DQ_SPECIALISED, DQ_CRC, DQ_GENIE, DQ_FLIP_LIST, DQ_SPLIT, DQ_FORCE_LDS, DQ_FORCE_GENERIC = 1, 2, 4, 8, 16, 32, 64, 128


def dispatch_select(op, N, K, dtype, L=0, outputs=OUT_INFO, flags=0):
    """(row of the dispatch table, kernel name) that serves the call; raises PolarError for a call no kernel covers."""
    buf = C.create_string_buffer(96)
    row = lib().polar_dispatch_select(int(op), int(N), int(K), _dt(dtype), int(L), int(outputs), int(flags), buf, len(buf))
    if row < 0:
        _check(row)
    return row, buf.value.decode()


def dispatch_table():
    """The dispatch table as a list of dicts (one per row, in the order the launchers read it)."""
    need = lib().polar_dispatch_table(None, 0)
    buf = C.create_string_buffer(need)
    lib().polar_dispatch_table(buf, need)
    lines = buf.value.decode().strip().split("\n")
    head = lines[0].split("\t")
    return [dict(zip(head, ln.split("\t"))) for ln in lines[1:]]


def default_ctx(device=None):
    """Process-wide context on `device` (default: POLAR_DEVICE, else LOCAL_RANK, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        if device is None:
            device = int(os.environ.get("POLAR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(device)
    return _default_ctx


_scratch = {}


def scratch_code(N):
    """A rate-1 code object of length N on the default context (for calls that only need N)."""
    if N not in _scratch:
        _scratch[N] = Code(default_ctx(), np.ones(N, dtype=np.uint8))
    return _scratch[N]
