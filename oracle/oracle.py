"""ctypes front-end of the CPU oracle (oracle/polar_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpolar_oracle.so")
_lib = None

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "polar_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_polar_encode.argtypes = [_u8p, _u8p, C.c_int, _u8p]
        L.orc_mul_matrix.argtypes = [_u8p, C.c_int]
        L.orc_sc_decode_f64.argtypes = [_f64p, _u8p, C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_sc_decode_f32.argtypes = [_f32p, _u8p, C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_scl_decode_f64.argtypes = [_f64p, _u8p, C.c_int, C.c_int, _u8p, _f64p, _i64p]
        L.orc_scl_decode_f32.argtypes = [_f32p, _u8p, C.c_int, C.c_int, _u8p, _f64p, _i64p]
        L.orc_scl_decode_f64.restype = C.c_int
        L.orc_scl_decode_f32.restype = C.c_int
        L.orc_sc_decode_batch_f32.argtypes = [_f32p, _u8p, C.c_int, C.c_int64, _u8p]
        L.orc_sc_decode_batch_f64.argtypes = [_f64p, _u8p, C.c_int, C.c_int64, _u8p]
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_gen_info.argtypes = [C.c_uint64, C.c_uint64, C.c_int, _u8p]
        L.orc_gen_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_int, _f64p]
        L.orc_gen_frame.argtypes = [C.c_uint64, C.c_uint64, _u8p, C.c_int, C.c_int, C.c_double, _u8p, _u8p, _f64p]
        L.orc_mc_run.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_uint64, C.c_uint64, C.c_int64, _u64p]
        for nm in ("orc_scl2_decode_f64", "orc_scl2_decode_f32"):
            getattr(L, nm).argtypes = [_f64p if nm.endswith("64") else _f32p, _u8p, C.c_int, C.c_int, _u8p, _f64p, _i64p]
            getattr(L, nm).restype = C.c_int
        L.orc_genie_select.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
        L.orc_mc_run_genie.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_uint64, C.c_uint64, C.c_int64, _u64p]
        _i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.orc_critical_set.argtypes = [_u8p, C.c_int, _i32p]
        L.orc_modify_critical_set.argtypes = [_u8p, C.c_int, C.c_int, _i32p]
        L.orc_pbf_decode_f64.argtypes = [_f64p, _u8p, C.c_int, C.c_uint64, C.c_int, _f64p, _u8p, _i32p]
        L.orc_pbf_decode_f32.argtypes = [_f32p, _u8p, C.c_int, C.c_uint64, C.c_int, _f64p, _u8p, _i32p]
        L.orc_crc_encode.argtypes = [_u8p, C.c_int, C.c_uint64, _u8p]
        L.orc_ca_select.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_uint64]
        L.orc_mc_run_crc.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
                                     C.c_uint64, C.c_uint64, C.c_int64, _u64p]
        _lib = L
    return _lib


def _mask(mask):
    return np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)


def polar_encode(info, mask):
    """info [K] or [B,K] of 0/1 -> codeword x [N] or [B,N] (uint8)."""
    m = _mask(mask)
    info = np.asarray(info)
    single = info.ndim == 1
    inf2 = np.ascontiguousarray(np.atleast_2d(info), dtype=np.uint8)
    out = np.empty((inf2.shape[0], m.size), dtype=np.uint8)
    for b in range(inf2.shape[0]):
        lib().orc_polar_encode(inf2[b], m, m.size, out[b])
    return out[0] if single else out


def mul_matrix(u):
    x = np.ascontiguousarray(u, dtype=np.uint8).copy()
    lib().orc_mul_matrix(x, x.size)
    return x


def sc_decode(llr, mask, want_llrs=False, flips=()):
    """Single frame.  Returns uhat [N] (and decision LLRs [N]); dtype of llr selects f32/f64."""
    m = _mask(mask)
    llr = np.ascontiguousarray(llr)
    N = m.size
    uh = np.empty(N, dtype=np.uint8)
    f32 = llr.dtype == np.float32
    if not f32:
        llr = llr.astype(np.float64, copy=False)
    dl = np.empty(N, dtype=llr.dtype) if want_llrs else None
    fl = np.ascontiguousarray(flips, dtype=np.int32)
    fn = lib().orc_sc_decode_f32 if f32 else lib().orc_sc_decode_f64
    fn(llr, m, N, uh, dl.ctypes.data if want_llrs else None, fl.ctypes.data if fl.size else None, fl.size)
    return (uh, dl) if want_llrs else uh


def sc_decode_batch(llr, mask):
    m = _mask(mask)
    llr = np.ascontiguousarray(llr)
    B, N = llr.shape
    uh = np.empty((B, N), dtype=np.uint8)
    if llr.dtype == np.float32:
        lib().orc_sc_decode_batch_f32(llr, m, N, B, uh)
    else:
        lib().orc_sc_decode_batch_f64(llr.astype(np.float64, copy=False), m, N, B, uh)
    return uh


def scl_decode(llr, mask, L):
    """Single frame.  Returns (uhat_sorted [n,N], pm_sorted [n], counts [3]); n = final list size."""
    m = _mask(mask)
    llr = np.ascontiguousarray(llr)
    N = m.size
    uh = np.zeros((L, N), dtype=np.uint8)
    pm = np.zeros(L, dtype=np.float64)
    cnt = np.zeros(3, dtype=np.int64)
    if llr.dtype == np.float32:
        n = lib().orc_scl_decode_f32(llr, m, N, L, uh, pm, cnt)
    else:
        n = lib().orc_scl_decode_f64(llr.astype(np.float64, copy=False), m, N, L, uh, pm, cnt)
    return uh[:n], pm[:n], cnt


def scl_last_min_gap():
    """Smallest relative gap between neighbouring candidate metrics in the prunes and the final sort of the last
    scl_decode call of this thread (inf if there was none): how small a metric error could have changed a choice."""
    fn = lib().orc_scl_last_min_gap
    fn.restype = C.c_double
    return float(fn())


def extract(uhat, mask):
    return np.asarray(uhat)[..., np.asarray(mask) != 0]


def philox(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


def gen_frame(seed, frame, mask, N0):
    m = _mask(mask)
    N, K = m.size, int(m.sum())
    info = np.empty(K, dtype=np.uint8)
    x = np.empty(N, dtype=np.uint8)
    llr = np.empty(N, dtype=np.float64)
    lib().orc_gen_frame(seed, frame, m, N, K, float(N0), info, x, llr)
    return info, x, llr


def gen_noise(seed, frame, N):
    z = np.empty(N, dtype=np.float64)
    lib().orc_gen_noise(seed, frame, N, z)
    return z


def mc_run(mask, decoder, L, dtype, N0, seed, frame0, nframes):
    """decoder 0=SC 1=SCL; dtype 0=f32 1=f64.  Returns counters [frames, frame errors, bit errors]."""
    m = _mask(mask)
    c = np.zeros(3, dtype=np.uint64)
    lib().orc_mc_run(m, m.size, int(m.sum()), decoder, L, dtype, float(N0), seed, frame0, nframes, c)
    return c


def crc_encode(message, crc_poly):
    """message ++ CRC (crc_poly with its leading term, e.g. 0x107)."""
    msg = np.ascontiguousarray(message, dtype=np.uint8)
    w = int(crc_poly).bit_length() - 1
    out = np.empty(msg.size + w, dtype=np.uint8)
    lib().orc_crc_encode(msg, msg.size, int(crc_poly), out)
    return out


def ca_select(uhat_sorted, mask, crc_poly):
    """Row of the sorted list that CRC-aided decoding returns."""
    m = _mask(mask)
    uh = np.ascontiguousarray(uhat_sorted, dtype=np.uint8)
    return lib().orc_ca_select(uh, m, m.size, uh.shape[0], int(crc_poly))


def ca_scl_decode(llr, mask, L, crc_poly):
    """Single frame, CRC-aided: decoded message bits (K - width)."""
    uh, pm, _ = scl_decode(llr, mask, L)
    j = ca_select(uh, mask, crc_poly)
    w = int(crc_poly).bit_length() - 1
    word = extract(uh[j], mask)
    return word[:word.size - w]


def mc_run_crc(mask, L, dtype, N0, crc_poly, seed, frame0, nframes):
    m = _mask(mask)
    c = np.zeros(3, dtype=np.uint64)
    lib().orc_mc_run_crc(m, m.size, int(m.sum()), L, dtype, float(N0), int(crc_poly), seed, frame0, nframes, c)
    return c


def scl2_decode(llr, mask, L):
    """Single frame through the in-place-fork list decoder (polarSCLFork2).  Same returns as scl_decode."""
    m = _mask(mask)
    llr = np.ascontiguousarray(llr)
    N = m.size
    uh = np.zeros((L, N), dtype=np.uint8)
    pm = np.zeros(L, dtype=np.float64)
    cnt = np.zeros(3, dtype=np.int64)
    if llr.dtype == np.float32:
        n = lib().orc_scl2_decode_f32(llr, m, N, L, uh, pm, cnt)
    else:
        n = lib().orc_scl2_decode_f64(llr.astype(np.float64, copy=False), m, N, L, uh, pm, cnt)
    return uh[:n], pm[:n], cnt


def genie_select(uhat_sorted, u_true):
    uh = np.ascontiguousarray(uhat_sorted, dtype=np.uint8)
    ut = np.ascontiguousarray(u_true, dtype=np.uint8)
    return lib().orc_genie_select(uh, uh.shape[0], uh.shape[1], ut)


def mc_run_genie(mask, L, dtype, N0, seed, frame0, nframes):
    m = _mask(mask)
    c = np.zeros(3, dtype=np.uint64)
    lib().orc_mc_run_genie(m, m.size, int(m.sum()), L, dtype, float(N0), seed, frame0, nframes, c)
    return c


def critical_set(frozen):
    fz = np.ascontiguousarray(np.asarray(frozen) != 0, dtype=np.uint8)
    out = np.empty(fz.size, dtype=np.int32)
    n = lib().orc_critical_set(fz, fz.size, out)
    return out[:n].copy()


def modify_critical_set(mask, i_max):
    m = _mask(mask)
    out = np.empty(m.size, dtype=np.int32)
    n = lib().orc_modify_critical_set(m, m.size, int(i_max), out)
    return out[:n].copy()


def pbf_decode(llr, mask, crc_poly, max_level, mu):
    """Single frame through progressive bit flipping.  Returns (information word [K] of the last SC run,
    number of SC runs, flips of the last run)."""
    m = _mask(mask)
    llr = np.ascontiguousarray(llr)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    word = np.empty(int(m.sum()), dtype=np.uint8)
    last = np.empty(max_level, dtype=np.int32)
    if llr.dtype == np.float32:
        runs = lib().orc_pbf_decode_f32(llr, m, m.size, int(crc_poly), max_level, mu, word, last)
    else:
        runs = lib().orc_pbf_decode_f64(llr.astype(np.float64, copy=False), m, m.size, int(crc_poly), max_level, mu, word, last)
    return word, runs, last
