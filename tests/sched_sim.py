"""Reference interpreter of the decoder schedule (one frame, numpy) used to check the host-side schedule
compiler without a GPU.  Mirrors the operation semantics of the SC kernel."""
import numpy as np

OP_F, OP_G, OP_LEAF_INFO, OP_LEAF_FROZEN, OP_R0, OP_R1, OP_COMBINE, OP_SUB16, OP_END = 0, 1, 2, 3, 4, 5, 6, 7, 15


def sc_node(llr, mask):
    """Partial sums of a node by plain recursive SC (f = min-sum, g, combine)."""
    if llr.size == 1:
        return np.array([1 if (mask[0] and llr[0] < 0) else 0], dtype=np.uint8)
    h = llr.size // 2
    a, b = llr[:h], llr[h:]
    xl = sc_node((np.sign(a) * np.sign(b) * np.minimum(np.abs(a), np.abs(b))).astype(llr.dtype), mask[:h])
    xr = sc_node(np.where(xl == 1, b - a, b + a).astype(llr.dtype), mask[h:])
    return np.concatenate([xl ^ xr, xr])


def decode_op(w):
    w = int(w)
    return w & 15, (w >> 4) & 31, (w >> 9) & 1, w >> 12


def run_schedule(ops, llr, N):
    n = int(np.log2(N))
    dt = llr.dtype
    lv = {n: np.asarray(llr).copy()}
    ps = [dict(), dict()]  # plane -> level -> bit array
    dec_llr = np.full(N, np.nan, dtype=dt)
    for w in ops:
        t, lev, plane, leaf = decode_op(w)
        if t == OP_END:
            break
        if t == OP_F:
            h = 1 << (lev - 1)
            a, b = lv[lev][:h], lv[lev][h:]
            lv[lev - 1] = (np.sign(a) * np.sign(b) * np.minimum(np.abs(a), np.abs(b))).astype(dt)
        elif t == OP_G:
            h = 1 << (lev - 1)
            a, b = lv[lev][:h], lv[lev][h:]
            bits = ps[0][lev - 1]
            lv[lev - 1] = np.where(bits == 1, b - a, b + a).astype(dt)
        elif t in (OP_LEAF_INFO, OP_LEAF_FROZEN):
            v = lv[0][0]
            dec_llr[leaf] = v
            ps[plane][0] = np.array([1 if (t == OP_LEAF_INFO and v < 0) else 0], dtype=np.uint8)
        elif t == OP_SUB16:
            ps[plane][4] = sc_node(lv[4], [(leaf >> i) & 1 for i in range(16)])
        elif t == OP_R0:
            ps[plane][lev] = np.zeros(1 << lev, dtype=np.uint8)
        elif t == OP_R1:
            ps[plane][lev] = (lv[lev] < 0).astype(np.uint8)
        elif t == OP_COMBINE:
            a, b = ps[0][lev - 1], ps[1][lev - 1]
            ps[plane][lev] = np.concatenate([a ^ b, b])
    x = ps[0][n]
    u = x.copy()
    for s in range(n):
        d = 1 << s
        idx = np.arange(N)
        lo = idx[(idx & d) == 0]
        u[lo] ^= u[lo + d]
    return u, dec_llr
