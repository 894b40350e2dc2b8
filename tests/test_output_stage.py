"""The output stage of the register SC kernels, restated in numpy on the tables the library builds for it: u_hat on the
sliced layout (element k of the frame = bit k // 16 of lane k % 16) -> 16 x 16 bit transpose across the lanes -> information
bits of every 16-position piece compressed to its low end -> pieces OR-ed into the information words at their bit offsets.
Checks table builder and method against uHat[polarMask == 1] (cn/polarCode.py:305-312) for every length and several
rate profiles; the GPU tests then check the kernels against the oracle."""
import numpy as np
import pytest

import _native
import functions as pcf


def lane_constants():
    sub = np.arange(16)
    return [np.where(sub & d == 0, 0xFFFFFFFF, 0).astype(np.uint64) for d in (8, 4, 2, 1)]


def rot(t, amt):          # rotate right by amt (what v_alignbit_b32(t, t, amt) returns)
    amt = amt.astype(np.uint64) & 31
    return ((t >> amt) | (t << ((32 - amt) & 31))) & 0xFFFFFFFF


def row_transpose16(x, low):
    """x: uint64[16] holding 32-bit words, one per lane; the four exchange steps of the kernel."""
    sub = np.arange(16)
    # distance 8: byte permute
    t = x[sub ^ 8]
    b = lambda v, i: (v >> (8 * i)) & 0xFF
    lower = (b(x, 0) | (b(t, 0) << 8) | (b(x, 2) << 16) | (b(t, 2) << 24))
    upper = (b(t, 1) | (b(x, 1) << 8) | (b(t, 3) << 16) | (b(x, 3) << 24))
    x = np.where(low[0] != 0, lower, upper)
    for d, lm, base, k in ((4, low[1], 0xF0F0F0F0, 24), (2, low[2], 0xCCCCCCCC, 28), (1, low[3], 0xAAAAAAAA, 30)):
        t = x[sub ^ d]
        amt = np.uint64(d) ^ (lm & np.uint64(k))
        m = np.uint64(base) ^ lm
        x = (x & m) | (rot(t, amt) & ~m & np.uint64(0xFFFFFFFF))
    return x


def compress_halves(x, m, mv):
    x = x & m
    for i in range(4):
        t = x & mv[i]
        x = (x ^ t) | (t >> np.uint64(1 << i))
    return x


@pytest.mark.parametrize("N,K,cons", [(32, 16, "dega"), (64, 20, "dega"), (128, 64, "dega"), (256, 200, "dega"), (512, 256, "rmpolar"),
                                        (1024, 512, "dega"), (1024, 1000, "dega"), (1024, 40, "dega"), (512, 300, "random"),
                                        (1024, 777, "random")])
def test_output_stage_tables_and_method(N, K, cons):
    rs = np.random.RandomState(N + K)
    if cons == "random":
        mask = np.zeros(N, dtype=np.uint8)
        mask[rs.choice(N, K, replace=False)] = 1
    else:
        mask = (pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)).astype(np.uint8)
    tab = _native.output_tables(mask).astype(np.uint64)
    per_lane = np.zeros((16, 16), dtype=np.uint64)
    for j in range(16):
        for l in range(16):
            per_lane[l, j] = tab[(16 * (j >> 2) + l) * 4 + (j & 3)]
    low = lane_constants()
    Kw = max(1, (K + 31) // 32)
    for trial in range(20):
        u = (rs.randint(0, 2, N) & mask).astype(np.uint64)          # u_hat is zero at frozen positions
        if trial == 0:
            u = mask.astype(np.uint64)
        # sliced layout: lane s, bit j = element 16 j + s; 32-bit words lo / hi
        words = np.zeros((2, 16), dtype=np.uint64)
        for k in range(N):
            j, s = k >> 4, k & 15
            words[j >> 5, s] |= u[k] << np.uint64(j & 31)
        out = np.zeros(40, dtype=np.uint64)
        for w in range(2 if N > 512 else 1):
            x = row_transpose16(words[w], low)
            x = compress_halves(x, per_lane[:, 5 * w], [per_lane[:, 5 * w + 1 + i] for i in range(4)])
            for half in range(2):
                q = 2 * w + half
                if 16 * 16 * q >= N:
                    continue
                piece = (x >> np.uint64(16 * half)) & np.uint64(0xFFFF)
                e = per_lane[:, 10 + q]
                for l in range(16):
                    v = int(piece[l]) << int(e[l] >> np.uint64(16))
                    at = int(e[l] & np.uint64(0xFFFF)) // 4
                    out[at] |= np.uint64(v & 0xFFFFFFFF)
                    out[at + 1] |= np.uint64(v >> 32)
        want = np.zeros(40, dtype=np.uint64)
        for k, bit in enumerate(u[mask == 1]):
            want[k >> 5] |= np.uint64(int(bit) << (k & 31))
        assert np.array_equal(out[:Kw], want[:Kw]), (trial, np.flatnonzero(out[:Kw] != want[:Kw]))
        assert not out[Kw:].any()


def expand_halves(x, m, mv):
    for i in (3, 2, 1, 0):
        t = x & (mv[i] >> np.uint64(1 << i))
        x = (x ^ t) | (t << np.uint64(1 << i))
    return x & m


@pytest.mark.parametrize("N,K,cons", [(128, 64, "dega"), (256, 200, "dega"), (512, 256, "rmpolar"), (1024, 512, "dega"), (1024, 1000, "dega"),
                                        (1024, 40, "dega"), (512, 300, "random"), (1024, 777, "random")])
def test_rate_profile_as_the_output_stage_run_backwards(N, K, cons):
    """The encoder side of the one-kernel Monte-Carlo loop: u[polarMask == 1] = info (cn/polarCode.py:225-230) on the sliced
    layout, made from the information words by cutting every lane's pieces out at the table's bit offsets, masking them to
    the halfword's number of information positions (table entries 14, 15), undoing the compress moves in reverse order
    and transposing -- restated in numpy on the library's tables and compared with the scatter done bit by bit."""
    rs = np.random.RandomState(N * 3 + K)
    if cons == "random":
        mask = np.zeros(N, dtype=np.uint8)
        mask[rs.choice(N, K, replace=False)] = 1
    else:
        mask = (pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)).astype(np.uint8)
    tab = _native.output_tables(mask).astype(np.uint64)
    per_lane = np.zeros((16, 16), dtype=np.uint64)
    for j in range(16):
        for l in range(16):
            per_lane[l, j] = tab[(16 * (j >> 2) + l) * 4 + (j & 3)]
    low = lane_constants()
    for trial in range(12):
        info = rs.randint(0, 2, K)
        if trial == 0:
            info[:] = 1
        row = np.zeros(64, dtype=np.uint64)
        for k, bit in enumerate(info):
            row[k >> 5] |= np.uint64(int(bit) << (k & 31))
        row[(K + 31) // 32:] = rs.randint(0, 1 << 32, 64 - (K + 31) // 32, dtype=np.uint64)  # whatever lies behind the words is cut off
        row[(K + 31) // 32 - 1] |= np.uint64(((1 << 32) - 1) ^ ((1 << (((K - 1) & 31) + 1)) - 1)) if K & 31 else np.uint64(0)
        got = np.zeros((2, 16), dtype=np.uint64)
        for w in range(2 if N > 512 else 1):
            x = np.zeros(16, dtype=np.uint64)
            for half in range(2):
                q = 2 * w + half
                if 16 * 16 * q >= N:
                    continue
                e = per_lane[:, 10 + q]
                for l in range(16):
                    at, sh = int(e[l] & np.uint64(0xFFFF)) // 4, int(e[l] >> np.uint64(16))
                    stream = (int(row[at]) | (int(row[at + 1]) << 32)) >> sh
                    x[l] |= np.uint64((stream & 0xFFFF) << (16 * half))
            x = expand_halves(x & per_lane[:, 14 + w], per_lane[:, 5 * w], [per_lane[:, 5 * w + 1 + i] for i in range(4)])
            got[w] = row_transpose16(x, low)
        u = np.zeros(N, dtype=np.uint64)
        u[mask == 1] = info
        want = np.zeros((2, 16), dtype=np.uint64)
        for k in range(N):
            j, s = k >> 4, k & 15
            want[j >> 5, s] |= u[k] << np.uint64(j & 31)
        assert np.array_equal(got, want), trial


def test_output_tables_argument_errors():
    with pytest.raises(_native.PolarError):
        _native.output_tables(np.ones(2048, dtype=np.uint8))
    with pytest.raises(_native.PolarError):
        _native.output_tables(np.ones(16, dtype=np.uint8))
