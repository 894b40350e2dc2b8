"""Seeded test frames, reproducible from (mask, snr_db, seed) alone.

Same convention as the fixture generator: the legacy MT19937 stream seeded with `seed` yields the
K message bits first and the N standard-normal noise samples second; the channel arithmetic follows
cn/channel.py:9-14,19,35,56 of the reference (N0 from Eb/N0, s = 1-2x, y = s + sqrt(N0/2) z,
llr = (4/N0) y), all in fp64.
"""
import numpy as np


def unpack(bits, n):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), axis=-1, bitorder="little")[..., :n]


def pack(bits):
    return np.packbits(np.asarray(bits, dtype=np.uint8), axis=-1, bitorder="little")


def n0_ebn0(snr_db, rate):
    return 1 / (np.log2(2) * rate * pow(10, snr_db / 10))


def make_frame(mask, snr_db, seed, encode):
    mask = np.asarray(mask)
    N, K = mask.size, int(mask.sum())
    rs = np.random.RandomState(seed)
    msg = rs.randint(0, 2, size=K, dtype=int)
    z = rs.standard_normal(N)
    x = np.asarray(encode(msg, mask)).astype(np.int64)
    n0 = n0_ebn0(snr_db, K / N)
    s = 1 - 2 * x
    y = s + np.sqrt(n0 / 2) * z
    llr = (4 / n0) * y
    return msg.astype(np.uint8), x.astype(np.uint8), llr
