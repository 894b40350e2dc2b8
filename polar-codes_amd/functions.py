"""Host-side function library: code construction and small helpers.

Mirrors the names and results of the reference's cn/functions.py for the functions the
encode -> channel -> SC/SCL decode path uses.  Construction is a one-off per code and stays on the
host; masks are bit-identical to the reference's (tests/test_construction.py).
"""
import math

import numpy as np


def fails(list1, list2):
    """Number of bit errors, sum |a - b| (cn/functions.py:107-109). Accepts a leading batch axis."""
    a = np.asarray(list1).astype(np.int64)
    b = np.asarray(list2).astype(np.int64)
    return np.sum(np.absolute(a - b))


def bitreversed(num: int, n) -> int:
    """n-bit reversal of num (cn/functions.py:111-113)."""
    r = 0
    for _ in range(n):
        r = (r << 1) | (num & 1)
        num >>= 1
    return r


def lowerconv(upperdecision, upperllr, lowerllr):
    """g-function of SC decoding (cn/functions.py:115-119)."""
    return lowerllr + upperllr if upperdecision == 0 else lowerllr - upperllr


def countOnes(num: int):
    """Hamming weight of num (cn/functions.py:121-130)."""
    return bin(int(num)).count("1")


def GRowsWeight(N: int, K: int = 0):
    """log2 of the row weights of G_N, i.e. popcount(i) (cn/functions.py:132-139)."""
    idx = np.arange(N, dtype=np.uint64)
    w = np.zeros(N, dtype=int)
    for b in range(max(1, int(N).bit_length())):
        w += ((idx >> np.uint64(b)) & np.uint64(1)).astype(int)
    return w


def phi_inv(x):
    """Four-piece approximation used by the Gaussian-approximation recursion (cn/functions.py:141-149).
    Works on scalars and arrays; every branch keeps the reference's order of operations."""
    if np.ndim(x) == 0:
        x = float(x)
        if x > 12:
            return 0.9861 * x - 2.3152
        elif x > 3.5:
            return x * (0.009005 * x + 0.7694) - 0.9507
        elif x > 1:
            return x * (0.062883 * x + 0.3678) - 0.1627
        return x * (0.2202 * x + 0.06448)
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 12, 0.9861 * x - 2.3152,
                    np.where(x > 3.5, x * (0.009005 * x + 0.7694) - 0.9507,
                             np.where(x > 1, x * (0.062883 * x + 0.3678) - 0.1627,
                                      x * (0.2202 * x + 0.06448))))


def _ga_from_sigma(N, sigma_sq):
    m = np.array([2 / sigma_sq], dtype=np.float64)
    while m.size != N:
        nxt = np.empty(2 * m.size, dtype=np.float64)
        nxt[0::2] = phi_inv(m)
        nxt[1::2] = 2 * m
        m = nxt
    return m


def GA(N: int, K: int, dsnr_db: float):
    """Mean LLRs of the N bit channels by density evolution with Gaussian approximation, natural index
    order (cn/functions.py:201-211); design SNR is Eb/N0 in dB."""
    sigma_sq = 1 / (2 * K / N * np.power(10, dsnr_db / 10))
    return _ga_from_sigma(N, sigma_sq)


def degaBuild(N: int, K: int, design_snr=0):
    """Rate profile (1 = information, 0 = frozen) keeping the K most reliable channels; equal
    reliabilities freeze the lower index first (cn/functions.py:484-494)."""
    order = np.argsort(GA(N, K, design_snr), kind="stable")
    mask = np.ones(N, dtype=int)
    mask[order[:N - K]] = 0
    return mask


def rmPolarBuild(N: int, K: int, design_snr=0):
    """Reed-Muller/polar hybrid (cn/functions.py:399-439): freeze whole row-weight classes while they
    fit into N-K, then the least reliable members of the next class."""
    w = GRowsWeight(N, K)
    mu = GA(N, K, design_snr)
    count = np.bincount(w, minlength=int(math.log2(N)) + 1)
    mask = np.ones(N, dtype=int)
    frozen = 0
    k = 0
    while frozen + count[k] <= N - K:
        mask[w == k] = 0
        frozen += count[k]
        k += 1
    cls = np.flatnonzero(w == k)
    cls = cls[np.argsort(mu[cls], kind="stable")]
    mask[cls[:(N - K) - frozen]] = 0
    return mask


def rmBuild(N: int, K: int):
    """Pure Reed-Muller profile: the K indices of largest popcount, lower index first among equals
    (rateProfile.m:63-68)."""
    order = np.argsort(-GRowsWeight(N), kind="stable")
    mask = np.zeros(N, dtype=int)
    mask[order[:K]] = 1
    return mask


def rateProfile(N, K, dsnrdB, snrType, construction):
    """MATLAB-style entry point (rateProfile.m:1-13,95-112): like the builders above but the design
    SNR may be Es/N0 ('snr') or Eb/N0 ('snrb').  No MATLAB/Octave is available to run the original, so
    only the 'snrb' cases are pinned (they coincide with degaBuild/rmPolarBuild away from ties)."""
    c = construction.lower()
    if c == "rm":
        return rmBuild(N, K)
    if snrType.lower() == "snr":
        sigma_sq = 1 / (2 * np.power(10, dsnrdB / 10))
    else:
        sigma_sq = 1 / (2 * (K / N) * np.power(10, dsnrdB / 10))
    mu = _ga_from_sigma(N, sigma_sq)
    if c == "dega":
        order = np.argsort(-mu, kind="stable")  # maxk: largest first, first occurrence wins
        mask = np.zeros(N, dtype=int)
        mask[order[:K]] = 1
        return mask
    if c == "rmpolar":
        w = GRowsWeight(N)
        count = np.bincount(w, minlength=int(math.log2(N)) + 1)
        mask = np.ones(N, dtype=int)
        frozen, k = 0, 0
        while frozen + count[k] <= N - K:
            mask[w == k] = 0
            frozen += count[k]
            k += 1
        cls = np.flatnonzero(w == k)
        cls = cls[np.argsort(mu[cls], kind="stable")]
        mask[cls[:(N - K) - frozen]] = 0
        return mask
    raise ValueError("None such type: %r" % (construction,))


def GN(N: int, Fn=None):
    """Dense generator matrix F^{(x)n} (cn/functions.py:351-362).  O(N^2) memory: built only on demand."""
    F = np.array([[1, 0], [1, 1]], dtype=int) if Fn is None else np.asarray(Fn)
    G = np.array([[1]], dtype=int)
    for _ in range(int(math.log2(N))):
        G = np.kron(G, F)
    return G


# ---------------- CRC (for CRC-aided list decoding) ----------------

def standardForm(crcPoly):
    """Coefficients of the generator polynomial below its leading term, MSB first, and the same as a hex string
    (cn/functions.py:83-92).  0x107 -> [0,0,0,0,0,1,1,1], '0x007'."""
    bits = bin(crcPoly)[3:]
    arrayForm = np.array([int(b) for b in bits])
    num = int(bits, 2) if bits else 0
    return arrayForm, "0x" + hex(num)[2:].zfill(len(hex(crcPoly)) - len(hex(num)))


def crcRemainder(message, crcPoly):
    """Remainder register after shifting message + width zeros through the MSB-first LFSR of
    cn/functions.py:567-583, as integers (one per row of `message`)."""
    poly, _ = standardForm(crcPoly)
    w = len(poly)
    pint = int("".join(str(int(b)) for b in poly), 2) if w else 0
    msg = np.atleast_2d(np.asarray(message)).astype(np.uint64)
    reg = np.zeros(msg.shape[0], dtype=np.uint64)
    top_shift, keep = np.uint64(max(w - 1, 0)), np.uint64((1 << w) - 1)
    one = np.uint64(1)
    for i in range(msg.shape[1] + w):
        bit = msg[:, i] if i < msg.shape[1] else np.zeros(msg.shape[0], dtype=np.uint64)
        top = (reg >> top_shift) & one
        reg = ((reg << one) & keep) | bit
        reg ^= top * np.uint64(pint)
    return reg, w


def crcEncode(message, crcPoly):
    """message followed by its CRC remainder (cn/functions.py:567-583).  Accepts a leading batch axis."""
    message = np.asarray(message)
    reg, w = crcRemainder(message, crcPoly)
    tail = ((reg[:, None] >> np.arange(w - 1, -1, -1, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(int)
    out = np.concatenate([np.atleast_2d(message).astype(int), tail], axis=1)
    return out[0] if message.ndim == 1 else out


def crcCheck(word, crcPoly):
    """True where a word (message + CRC) passes the reference's test: the remainder of crcEncode(word) is all
    zero (cn/polarCode.py:2083-2087)."""
    reg, _ = crcRemainder(word, crcPoly)
    ok = reg == 0
    return bool(ok[0]) if np.asarray(word).ndim == 1 else ok


def polynomialMod(a, b):
    """Remainder of a(x) / b(x) over GF(2), coefficient arrays MSB first, without b's leading position
    (cn/functions.py:54-70)."""
    a = [int(v) & 1 for v in a]
    b = [int(v) & 1 for v in b]
    for i in range(len(a) - len(b) + 1):
        if a[i]:
            for j, bj in enumerate(b):
                a[i + j] ^= bj
    return np.array(a[len(a) - len(b) + 1:], dtype=int)


def getGC(KI, g):
    """Systematic CRC generator matrix [I | parity] for KI message bits and generator coefficients g (MSB first,
    leading one included) -- cn/functions.py:536-547."""
    KC = len(g) - 1
    right = np.zeros((KI, KC), dtype=int)
    for i in range(KI):
        x = np.zeros(KC + KI - i, dtype=int)
        x[0] = 1
        right[i, :] = polynomialMod(x, g)
    return np.concatenate([np.eye(KI, dtype=int), right], axis=1)


# ---------------- flip candidates (bit-flipping decoders) ----------------

class Node:
    """One node of the flip tree: the positions flipped together (cn/polarCode.py:8-12)."""

    def __init__(self):
        self.sequenceU = []


def generateCriticalSet(frozenBits):
    """First index of every maximal subtree of the code tree without frozen leaves, ascending
    (cn/functions.py:279-310).  frozenBits: 1 = frozen."""
    fz = np.asarray(frozenBits).astype(int)
    heads = []

    def visit(lo, size):
        nf = int(fz[lo:lo + size].sum())
        if nf == size:
            return
        if nf == 0:
            heads.append(lo)
            return
        visit(lo, size // 2)
        visit(lo + size // 2, size // 2)

    visit(0, fz.size)
    return np.array(heads, dtype=int)


def modifyCriticalSet(polarMask, iMax):
    """Critical set of the information bits after position iMax (cn/functions.py:312-319)."""
    m = np.asarray(polarMask).astype(int).copy()
    m[:int(iMax) + 1] = 0
    return generateCriticalSet(1 - m)


def constructTree(polarMask, maxLevel):
    """All flip sequences up to maxLevel simultaneous flips, level by level (cn/functions.py:321-349)."""
    polarMask = np.asarray(polarMask).astype(int)
    nodes = [[] for _ in range(maxLevel)]
    for lvl in range(maxLevel):
        parents = [None] if lvl == 0 else nodes[lvl - 1]
        for parent in parents:
            cand = generateCriticalSet(1 - polarMask) if parent is None else modifyCriticalSet(polarMask, parent.sequenceU[-1])
            for u in cand:
                node = Node()
                node.sequenceU = ([] if parent is None else list(parent.sequenceU)) + [int(u)]
                nodes[lvl].append(node)
    return nodes
