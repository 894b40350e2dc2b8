#!/usr/bin/env python3
"""Golden-vector generator (development container only).

Imports the reference implementation from /root/reference/cn, runs it on seeded
inputs and writes small .npz fixtures under tests/golden/.  Nothing from the
reference is copied: only inputs and the outputs it produced are stored.

Frame convention shared with tests/ (see tests/frames.py):
    np.random.seed(seed)                      # legacy MT19937 stream
    msg   = np.random.randint(0, 2, size=K, dtype=int)
    noise = np.random.standard_normal(N)      # drawn by Channel.addNoise
so a frame is reproducible from (N, K, mask, snr_db, seed) alone.

Usage: python tools/gen_golden.py [masks encoder channel sc scl curves ...]
"""
import contextlib
import hashlib
import io
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import warnings

import numpy as np

REF = "/root/reference/cn"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_ref():
    sys.path.insert(0, REF)
    sys.stdin = open(os.devnull)  # any input() in the reference raises instead of blocking
    with contextlib.redirect_stdout(io.StringIO()):
        import channel  # noqa
        import functions  # noqa
        import polarCode  # noqa
    return functions, polarCode, channel


def packbits(a):
    return np.packbits(np.asarray(a, dtype=np.uint8), axis=-1, bitorder="little")


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_code(polarCode, functions, N, K, construct, dsnr):
    """PolarCode without paying the dense kron(G_N) of its constructor at large N."""
    if N <= 2048:
        with contextlib.redirect_stdout(io.StringIO()):
            return polarCode.PolarCode(N, K, construct, dsnr)
    pc = polarCode.PolarCode.__new__(polarCode.PolarCode)
    pc.codewordLength, pc.infoLen, pc.n, pc.designSNR = N, K, int(np.log2(N)), dsnr
    pc.polarMask = functions.degaBuild(N, K, dsnr) if construct == "dega" else functions.rmPolarBuild(N, K, dsnr)
    pc.L = 0
    return pc


def frame(pc, ch, seed):
    np.random.seed(seed)
    msg = np.random.randint(0, 2, size=pc.infoLen, dtype=int)
    x = pc.polarEncode(msg)
    y = ch.addNoise(ch.modulate(x))
    llr = ch.calcLLR(y)
    return msg, np.asarray(x), np.asarray(y), np.asarray(llr, dtype=np.float64)


# --------------------------------------------------------------------------- G1
def gen_masks(functions, polarCode, channel):
    out = {}
    for (N, K) in [(8, 4), (128, 64), (512, 256), (1024, 512), (16384, 8192), (256, 64), (256, 192)]:
        for ds in (2.0, 2.25):
            m = np.asarray(functions.degaBuild(N, K, ds), dtype=np.uint8)
            out[f"dega_{N}_{K}_{ds}"] = packbits(m) if N > 8 else m
    for (N, K) in [(512, 256), (128, 64), (1024, 512), (256, 100)]:
        m = np.asarray(functions.rmPolarBuild(N, K, 2.0), dtype=np.uint8)
        out[f"rmpolar_{N}_{K}_2.0"] = packbits(m)
    for (N, K) in [(8, 4), (128, 64), (512, 256), (1024, 512)]:
        out[f"ga_{N}_{K}_2.0"] = np.asarray(functions.GA(N, K, 2.0), dtype=np.float64)
    out["ga_1024_512_2.25"] = np.asarray(functions.GA(1024, 512, 2.25), dtype=np.float64)
    xs = np.array([0.0, 0.5, 1.0, 1.0000001, 2.0, 3.5, 3.5000001, 7.0, 12.0, 12.0000001, 40.0, 1e3])
    out["phi_inv_x"] = xs
    out["phi_inv_y"] = np.array([functions.phi_inv(float(v)) for v in xs])
    out["bitreversed_10"] = np.array([functions.bitreversed(i, 10) for i in range(1024)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)
    print("masks:", len(out), "arrays")


# --------------------------------------------------------------------------- G2
def gen_encoder(functions, polarCode, channel):
    out = {}
    for (N, K) in [(8, 4), (128, 64), (512, 256), (1024, 512), (16384, 8192)]:
        pc = make_code(polarCode, functions, N, K, "dega", 2.0)
        rs = np.random.RandomState(1000 + N)
        nrand = 64 if N <= 1024 else 8
        info = rs.randint(0, 2, size=(nrand, K)).astype(int)
        # unit vectors + all-ones + all-zeros
        units = np.zeros((min(K, 16) + 2, K), dtype=int)
        for j in range(min(K, 16)):
            units[j, (j * 37) % K] = 1
        units[-2, :] = 1
        info = np.concatenate([info, units], axis=0)
        xs = np.stack([np.asarray(pc.polarEncode(row.copy())) for row in info])
        out[f"info_{N}_{K}"] = packbits(info)
        out[f"x_{N}_{K}"] = packbits(xs)
        # raw butterfly on arbitrary u (no frozen structure)
        u = rs.randint(0, 2, size=(8, N)).astype(int)
        xu = np.stack([np.asarray(pc.mul_matrix(row.copy())) for row in u])
        out[f"u_{N}"] = packbits(u)
        out[f"xu_{N}"] = packbits(xu)
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)
    print("encoder:", len(out), "arrays")


# --------------------------------------------------------------------------- G3
def gen_channel(functions, polarCode, channel):
    out = {}
    snrs = np.arange(0.0, 4.01, 0.5)
    out["snr_db"] = snrs
    for st in ("SNRb", "SNR"):
        for rc in (0.5, 0.25, 0.75):
            out[f"N0_{st}_{rc}"] = np.array([channel.Channel("BPSK", float(s), st, rc).noisePower for s in snrs])
    ch = channel.Channel("BPSK", 2.0, "SNRb", 0.5)
    np.random.seed(77)
    x = np.random.randint(0, 2, size=256, dtype=int)
    s = ch.modulate(x)
    y = ch.addNoise(s)
    llr = ch.calcLLR(y)
    out["seed77_x"] = x.astype(np.uint8)
    out["seed77_s"] = np.asarray(s, dtype=np.int64)
    out["seed77_y"] = np.asarray(y, dtype=np.float64)
    out["seed77_llr"] = np.asarray(llr, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "channel.npz"), **out)
    print("channel:", len(out), "arrays")


# --------------------------------------------------------------------------- G4
def gen_sc(functions, polarCode, channel):
    out = {}
    # the 8-bit known-answer case quoted in the reference's main.py string block
    pc = make_code(polarCode, functions, 8, 4, "dega", 2.0)
    l8 = np.array([-2.0, -2.5, -4.0, 1.0, -6.5, 6.0, 16.6, 3.5])
    dec = pc.scDecoder(l8.copy())
    out["kat8_llr"] = l8
    out["kat8_uhat"] = np.asarray(pc.uHat, dtype=np.uint8)
    out["kat8_dec"] = np.asarray(dec, dtype=np.uint8)
    out["kat8_decllr"] = np.asarray(pc.llrs, dtype=np.float64)

    cfgs = [
        (128, 64, "dega", 2.0, [(1.0, 100), (2.0, 100), (4.0, 20)], 1 << 30),
        (1024, 512, "dega", 2.0, [(1.0, 24), (2.0, 48), (3.0, 12)], 8),
        (512, 256, "rmpolar", 2.0, [(2.0, 16)], 4),
        (16384, 8192, "dega", 2.0, [(2.0, 3)], 1),
    ]
    for (N, K, cons, ds, pts, nfull) in cfgs:
        pc = make_code(polarCode, functions, N, K, cons, ds)
        if N > 2048:  # scDecoder needs only these attributes
            pc.polarMask = np.asarray(pc.polarMask)
        for (snr, nfr) in pts:
            ch = channel.Channel("BPSK", snr, "SNRb", K / N)
            tag = f"{cons}_{N}_{K}_{snr}"
            seeds, uh, hsh, nerr, full_llr, full_dec = [], [], [], [], [], []
            t0 = time.time()
            for f in range(nfr):
                seed = 5000 + 17 * f + int(snr * 100) + N
                msg, x, y, llr = frame(pc, ch, seed)
                dec = pc.scDecoder(llr.copy())
                seeds.append(seed)
                uh.append(packbits(pc.uHat))
                hsh.append(np.frombuffer(hashlib.sha1(np.asarray(pc.llrs, dtype=np.float64).tobytes()).digest(), dtype=np.uint8))
                nerr.append(int(np.sum(np.abs(msg - dec))))
                if f < nfull:
                    full_llr.append(llr)
                    full_dec.append(np.asarray(pc.llrs, dtype=np.float64))
            out[f"{tag}_seeds"] = np.array(seeds, dtype=np.int64)
            out[f"{tag}_uhat"] = np.stack(uh)
            out[f"{tag}_decllr_sha1"] = np.stack(hsh)
            out[f"{tag}_biterr"] = np.array(nerr, dtype=np.int64)
            out[f"{tag}_llr"] = np.stack(full_llr)
            out[f"{tag}_decllr"] = np.stack(full_dec)
            print(f"sc {tag}: {nfr} frames, {sum(e > 0 for e in nerr)} frame errors, {time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, "sc.npz"), **out)


# --------------------------------------------------------------------------- G5
def _candidate_gap(pc, pos):
    """Smallest relative gap between adjacent candidate metrics at this fork
    (exact ties between -inf metrics are deterministic and ignored)."""
    pms = []
    for p in pc.pathList:
        Li = p.LLRs[0] / np.log(2)
        with np.errstate(over="ignore"):
            pms.append(p.pathMetric + (1 - np.log2(1 + 2 ** (-Li))))
            pms.append(p.pathMetric + (1 - np.log2(1 + 2 ** (Li))))
    pms = np.sort(np.array(pms, dtype=np.float64))
    fin = pms[np.isfinite(pms)]
    if len(fin) < 2:
        return np.inf
    d = np.diff(fin)
    sc = np.maximum(np.abs(fin[1:]), 1.0)
    return float(np.min(d / sc))


def run_scl(pc, polarCode, llr, L):
    pc.L = L
    gaps = []
    orig = polarCode.PolarCode.polarSCLFork

    def wrapped(self, pos):
        if 2 * len(self.pathList) > self.L:
            gaps.append(_candidate_gap(self, pos))
        return orig(self, pos)

    polarCode.PolarCode.polarSCLFork = wrapped
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dec = pc.sclDecoder(llr.copy())
    finally:
        polarCode.PolarCode.polarSCLFork = orig
    pms = np.array([float(p.pathMetric) for p in pc.pathList], dtype=np.float64)
    uh = np.stack([packbits(p.uHat) for p in pc.pathList])
    fin = np.sort(pms[np.isfinite(pms)])
    if len(fin) > 1:
        gaps.append(float(np.min(np.diff(fin) / np.maximum(np.abs(fin[1:]), 1.0))))
    return np.asarray(dec), pms, uh, (min(gaps) if gaps else np.inf), (pc.ANV, pc.sortNum, pc.unitCal)


def gen_scl(functions, polarCode, channel):
    out = {}
    pc = make_code(polarCode, functions, 8, 4, "dega", 2.0)
    l8 = np.array([-2.0, -2.5, -4.0, 1.0, -6.5, 6.0, 16.6, 3.5])
    dec, pms, uh, gap, cnt = run_scl(pc, polarCode, l8, 4)
    out["kat8_llr"], out["kat8_dec"], out["kat8_pm"], out["kat8_uhat"] = l8, dec.astype(np.uint8), pms, uh
    cfgs = [
        (128, 64, "dega", 2.0, 2, [(0.0, 150), (2.0, 50)]),
        (128, 64, "dega", 2.0, 4, [(0.0, 150), (2.0, 50)]),
        (128, 64, "dega", 2.0, 8, [(0.0, 60)]),
        (128, 64, "dega", 2.0, 16, [(0.0, 30)]),
        (128, 64, "dega", 2.0, 32, [(0.0, 30)]),
        (1024, 512, "dega", 2.0, 8, [(1.0, 40), (2.0, 60)]),
        (512, 256, "rmpolar", 2.0, 32, [(2.0, 12)]),
        (512, 256, "dega", 2.0, 32, [(1.0, 6), (2.0, 8)]),
    ]
    for (N, K, cons, ds, L, pts) in cfgs:
        pc = make_code(polarCode, functions, N, K, cons, ds)
        for (snr, nfr) in pts:
            ch = channel.Channel("BPSK", snr, "SNRb", K / N)
            tag = f"{cons}_{N}_{K}_L{L}_{snr}"
            seeds, decs, pml, uhl, gapl, nerr, cnts = [], [], [], [], [], [], []
            t0 = time.time()
            for f in range(nfr):
                seed = 9000 + 13 * f + int(snr * 100) + N + L
                msg, x, y, llr = frame(pc, ch, seed)
                dec, pms, uh, gap, cnt = run_scl(pc, polarCode, llr, L)
                seeds.append(seed)
                decs.append(packbits(dec))
                pml.append(pms)
                uhl.append(uh)
                gapl.append(gap)
                cnts.append(cnt)
                nerr.append(int(np.sum(np.abs(msg - dec))))
            out[f"{tag}_seeds"] = np.array(seeds, dtype=np.int64)
            out[f"{tag}_dec"] = np.stack(decs)
            out[f"{tag}_pm"] = np.stack(pml)
            out[f"{tag}_uhat"] = np.stack(uhl)
            out[f"{tag}_gap"] = np.array(gapl)
            out[f"{tag}_biterr"] = np.array(nerr, dtype=np.int64)
            out[f"{tag}_counts"] = np.array(cnts, dtype=np.int64)
            print(f"scl {tag}: {nfr} frames, {sum(e > 0 for e in nerr)} frame errors, min gap {min(gapl):.3e}, {time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, "scl.npz"), **out)


# --------------------------------------------------------------------------- G6
def _curve_worker(args):
    (N, K, cons, ds, dec, L, snr, seed0, nfr) = args
    functions, polarCode, channel = load_ref()
    pc = make_code(polarCode, functions, N, K, cons, ds)
    pc.L = L
    ch = channel.Channel("BPSK", snr, "SNRb", K / N)
    fe = be = be2 = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for f in range(nfr):
            msg, x, y, llr = frame(pc, ch, seed0 + f)
            d = pc.scDecoder(llr) if dec == "sc" else pc.sclDecoder(llr)
            e = int(np.sum(np.abs(msg - d)))
            be += e
            be2 += e * e
            fe += e > 0
    return (nfr, fe, be, be2)


def gen_curves(functions, polarCode, channel):
    import multiprocessing as mp

    nproc = int(os.environ.get("GOLDEN_PROCS", "6"))
    plan = [
        ("sc_128_64", 128, 64, "dega", 2.0, "sc", 0, [1.0, 1.5, 2.0, 2.5, 3.0], 25000),
        ("sc_1024_512", 1024, 512, "dega", 2.0, "sc", 0, [1.0, 1.5, 2.0, 2.5, 3.0], 20000),
        ("scl8_1024_512", 1024, 512, "dega", 2.0, "scl", 8, [1.0, 1.5, 2.0, 2.5], 5000),
        ("scl4_128_64", 128, 64, "dega", 2.0, "scl", 4, [1.0, 2.0, 3.0], 6000),
    ]
    out = {}
    dest = os.environ.get("GOLDEN_CURVES_OUT", os.path.join(OUT, "curves.npz"))
    if os.environ.get("GOLDEN_CURVES_RESUME") and os.path.exists(dest):  # keep the points an interrupted run finished
        out = dict(np.load(dest, allow_pickle=False))
    with mp.Pool(nproc) as pool:
        for (tag, N, K, cons, ds, dec, L, snrs, nfr) in plan:
            rows = [tuple(r) for r in out.get(tag, [])]
            for snr in snrs:
                if any(r[0] == snr for r in rows):
                    continue
                t0 = time.time()
                per = nfr // nproc
                jobs = [(N, K, cons, ds, dec, L, snr, 100000 * (1 + j) + int(snr * 1000), per) for j in range(nproc)]
                res = pool.map(_curve_worker, jobs)
                fr = sum(r[0] for r in res)
                fe = sum(r[1] for r in res)
                be = sum(r[2] for r in res)
                be2 = sum(r[3] for r in res)
                rows.append((snr, fr, fe, be, be2))  # be2 = sum of squared per-frame bit errors
                print(f"curve {tag} {snr} dB: frames {fr} fe {fe} be {be} ({time.time()-t0:.0f}s)", flush=True)
                out[tag] = np.array(rows, dtype=np.float64)
                np.savez_compressed(dest, **out)


# --------------------------------------------------------------------------- main.py flow (bit-reversed variant)
def gen_mainflow(functions, polarCode, channel):
    """cn/main.py is a script: executing it decodes one random (1024,512) frame at 1 dB in its bit-reversed
    ordering.  Seed numpy, execute it, and keep the inputs and outputs it leaves in its module globals."""
    import importlib

    out = {}
    main = None
    for s in range(1, 9):
        np.random.seed(700 + s)
        with contextlib.redirect_stdout(io.StringIO()):
            main = importlib.import_module("main") if main is None else importlib.reload(main)
        out[f"s{s}_message"] = packbits(main.message)
        out[f"s{s}_x"] = packbits(main.x)
        out[f"s{s}_llr"] = np.asarray(main.llrCh, dtype=np.float64)
        out[f"s{s}_uhat"] = packbits(main.uHat)
        out[f"s{s}_decoded"] = packbits(main.decoded)
        out[f"s{s}_decllr"] = np.asarray(main.llrOb, dtype=np.float64)
    out["mask"] = packbits(main.polarMask)
    out["dsnr"] = np.array([2.5])
    np.savez_compressed(os.path.join(OUT, "mainflow.npz"), **out)
    print("mainflow:", len(out), "arrays; last frame correct:", bool((main.message == main.decoded).all()))


# --------------------------------------------------------------------------- CRC-aided list decoding
def gen_cascl(functions, polarCode, channel):
    out = {}
    rs = np.random.RandomState(42)
    for poly, name in [(0x107, "crc8"), (0x1800063, "crc24"), (0x1621, "crc12")]:
        m = rs.randint(0, 2, size=(6, 40)).astype(int)
        out[f"{name}_msg"] = packbits(m)
        out[f"{name}_enc"] = packbits(np.stack([functions.crcEncode(r.copy(), poly) for r in m]))
        out[f"{name}_std"] = np.asarray(functions.standardForm(poly)[0], dtype=np.uint8)
    g = np.array([1, 0, 0, 0, 0, 0, 1, 1, 1])
    out["getGC_20_crc8"] = np.asarray(functions.getGC(20, g), dtype=np.uint8)
    cfgs = [(128, 64, 0x107, 8, 4, 1.0, 80), (256, 128, 0x107, 8, 8, 1.0, 60), (1024, 536, 0x1800063, 24, 8, 1.0, 24)]
    for (N, K, poly, w, L, snr, nfr) in cfgs:
        pc = make_code(polarCode, functions, N, K, "dega", 2.0)
        pc.L, pc.crcPoly, pc.crcWidth = L, poly, w
        ch = channel.Channel("BPSK", snr, "SNRb", (K - w) / N)
        tag = f"{N}_{K}_w{w}_L{L}"
        seeds, decs, xs, nerr, plain = [], [], [], [], []
        t0 = time.time()
        for f in range(nfr):
            seed = 31000 + 7 * f + N
            np.random.seed(seed)
            msg = np.random.randint(0, 2, size=K - w, dtype=int)
            x = pc.CRCPolarEncode2(msg)
            y = ch.addNoise(ch.modulate(x))
            llr = ch.calcLLR(y)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                dec = pc.sclDecoder(llr.copy(), isCRC=True)
                first = pc.extract(pc.pathList[0].uHat)[:K - w]
            seeds.append(seed)
            decs.append(packbits(dec))
            xs.append(packbits(x))
            nerr.append(int(np.sum(np.abs(msg - dec))))
            plain.append(int(np.sum(np.abs(msg - first))))
        out[f"{tag}_seeds"] = np.array(seeds, dtype=np.int64)
        out[f"{tag}_dec"] = np.stack(decs)
        out[f"{tag}_x"] = np.stack(xs)
        out[f"{tag}_biterr"] = np.array(nerr, dtype=np.int64)
        out[f"{tag}_biterr_first_path"] = np.array(plain, dtype=np.int64)
        print(f"cascl {tag}: {nfr} frames, {sum(e > 0 for e in nerr)} frame errors with CRC selection, "
              f"{sum(e > 0 for e in plain)} with the first path, {time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, "cascl.npz"), **out)


# --------------------------------------------------------------------------- N3: sclDecoderOracle / polarSCLFork2
def gen_genie(functions, polarCode, channel):
    """Frames decoded by sclDecoderOracle (cn/polarCode.py:1267-1317), the call cn/simulator.py:156 makes:
    list decoding with the in-place fork polarSCLFork2 and the genie's choice (pcode.message)."""
    out = {}
    cfgs = [(128, 64, "dega", 2, 0.0, 120), (128, 64, "dega", 4, 0.0, 120), (128, 64, "dega", 8, 1.0, 60),
            (256, 128, "dega", 16, 0.5, 30), (1024, 512, "dega", 8, 1.0, 40), (512, 256, "rmpolar", 32, 1.5, 10)]
    for (N, K, cons, L, snr, nfr) in cfgs:
        pc = make_code(polarCode, functions, N, K, cons, 2.0)
        pc.L = L
        ch = channel.Channel("BPSK", snr, "SNRb", K / N)
        tag = f"{cons}_{N}_{K}_L{L}_{snr}"
        seeds, decs, pml, uhl, nerr, where, plain, cnts = [], [], [], [], [], [], [], []
        t0 = time.time()
        for f in range(nfr):
            seed = 47000 + 11 * f + N + L
            msg, x, y, llr = frame(pc, ch, seed)
            pc.message = np.zeros(N, dtype=int)
            pc.message[pc.polarMask == 1] = msg
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                dec = np.asarray(pc.sclDecoderOracle(llr.copy()))
            hit = [j for j, q in enumerate(pc.pathList) if (q.uHat == pc.message).all()]
            first = pc.extract(pc.pathList[0].uHat)
            seeds.append(seed)
            decs.append(packbits(dec))
            pml.append(np.array([float(q.pathMetric) for q in pc.pathList]))
            uhl.append(np.stack([packbits(q.uHat) for q in pc.pathList]))
            nerr.append(int(np.sum(np.abs(msg - dec))))
            plain.append(int(np.sum(np.abs(msg - first))))
            where.append(hit[0] if hit else -1)
            cnts.append((pc.ANV, pc.sortNum, pc.unitCal))
        out[f"{tag}_seeds"] = np.array(seeds, dtype=np.int64)
        out[f"{tag}_dec"] = np.stack(decs)
        out[f"{tag}_pm"] = np.stack(pml)
        out[f"{tag}_uhat"] = np.stack(uhl)
        out[f"{tag}_biterr"] = np.array(nerr, dtype=np.int64)
        out[f"{tag}_biterr_first_path"] = np.array(plain, dtype=np.int64)
        out[f"{tag}_truth_at"] = np.array(where, dtype=np.int64)
        out[f"{tag}_counts"] = np.array(cnts, dtype=np.int64)
        print(f"genie {tag}: {nfr} frames, {sum(e > 0 for e in nerr)} frame errors with the genie, "
              f"{sum(e > 0 for e in plain)} with the first path, truth deeper than first in {sum(w > 0 for w in where)}, "
              f"{time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, "genie.npz"), **out)


# --------------------------------------------------------------------------- N4: bit flipping
def gen_pbf(functions, polarCode, channel):
    """generateCriticalSet / modifyCriticalSet / constructTree, scDecoder with a flip list, and frames decoded
    by progressiveBitFlipping (cn/polarCode.py:1573-1628) with the number of SC runs each took."""
    out = {}
    for (N, K, cons) in [(128, 72, "dega"), (256, 136, "dega"), (1024, 536, "dega"), (512, 256, "rmpolar"), (64, 20, "dega")]:
        mask = np.asarray(functions.degaBuild(N, K, 2.0) if cons == "dega" else functions.rmPolarBuild(N, K, 2.0))
        cs = functions.generateCriticalSet((mask + 1) % 2)
        out[f"crit_{cons}_{N}_{K}"] = np.asarray(cs, dtype=np.int64)
        A = np.flatnonzero(mask)
        for t in (0, len(A) // 3, len(A) - 2):
            out[f"crit_{cons}_{N}_{K}_after_{A[t]}"] = np.asarray(functions.modifyCriticalSet(mask, A[t]), dtype=np.int64)
    mask = np.asarray(functions.degaBuild(64, 20, 2.0))
    tree = functions.constructTree(mask, 2)
    out["tree_64_20_level0"] = np.array([nd.sequenceU for nd in tree[0]], dtype=np.int64)
    out["tree_64_20_level1"] = np.array([nd.sequenceU for nd in tree[1]], dtype=np.int64)

    # scDecoder(softMess, sequenceU)
    pc = make_code(polarCode, functions, 128, 64, "dega", 2.0)
    ch = channel.Channel("BPSK", 1.0, "SNRb", 0.5)
    A = np.flatnonzero(pc.polarMask)
    flips = [[int(A[3])], [int(A[5]), int(A[20])], [int(A[0]), int(A[1]), int(A[63])], [5]]  # 5 is a frozen index
    uh, dl, fl = [], [], []
    for f in range(16):
        msg, x, y, llr = frame(pc, ch, 8100 + f)
        seq = flips[f % len(flips)]
        pc.scDecoder(llr.copy(), seq)
        uh.append(packbits(pc.uHat))
        dl.append(np.asarray(pc.llrs, dtype=np.float64))
        fl.append(seq + [-1] * (3 - len(seq)))
    out["flip_128_64_seeds"] = np.arange(8100, 8116, dtype=np.int64)
    out["flip_128_64_uhat"] = np.stack(uh)
    out["flip_128_64_decllr"] = np.stack(dl)
    out["flip_128_64_seq"] = np.array(fl, dtype=np.int64)

    cfgs = [(128, 72, 0x107, 8, 1, 1.5, 150), (128, 72, 0x107, 8, 2, 1.5, 60), (256, 136, 0x107, 8, 1, 1.5, 60),
            (128, 72, 0x107, 8, 3, 2.0, 24), (128, 72, 0x107, 8, 3, 1.0, 8), (1024, 536, 0x1800063, 24, 1, 1.75, 30)]
    for (N, K, poly, w, maxLevel, snr, nfr) in cfgs:
        pc = make_code(polarCode, functions, N, K, "dega", 2.0)
        pc.crcPoly, pc.crcWidth, pc.maxLevel = poly, w, maxLevel
        ch = channel.Channel("BPSK", snr, "SNRb", (K - w) / N)
        tag = f"{N}_{K}_w{w}_T{maxLevel}_{snr}"
        calls = []
        orig = polarCode.PolarCode.scDecoder

        def counted(self, softMess, sequenceU=[]):
            calls.append(list(sequenceU))
            return orig(self, softMess, sequenceU)

        polarCode.PolarCode.scDecoder = counted
        seeds, decs, nerr, nrun, last, okl = [], [], [], [], [], []
        t0 = time.time()
        try:
            for f in range(nfr):
                seed = 52000 + 7 * f + N + maxLevel
                np.random.seed(seed)
                msg = np.random.randint(0, 2, size=K - w, dtype=int)
                x = pc.CRCPolarEncode2(msg)
                llr = np.asarray(ch.calcLLR(ch.addNoise(ch.modulate(x))), dtype=np.float64)
                del calls[:]
                dec = np.asarray(pc.progressiveBitFlipping(llr.copy()))
                word = pc.extract(pc.uHat)
                seeds.append(seed)
                decs.append(packbits(dec))
                nerr.append(int(np.sum(np.abs(msg - dec))))
                nrun.append(len(calls))
                last.append(calls[-1] + [-1] * (3 - len(calls[-1])))
                okl.append(int(np.sum(functions.crcEncode(word, poly)[-w:]) == 0))
        finally:
            polarCode.PolarCode.scDecoder = orig
        out[f"{tag}_seeds"] = np.array(seeds, dtype=np.int64)
        out[f"{tag}_dec"] = np.stack(decs)
        out[f"{tag}_biterr"] = np.array(nerr, dtype=np.int64)
        out[f"{tag}_runs"] = np.array(nrun, dtype=np.int64)
        out[f"{tag}_last_seq"] = np.array(last, dtype=np.int64)
        out[f"{tag}_crc_ok"] = np.array(okl, dtype=np.int64)
        print(f"pbf {tag}: {nfr} frames, {sum(e > 0 for e in nerr)} frame errors, {sum(r > 1 for r in nrun)} needed flips, "
              f"max runs {max(nrun)}, {sum(1 - o for o in okl)} gave up, {time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(OUT, "pbf.npz"), **out)


SECTIONS = {"masks": gen_masks, "encoder": gen_encoder, "channel": gen_channel, "sc": gen_sc, "scl": gen_scl, "curves": gen_curves, "mainflow": gen_mainflow, "cascl": gen_cascl, "genie": gen_genie, "pbf": gen_pbf}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["masks", "encoder", "channel", "sc", "scl"]
    mods = load_ref()
    for w in which:
        SECTIONS[w](*mods)
