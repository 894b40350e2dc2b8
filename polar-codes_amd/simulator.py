"""Monte-Carlo BER/FER driver: the frame loop of the reference's cn/simulator.py as a function.

The reference's loop (cn/simulator.py:117-285) draws a message, encodes, adds noise, decodes and counts
errors one frame at a time.  Here one call of the GPU library does that for a whole batch of frames
(random info bits and Philox noise generated on the device; only three counters come back), frames are
sharded over the ranks of a torch.distributed job (one process per GPU) and the counters are summed with
one all-reduce (RCCL when the tensors live on the GPU).

Frame f of an SNR point always sees the same message and noise, whatever the batch size or the number
of ranks, because the generator is keyed by (seed, global frame index).
"""
import os
import time
import warnings

import numpy as np

from channel import Channel


class BERFER:
    """Result container with the reference's field names (cn/simulator.py:109-115) plus raw counts."""

    def __init__(self):
        self.snrRange = list()
        self.ber = list()
        self.fer = list()
        self.frames = list()
        self.frameErrors = list()
        self.bitErrors = list()
        self.seconds = list()

    def as_dict(self):
        return {k: list(map(float, v)) for k, v in self.__dict__.items()}


def _dist_state(dist):
    """(rank, world, all_reduce) for an initialised torch.distributed process group, or a single rank."""
    if dist is None:
        try:
            import torch.distributed as td

            if td.is_available() and td.is_initialized():
                dist = td
        except ImportError:
            dist = None
    if dist is None:
        return 0, 1, None
    return dist.get_rank(), dist.get_world_size(), dist


def _all_reduce_counters(dist, counters, device):
    """Sum uint64[3] counters over all ranks; tensors on `device` ('cuda:<i>' -> RCCL, 'cpu' -> gloo)."""
    if dist is None:
        return counters
    import torch

    t = torch.tensor([int(c) for c in counters], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return np.array(t.cpu().tolist(), dtype=np.uint64)


def _mask_id(pcode):
    import hashlib

    return hashlib.sha1(np.ascontiguousarray(np.asarray(pcode.polarMask) != 0, dtype=np.uint8).tobytes()).hexdigest()[:16]


def _checkpoint_load(path, ident, rank=0, dist=None):
    """State of an earlier, interrupted sweep with the same arguments, or an empty one.

    Only rank 0 reads the file (it is the only writer as well) and every other rank receives what rank 0 found:
    ranks that looked for themselves could see different contents (no shared file system, a stale cache) and then
    disagree about which points are finished -- one would skip an all-reduce the others wait in."""
    import json

    found = None
    if rank == 0:
        try:
            if path and os.path.exists(path):
                with open(path) as fh:
                    state = json.load(fh)
                if state.get("ident") != ident:
                    raise ValueError("checkpoint %s was written for another sweep: %r" % (path, state.get("ident")))
                found = state
            else:
                found = {"ident": ident, "points": {}}
        except Exception as e:  # the other ranks must hear of it too, or they wait in the broadcast for ever
            found = e
    if dist is not None:
        box = [found]
        dist.broadcast_object_list(box, src=0)
        found = box[0]
    if isinstance(found, Exception):
        raise found
    return found


def _checkpoint_save(path, state, rank):
    import json

    if not path or rank != 0:
        return
    tmp = path + ".tmp"
    with open(tmp, "w") as fh:
        json.dump(state, fh)
    os.replace(tmp, path)  # never a half-written file


def shard_range(total, rank, world):
    """Contiguous share [lo, hi) of `total` frames for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_warned_specialize = False


def native_runner(pcode, decoder, L, dtype, crc=False, genie=False):
    """Default frame runner: polar_mc_run of the GPU library."""
    import _native

    dec = {"scl": _native.DEC_SCL, "pbf": _native.DEC_PBF}.get(decoder, _native.DEC_SC)
    if dec != _native.DEC_SCL and np.dtype(dtype) == np.float32 and 32 <= pcode.codewordLength <= 1024 \
            and 0 < pcode.infoLen < pcode.codewordLength:
        try:
            pcode.native.specialize()  # SC runs of a whole sweep repay the time the run-time compiler takes
        except _native.PolarError as e:  # without the run-time compiler the generic GPU kernel decodes the same bits
            global _warned_specialize
            if not _warned_specialize:
                _warned_specialize = True
                warnings.warn("run-time specialisation of the SC decoder failed (%s); the generic GPU kernel decodes the same "
                              "bits at about half the rate" % e, RuntimeWarning)
    if crc:
        pcode._attach_crc()
    if decoder == "pbf":
        import functions as pcf

        pcode.native.set_reliability(pcf.GA(pcode.codewordLength, pcode.infoLen, pcode.designSNR))

    def run(N0, seed, frame0, nframes):
        return pcode.native.mc_run(dec, L, dtype, N0, seed, frame0, nframes, crc=crc, genie=genie)

    return run


def simulate(pcode, snrRange, frames=None, errCnt=100, minFrames=100000, maxFrames=10 ** 8, seed=2024, decoder=None,
             L=None, dtype=np.float32, batch=1 << 18, snrType=None, modu=None, verbose=True, dist=None, runner=None,
             reduce_device=None, isCRC=None, genie=False, checkpoint=None):
    """Eb/N0 (or Es/N0) sweep.

    frames=int   -> exactly that many frames per SNR point (split evenly over the ranks);
    frames=None  -> the reference's stopping rule `fer >= errCnt and t >= minFrames - 1`
                    (cn/simulator.py:257, errCnt=100, t >= 99999), tested after every round of
                    world_size * batch frames, so a point may overshoot by less than one round.
    decoder/L/snrType/modu default to the attributes set on `pcode` like the reference's driver does
    (cn/simulator.py:78-83).  isCRC (default pcode.isCRC) runs CRC-aided list decoding: pcode.infoLen then
    counts the CRC bits (the reference's K2 = K + crcWidth, cn/simulator.py:22,75), the message has
    infoLen - crcWidth bits, and rate and BER refer to the message (cn/simulator.py:20,133,360).
    genie=True (or decoder='scl-oracle') decodes with sclDecoderOracle, the call the reference's loop actually
    makes (cn/simulator.py:156): the transmitted word is returned whenever the final list holds it.
    checkpoint=path keeps the state of the sweep in a small JSON file (the reference keeps results in memory only,
    cn/simulator.py:109-115, 282-284): per SNR point its seed, the counters so far, the next round of the stopping-rule
    mode and whether the point is finished; rank 0 rewrites it after every finished point and, in stopping-rule mode,
    after every round.  A later call with the same arguments and the same file skips what is finished and continues
    the point that was in progress with the frames it has not seen yet (frames are keyed by their global index, so the
    totals equal those of an uninterrupted run).  A file written for other arguments is refused.
    Returns a BERFER; every rank returns the same totals.
    """
    rank, world, dist = _dist_state(dist)
    decoder = (decoder or getattr(pcode, "decoderType", "sc") or "sc").lower()
    if decoder in ("scl-oracle", "scloracle", "genie"):
        decoder, genie = "scl", True
    if decoder in ("pbf", "progressivebitflipping"):
        decoder, isCRC = "pbf", True
        L = int(L if L is not None else getattr(pcode, "maxLevel", 1))
    if decoder not in ("sc", "scl", "pbf"):
        raise NotImplementedError("decoder %r is not provided" % (decoder,))
    L = int(L if L is not None else (getattr(pcode, "L", 0) or 1))
    snrType = snrType or getattr(pcode, "snrType", "") or "SNRb"
    modu = modu or getattr(pcode, "modu", "") or "BPSK"
    isCRC = bool(getattr(pcode, "isCRC", False) if isCRC is None else isCRC)
    if isCRC and decoder == "sc":
        raise ValueError("isCRC needs the list or the flip decoder")
    msgLen = pcode.infoLen - (pcode.crcWidth if isCRC else 0)
    rate = msgLen / pcode.codewordLength
    if genie and (decoder != "scl" or isCRC):
        raise ValueError("the genie goes with the plain list decoder")
    run = runner or native_runner(pcode, decoder, L, dtype, crc=isCRC, genie=genie)
    if reduce_device is None:
        reduce_device = "cpu"
        if dist is not None and dist.get_backend() == "nccl":
            reduce_device = "cuda:%d" % int(os.environ.get("POLAR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    result = BERFER()
    ident = {"N": int(pcode.codewordLength), "K": int(pcode.infoLen), "mask_sha1": _mask_id(pcode), "decoder": decoder, "L": L,
             "dtype": np.dtype(dtype).name, "snrType": snrType, "modu": modu, "isCRC": isCRC, "genie": bool(genie), "seed": int(seed),
             "snrRange": [float(v) for v in snrRange], "frames": None if frames is None else int(frames), "errCnt": int(errCnt),
             "minFrames": int(minFrames), "maxFrames": int(maxFrames), "batch": int(batch), "world": int(world)}
    state = _checkpoint_load(checkpoint, ident, rank, dist)
    for ip, snr in enumerate(snrRange):
        snr = float(snr)
        if verbose and rank == 0:
            print(f"\nSNR = {snr} dB")
        ch = Channel(modu, snr, snrType, rate)
        pseed = int(seed) + ip
        point = state["points"].get(str(ip), {"seed": pseed, "counters": [0, 0, 0], "next_round": 0, "done": False, "seconds": 0.0})
        total = np.array(point["counters"], dtype=np.uint64)
        t0 = time.perf_counter() - float(point["seconds"])
        if point["done"]:
            pass
        elif frames is not None:
            lo, hi = shard_range(frames, rank, world)
            mine = np.zeros(3, dtype=np.uint64)
            for f0 in range(lo, hi, batch):
                mine += run(ch.noisePower, pseed, f0, min(batch, hi - f0))
            total = _all_reduce_counters(dist, mine, reduce_device)
        else:
            rnd = int(point["next_round"])
            while True:
                f0 = (rnd * world + rank) * batch
                mine = run(ch.noisePower, pseed, f0, batch)
                total += _all_reduce_counters(dist, mine, reduce_device)
                rnd += 1
                t = int(total[0]) - 1
                if verbose and rank == 0:
                    print(f"\nt = {t} FER = {int(total[1]) / (t + 1):0.2e}")
                if (int(total[1]) >= errCnt and t >= minFrames - 1) or int(total[0]) >= maxFrames:
                    break
                state["points"][str(ip)] = {"seed": pseed, "counters": [int(v) for v in total], "next_round": rnd, "done": False,
                                            "seconds": time.perf_counter() - t0}
                _checkpoint_save(checkpoint, state, rank)
        dt = time.perf_counter() - t0
        state["points"][str(ip)] = {"seed": pseed, "counters": [int(v) for v in total], "next_round": 0, "done": True, "seconds": dt}
        _checkpoint_save(checkpoint, state, rank)
        nfr, fe, be = int(total[0]), int(total[1]), int(total[2])
        if verbose and rank == 0:
            print(f"\nAt {snr} dB FER is {fe / max(nfr, 1):0.2e}")
        result.snrRange.append(snr)
        result.ber.append(be / max(nfr * msgLen, 1))
        result.fer.append(fe / max(nfr, 1))
        result.frames.append(nfr)
        result.frameErrors.append(fe)
        result.bitErrors.append(be)
        result.seconds.append(dt)
    return result


def main(argv=None):
    """Small command line around simulate(): python simulator.py --N 1024 --K 512 --decoder scl --L 8 ..."""
    import argparse
    import json

    from polarCode import PolarCode

    ap = argparse.ArgumentParser(description="Polar code BER/FER sweep on MI355X")
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--K", type=int, default=512)
    ap.add_argument("--construct", default="dega")
    ap.add_argument("--dsnr", type=float, default=2.0)
    ap.add_argument("--decoder", default="sc", choices=["sc", "scl", "scl-oracle", "pbf"])
    ap.add_argument("--L", type=int, default=8, help="list size (scl) or the largest number of simultaneous flips (pbf)")
    ap.add_argument("--snr", type=float, nargs="+", default=[1.0, 1.5, 2.0, 2.5, 3.0])
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--errCnt", type=int, default=100)
    ap.add_argument("--minFrames", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--crc", type=lambda s: int(s, 0), default=0,
                    help="CRC generator with its leading term, e.g. 0x107; K then counts message bits and the code "
                         "carries K + width information bits (cn/simulator.py:22)")
    ap.add_argument("--checkpoint", default=None, help="JSON file that keeps the state of the sweep; a rerun continues from it")
    a = ap.parse_args(argv)
    dist = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist

        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    width = max(a.crc.bit_length() - 1, 0)
    pc = PolarCode(a.N, a.K + width, a.construct, a.dsnr, listSize=a.L)
    pc.decoderType, pc.snrType, pc.modu = a.decoder, "SNRb", "BPSK"
    pc.crcPoly, pc.crcWidth, pc.isCRC = a.crc, width, width > 0
    if a.decoder == "pbf":
        pc.maxLevel = a.L
    res = simulate(pc, a.snr, frames=a.frames, errCnt=a.errCnt, minFrames=a.minFrames, seed=a.seed,
                   dtype=np.float64 if a.f64 else np.float32, dist=dist, checkpoint=a.checkpoint)
    if dist is None or dist.get_rank() == 0:
        print(json.dumps(res.as_dict()))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
