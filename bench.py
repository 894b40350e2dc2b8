#!/usr/bin/env python3
"""Benchmark of the polar decode hot path on MI355X.

python bench.py --gpus N --steps K --warmup W
  N = 1: run directly.  N > 1: launched by torch.distributed.run, one rank per GPU.

Workload (the configuration the headline metric is quoted on): Polar(1024,512) `dega` code, SC decoding of 10^7 frames per GPU,
BPSK-AWGN at Eb/N0 = 2 dB, fp32 LLRs already resident in HBM (generated on the device by the library's
Philox channel before the timed region).  A step = one pass of the decoder over the whole batch
(decode + extract information bits + count errors); with N > 1 every rank decodes its own 10^7 frames
(weak scaling, frames keyed by global index) and the three error counters are all-reduced (RCCL) at the
end of each step.  value = decoded information Mb/s summed over all ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "polar-codes_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

# Peaks from the MI355X micro-architecture guide: HBM3E 8 TB/s; vector ALU 256 CUs x 4 SIMD x 32 lanes
# x 2.4 GHz = 78.6 T lane-operations/s.
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TLOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def measured_traffic(kernel_name, frames):
    """(HBM bytes per launch, where that figure comes from) for a kernel.  The figure is NOT measured by this run: it is
    looked up in profiles/traffic.json, which tools/refresh_profiles.py writes from separate rocprofv3 PMC passes
    ({kernel prefix: {bytes_per_frame, what, commit, date, file}}; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / frames,
    with the gfx950 FETCH_SIZE correction), scaled by the frames of this launch.  (None, None) when no pass is on file."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, None
    for k, v in tab.items():
        if k.startswith("_") or not kernel_name.startswith(k):
            continue
        if isinstance(v, dict):
            src = {key: v.get(key) for key in ("file", "commit", "date", "what", "frames_per_launch")}
            src["lookup"] = "profiles/traffic.json (static table from earlier PMC passes, scaled to this launch)"
            return v["bytes_per_frame"] * frames, src
        return v * frames, {"lookup": "profiles/traffic.json (static table from earlier PMC passes, scaled to this launch)"}
    return None, None


def leg_roofline(kernel, kernel_ms, frames, N, K, paths=1, hbm_bytes_per_frame=None):
    """Roofline of a secondary leg from the algorithmic counts per frame: vector-ALU lane operations
    paths x 3.5 N log2 N (f: 4, g: 2, partial-sum xor: 1 per element and level, per path of a list), and HBM bytes
    4 N in + K/8 out (external-LLR mode; 0 in for the one-kernel Monte-Carlo loop).  Kernel time from the HIP events
    on the launch stream; `traffic` from the PMC passes on file for this kernel, if any."""
    ops = paths * 3.5 * N * np.log2(N)
    byts = (4 * N + K / 8) if hbm_bytes_per_frame is None else hbm_bytes_per_frame
    secs = kernel_ms * 1e-3
    traffic, traffic_source = measured_traffic(kernel, frames)
    return {"bound": "valu", "achieved": ops * frames / secs / 1e12, "peak": VALU_PEAK_TLOPS, "unit": "Tlane-op/s",
            "frac": ops * frames / secs / 1e12 / VALU_PEAK_TLOPS, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_lane_ops_per_frame": ops, "algorithmic_bytes_per_frame": byts,
            "hbm_GBs_algorithmic": byts * frames / secs / 1e9, "hbm_frac": byts * frames / secs / 1e9 / HBM_PEAK_GBS,
            "kernel": kernel, "kernel_ms": kernel_ms}


def lds_bytes_algorithmic(N):
    """Per-frame LDS traffic of plain SC: N log2 N LLR updates of 2 loads + 1 store (fp32) + 1 partial-sum
    byte per g, and (N/2) log2 N partial-sum XOR pairs of 2 B in + 2 B out: 14.5 N log2 N bytes."""
    return 14.5 * N * np.log2(N)


def cpu_baseline(mask, llr_sample, threads, min_seconds=10.0):
    """The CPU oracle's SC decoder (a C restatement of the reference) on the host cores: whole passes
    over the sample, one slice per thread, until at least `min_seconds` have elapsed."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle

    oracle.lib()
    parts = np.array_split(llr_sample, threads)
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        while True:
            list(ex.map(lambda a: oracle.sc_decode_batch(a, mask), parts))
            done += llr_sample.shape[0]
            dt = time.perf_counter() - t0
            if dt >= min_seconds:
                break
    return done / dt, dt, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=10 ** 7, help="frames per GPU per step")
    ap.add_argument("--snr", type=float, default=2.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary SCL / Monte-Carlo legs")
    a = ap.parse_args()

    # N > 1 without a launcher: start the N ranks as fresh child processes (torch.distributed.run) before this
    # process makes any GPU call, relay their output (rank 0 prints the JSON line) and leave with their exit code
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        import socket
        import subprocess

        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("POLAR_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse the multi-rank path with fewer GPUs than ranks
    if world > ndev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) visible; RCCL needs one GPU per rank "
                         "(POLAR_BENCH_BACKEND=gloo rehearses the multi-rank path on shared GPUs)")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op=None):
        """RCCL all-reduce on the device tensor; through a host copy when rehearsing over gloo."""
        op = op or dist.ReduceOp.SUM
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    import _native
    import functions as pcf
    from channel import Channel

    N, K = 1024, 512
    mask = pcf.degaBuild(N, K, 2.0)
    ctx = _native.default_ctx(local)
    code = _native.Code(ctx, mask)
    t0 = time.perf_counter()
    specialize_error = None
    try:
        code.specialize()  # per-code set-up: the SC decoder unrolled for this rate profile, built by the run-time compiler
    except _native.PolarError as e:  # the generic register kernel decodes the same bits, only slower
        specialize_error = str(e)
    specialize_s = time.perf_counter() - t0
    ch = Channel("BPSK", a.snr, "SNRb", K / N)
    B = a.frames
    seed = 2024
    frame0 = rank * B  # global frame index: ranks see disjoint frames

    llr = torch.empty((B, N), dtype=torch.float32, device=dev)
    info = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
    hat = torch.empty((B, code.Kw), dtype=torch.int32, device=dev)
    # one zeroed counter row per step (warm-up included): a step accumulates into its own row, so nothing has to be cleared
    # inside the timed region
    counter_rows = torch.zeros((a.warmup + a.steps + 1, 3), dtype=torch.int64, device=dev)
    step_no = [0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, ch.noisePower, seed, frame0, B)
    ctx.sync()
    gen_s = time.perf_counter() - t0

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        """One step; returns (seconds until this rank's decode + count had finished, seconds of the counter all-reduce)."""
        counters = counter_rows[step_no[0]]
        step_no[0] += 1
        s0 = time.perf_counter()
        # decode, store the extracted information words and count the errors against the transmitted ones, one pass
        code.sc_decode_count_dev(llr.data_ptr(), np.float32, info.data_ptr(), B, counters.data_ptr(), fast=True,
                                 info_hat_ptr=hat.data_ptr())
        ctx.sync()
        s1 = time.perf_counter()
        if world > 1:
            all_reduce(counters)
            torch.cuda.synchronize()  # the step is over when its counters are summed
        return s1 - s0, time.perf_counter() - s1

    def over_ranks(values):
        """{name: {"per_rank": [...], "min": .., "max": ..}} of this rank's float values, gathered over the ranks."""
        names = list(values)
        mine = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=dev)
        if world > 1:
            if backend == "nccl":
                rows = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(rows, mine)
            else:
                rows = [torch.empty_like(mine.cpu()) for _ in range(world)]
                dist.all_gather(rows, mine.cpu())
            tab = torch.stack([r.cpu() for r in rows])
        else:
            tab = mine.cpu()[None]
        return {k: {"per_rank": tab[:, i].tolist(), "min": float(tab[:, i].min()), "max": float(tab[:, i].max())} for i, k in enumerate(names)}

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    parts = [step() for _ in range(a.steps)]
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(tmax, dist.ReduceOp.MAX)
    my_dt, dt = dt, float(tmax.item())
    cnt = counter_rows[step_no[0] - 1].cpu().tolist()

    # live kernel timing: HIP events on the library's launch stream around the decode kernel alone
    ctx.enable_timing(True)
    kms = []
    for _ in range(max(2, a.steps)):
        code.sc_decode_dev(llr.data_ptr(), np.float32, hat.data_ptr(), B, fast=True)
        kms.append(code.last_launch()["last_ms"])
    ctx.enable_timing(False)
    launch = code.last_launch()
    k_ms = float(np.mean(kms))
    # where a step's time goes on every rank: a scaling loss shows as a slow rank (kernel), a long wait in the collective
    # (the slowest rank's decode ends later than mine) or neither (launch / host)
    per_rank = over_ranks({"kernel_ms": k_ms, "decode_and_count_ms_per_step": 1e3 * sum(p_[0] for p_ in parts) / a.steps,
                           "allreduce_ms_per_step": 1e3 * sum(p_[1] for p_ in parts) / a.steps,
                           "wall_ms_per_step": 1e3 * my_dt / a.steps})

    # secondary leg, all ranks: SCL L=8 on the first 2^20 frames of each rank's batch, counters all-reduced after every step;
    # one warm-up step, then SCL_STEPS timed ones between barriers; kernel time per rank from the HIP events
    scl = None
    SCL_STEPS = 3
    if not a.no_extras:
        nb = min(B, 1 << 20)
        c2rows = torch.zeros((SCL_STEPS + 1, 3), dtype=torch.int64, device=dev)
        ctx.enable_timing(True)  # the events only add the wait for the kernel, which the step needs anyway before it counts

        def scl_step(i):
            s0 = time.perf_counter()
            code.scl_decode_dev(llr.data_ptr(), np.float32, 8, hat.data_ptr(), nb)
            kms_ = code.last_launch()["last_ms"]
            code.count_errors_dev(info.data_ptr(), hat.data_ptr(), nb, c2rows[i].data_ptr())
            ctx.sync()
            s1 = time.perf_counter()
            if world > 1:
                all_reduce(c2rows[i])
                torch.cuda.synchronize()
            return kms_, s1 - s0, time.perf_counter() - s1

        scl_step(SCL_STEPS)  # warm-up
        barrier()
        t0 = time.perf_counter()
        sparts = [scl_step(i) for i in range(SCL_STEPS)]
        barrier()
        my_sdt = time.perf_counter() - t0
        ctx.enable_timing(False)
        sdt = torch.tensor([my_sdt], dtype=torch.float64, device=dev)
        if world > 1:
            all_reduce(sdt, dist.ReduceOp.MAX)
        sdt = float(sdt.item())
        c2 = c2rows[0].cpu().tolist()
        scl = {"frames_per_s": nb * world * SCL_STEPS / sdt, "info_mbps": nb * world * SCL_STEPS * K / sdt / 1e6,
               "frames_per_gpu": nb, "steps": SCL_STEPS, "warmup": 1, "ms_per_step": 1e3 * sdt / SCL_STEPS,
               "fer": c2[1] / max(c2[0], 1), "kernel": code.last_launch()["kernel"],
               "note": "wall time of SCL_STEPS x (decode + count + counter all-reduce) between barriers, max over ranks",
               "per_rank": over_ranks({"kernel_ms": float(np.mean([p_[0] for p_ in sparts])),
                                       "decode_and_count_ms_per_step": 1e3 * sum(p_[1] for p_ in sparts) / SCL_STEPS,
                                       "allreduce_ms_per_step": 1e3 * sum(p_[2] for p_ in sparts) / SCL_STEPS,
                                       "wall_ms_per_step": 1e3 * my_sdt / SCL_STEPS})}

    out = None
    if rank == 0:
        frames_total = B * world * a.steps
        fps = frames_total / dt
        alg_lds = lds_bytes_algorithmic(N) * B
        alg_hbm = (4 * N + K / 8) * B
        out = {
            "metric": "decoded info Mb/s, Polar(1024,512) SC, Eb/N0=2dB",
            "value": fps * K / 1e6,
            "unit": "Mb/s",
            "n_gpus": world,
            "devices": min(world, ndev),
            "collective": None if world == 1 else ("rccl" if backend == "nccl" else backend + " (rehearsal: ranks share GPUs)"),
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Polar(1024,512) dega dSNR=2.0, SC decode, {B:.0e} frames per GPU, LLRs resident in HBM".replace("e+0", "e"),
                       "frames_per_gpu": B, "snr_db": a.snr, "schedule": "fast (rate-0 / rate-1 subtrees short-cut)",
                       "parallelism": f"frames sharded over {world} GPU(s), counter all-reduce"},
            "frames_per_s": fps,
            "fer": cnt[1] / max(cnt[0], 1),
            "ber": cnt[2] / max(cnt[0] * K, 1),
            "counters": cnt,
            "input_gen_s": gen_s,
        "specialize_s": specialize_s,
        "specialize_error": specialize_error,
            "scl8": scl,
            "per_rank": per_rank,
            "roofline": {
                "bound": "hbm",
                "achieved": alg_hbm / (k_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": alg_hbm / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": measured_traffic(launch["kernel"], B)[0],
                "traffic_source": measured_traffic(launch["kernel"], B)[1],
                "kernel": launch["kernel"],
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_frame": 4 * N + K / 8,
                "lanes_per_frame": launch["lanes_per_frame"],
                "grid_blocks": launch["grid_blocks"],
                "lds_bytes_per_block": launch["lds_bytes_per_block"],
            },
            "roofline_valu": {
                "bound": "valu",
                "achieved": 3.5 * N * np.log2(N) * B / (k_ms * 1e-3) / 1e12,
                "peak": VALU_PEAK_TLOPS,
                "unit": "Tlane-op/s",
                "frac": 3.5 * N * np.log2(N) * B / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TLOPS,
                "algorithmic_lane_ops_per_frame": 3.5 * N * np.log2(N),
            },
        }

    if rank == 0 and world == 1 and not a.no_extras:
        extras = {}
        # plain (every-leaf) schedule, same kernel
        ctx.enable_timing(True)
        nb = min(B, 1 << 20)
        code.sc_decode_dev(llr.data_ptr(), np.float32, hat.data_ptr(), nb, fast=False)
        ms = code.last_launch()["last_ms"]
        extras["sc_plain_schedule"] = {"frames_per_s": nb / (ms * 1e-3), "info_mbps": nb * K / (ms * 1e-3) / 1e6,
                                       "lds_GBs_algorithmic": lds_bytes_algorithmic(N) * nb / (ms * 1e-3) / 1e9}
        # SCL L=8 on a slice of the same LLRs
        try:
            nb = min(B, 1 << 20)
            code.scl_decode_dev(llr.data_ptr(), np.float32, 8, hat.data_ptr(), nb)
            code.scl_decode_dev(llr.data_ptr(), np.float32, 8, hat.data_ptr(), nb)
            li = code.last_launch()
            c2 = torch.zeros(3, dtype=torch.int64, device=dev)
            code.count_errors_dev(info.data_ptr(), hat.data_ptr(), nb, c2.data_ptr())
            ctx.sync()
            c2 = c2.cpu().tolist()
            extras["scl8_kernel"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "info_mbps": nb * K / (li["last_ms"] * 1e-3) / 1e6,
                              "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb, "fer": c2[1] / nb,
                              "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, N, K, paths=8)}
        except _native.PolarError as e:
            extras["scl8_kernel"] = {"error": str(e)}
        ctx.enable_timing(False)
        # SCL L=8 CPU oracle on a bounded sample, for the ratio
        if not a.no_cpu:
            from oracle import oracle
            smp = llr[:64].cpu().numpy()
            t0 = time.perf_counter()
            for b_ in range(smp.shape[0]):
                oracle.scl_decode(smp[b_], mask, 8)
            extras["scl8_kernel"]["cpu_oracle_frames_per_s_1core"] = smp.shape[0] / (time.perf_counter() - t0)
            extras["scl8_kernel"]["reference_python_1core_frames_per_s"] = 3.05
        # the other configurations of the brief, timed on smaller batches: rate-profiled N=512 with L=32,
        # and the long code whose LLR tree does not fit a wavefront's registers (LDS-staged kernel)
        ctx.enable_timing(True)
        try:
            m512 = pcf.rmPolarBuild(512, 256, 2.0)
            c512 = _native.Code(ctx, m512)
            nb = 1 << 17
            l512 = torch.empty((nb, 512), dtype=torch.float32, device=dev)
            c512.gen_frames_dev(None, l512.data_ptr(), np.float32, ch.noisePower, seed, 0, nb)
            h512 = torch.empty((nb, c512.Kw), dtype=torch.int32, device=dev)
            for _ in range(2):
                c512.scl_decode_dev(l512.data_ptr(), np.float32, 32, h512.data_ptr(), nb)
            li = c512.last_launch()
            extras["scl32_rmpolar_512_256"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "kernel": li["kernel"], "kernel_ms": li["last_ms"],
                                               "frames": nb, "lds_bytes_per_block": li["lds_bytes_per_block"],
                                               "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, 512, 256, paths=32)}
            del l512, h512
            # list decoding beyond the eight-lanes-per-path kernel: (2048,1024), L = 8 and 32, on the two-lanes-per-path kernel
            c2k = _native.Code(ctx, pcf.degaBuild(2048, 1024, 2.0))
            nb = 1 << 15
            l2k = torch.empty((nb, 2048), dtype=torch.float32, device=dev)
            c2k.gen_frames_dev(None, l2k.data_ptr(), np.float32, ch.noisePower, seed, 0, nb)
            h2k = torch.empty((nb, c2k.Kw), dtype=torch.int32, device=dev)
            for L2 in (8, 32):
                for _ in range(2):
                    c2k.scl_decode_dev(l2k.data_ptr(), np.float32, L2, h2k.data_ptr(), nb)
                li = c2k.last_launch()
                extras["scl%d_2048_1024" % L2] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "kernel": li["kernel"], "kernel_ms": li["last_ms"],
                                                  "frames": nb, "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, 2048, 1024, paths=L2)}
            del l2k, h2k
            m16k = pcf.degaBuild(16384, 8192, 2.0)
            c16k = _native.Code(ctx, m16k)
            nb = 1 << 15
            l16k = torch.empty((nb, 16384), dtype=torch.float32, device=dev)
            c16k.gen_frames_dev(None, l16k.data_ptr(), np.float32, ch.noisePower, seed, 0, nb)
            h16k = torch.empty((nb, c16k.Kw), dtype=torch.int32, device=dev)
            for _ in range(2):  # the first call builds the long-code tables and sizes the slab
                c16k.sc_decode_dev(l16k.data_ptr(), np.float32, h16k.data_ptr(), nb, fast=True)
            li = c16k.last_launch()
            extras["sc_16384_8192"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "info_mbps": nb * 8192 / (li["last_ms"] * 1e-3) / 1e6,
                                       "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb,
                                       "lds_bytes_per_block": li["lds_bytes_per_block"],
                                       "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, 16384, 8192)}
            del l16k, h16k
        except _native.PolarError as e:
            extras["other_configs_error"] = str(e)
        # CRC-aided list decoding on the configuration cn/simulator.py:12-22 sets: 512 message bits + CRC-24, L = 8
        try:
            mca = pcf.degaBuild(1024, 512 + 24, 2.0)
            cca = _native.Code(ctx, mca)
            cca.set_crc(0x1800063)
            nb = 1 << 20
            ica = torch.empty((nb, cca.Kw), dtype=torch.int32, device=dev)
            hca = torch.empty_like(ica)
            lca = torch.empty((nb, 1024), dtype=torch.float32, device=dev)
            cca.gen_frames_dev(ica.data_ptr(), lca.data_ptr(), np.float32, ch.noisePower, seed, 0, nb, crc=True)
            cca.scl_decode_dev(lca.data_ptr(), np.float32, 8, hca.data_ptr(), nb, crc=True)
            li = cca.last_launch()
            c3 = torch.zeros(3, dtype=torch.int64, device=dev)
            cca.count_errors_dev(ica.data_ptr(), hca.data_ptr(), nb, c3.data_ptr(), crc=True)
            ctx.sync()
            c3 = c3.cpu().tolist()
            extras["ca_scl8_1024_512_crc24"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "info_mbps": nb * 512 / (li["last_ms"] * 1e-3) / 1e6,
                                                "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb, "fer": c3[1] / nb,
                                                "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, 1024, 536, paths=8)}
            # progressive bit flipping on the same frames: fast SC on all, CRC test, flip search on the failures
            cca.set_reliability(pcf.GA(1024, 536, 2.0))
            for T, nb in ((1, 1 << 20), (2, 1 << 18), (3, 1 << 15)):  # a failing frame costs ~10^2 SC runs at T = 1, ~10^4 at T = 2, ~10^5 at T = 3 (the level cn/simulator.py:33 sets)
                rn = torch.zeros(nb, dtype=torch.int32, device=dev)
                cca.pbf_decode_dev(lca.data_ptr(), np.float32, T, hca.data_ptr(), nb, runs_ptr=rn.data_ptr())
                ctx.sync()
                t0 = time.perf_counter()
                cca.pbf_decode_dev(lca.data_ptr(), np.float32, T, hca.data_ptr(), nb, runs_ptr=rn.data_ptr())
                ctx.sync()
                pdt = time.perf_counter() - t0
                c3 = torch.zeros(3, dtype=torch.int64, device=dev)
                cca.count_errors_dev(ica.data_ptr(), hca.data_ptr(), nb, c3.data_ptr(), crc=True)
                ctx.sync()
                extras["pbf_T%d_1024_512_crc24" % T] = {"frames_per_s": nb / pdt, "info_mbps": nb * 512 / pdt / 1e6, "frames": nb,
                                                        "fer": c3.cpu().tolist()[1] / nb, "mean_sc_runs": float(rn.float().mean().item()),
                                                        "max_sc_runs": int(rn.max().item()), "search_kernel_ms": cca.last_launch()["last_ms"],
                                                        "kernel": cca.last_launch()["kernel"]}
                # roofline of the search: the lane operations of all its SC runs (the runs the reference would make: sum of the
                # per-frame counts) over the wall time of the call; HBM bytes: every frame's LLRs once
                runs_total = int(rn.sum().item())
                extras["pbf_T%d_1024_512_crc24" % T]["roofline"] = leg_roofline(
                    cca.last_launch()["kernel"], pdt * 1e3, runs_total, 1024, 536, hbm_bytes_per_frame=(4 * 1024 + 67) * nb / runs_total)
            del ica, hca, lca
        except _native.PolarError as e:
            extras["ca_scl8_1024_512_crc24"] = {"error": str(e)}
        # the reference's own arithmetic: fp64 SC with every leaf visited (decision LLRs available) and fp64 list decoding
        # with the whole list returned -- what PolarCode.scDecoder / sclDecoder run by default (bit-exact mode)
        try:
            nb = 1 << 16
            l64 = llr[:nb].double()
            h64 = torch.empty((nb, code.Kw), dtype=torch.int32, device=dev)
            for _ in range(2):
                code.sc_decode_dev(l64.data_ptr(), np.float64, h64.data_ptr(), nb, fast=False)
            li = code.last_launch()
            extras["sc_fp64_parity_mode"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "info_mbps": nb * K / (li["last_ms"] * 1e-3) / 1e6,
                                             "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb,
                                             "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, N, K, hbm_bytes_per_frame=8 * N + K / 8)}
            pm64 = torch.empty((nb, 8), dtype=torch.float64, device=dev)
            for _ in range(2):
                code.scl_decode_dev(l64.data_ptr(), np.float64, 8, h64.data_ptr(), nb, pm_ptr=pm64.data_ptr())
            li = code.last_launch()
            extras["scl8_fp64_parity_mode"] = {"frames_per_s": nb / (li["last_ms"] * 1e-3), "info_mbps": nb * K / (li["last_ms"] * 1e-3) / 1e6,
                                               "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb,
                                               "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, N, K, paths=8, hbm_bytes_per_frame=8 * N + K / 8)}
            del l64, h64, pm64
            # the rate-profiled (512,256) code with L = 32 in fp64, sorted metrics as an output: two lanes per path
            nb = 1 << 16
            c512d = _native.Code(ctx, pcf.rmPolarBuild(512, 256, 2.0))
            l512d = torch.empty((nb, 512), dtype=torch.float64, device=dev)
            c512d.gen_frames_dev(None, l512d.data_ptr(), np.float64, ch.noisePower, seed, 0, nb)
            h512d = torch.empty((nb, c512d.Kw), dtype=torch.int32, device=dev)
            pm512d = torch.empty((nb, 32), dtype=torch.float64, device=dev)
            for _ in range(2):
                c512d.scl_decode_dev(l512d.data_ptr(), np.float64, 32, h512d.data_ptr(), nb, pm_ptr=pm512d.data_ptr())
            li = c512d.last_launch()
            extras["scl32_fp64_parity_mode_rmpolar_512_256"] = {
                "frames_per_s": nb / (li["last_ms"] * 1e-3), "kernel": li["kernel"], "kernel_ms": li["last_ms"], "frames": nb,
                "roofline": leg_roofline(li["kernel"], li["last_ms"], nb, 512, 256, paths=32, hbm_bytes_per_frame=8 * 512 + 256 / 8)}
            del l512d, h512d, pm512d
        except _native.PolarError as e:
            extras["fp64_error"] = str(e)
        ctx.enable_timing(False)
        # the Monte-Carlo loop (draw, encode, channel, decode, count), nothing resident beforehand: as one kernel without an
        # LLR buffer (what polar_mc_run does for a specialised code), and as generator -> buffer -> decoder
        nmc = 1 << 24
        for name, staged in (("mc_fused_sc", False), ("mc_pipeline_sc", True)):
            if staged:
                os.environ["POLAR_MC_KERNEL"] = "staged"
                os.environ["POLAR_EXPERIMENT"] = "1"  # the library reads its kernel switches only with this set
            try:
                code.mc_run(_native.DEC_SC, 1, np.float32, ch.noisePower, seed, 0, (1 << 20) + 4096)  # warm-up, sizes the buffers
                t0 = time.perf_counter()
                mc = code.mc_run(_native.DEC_SC, 1, np.float32, ch.noisePower, seed, 0, nmc)
                mdt = time.perf_counter() - t0
                extras[name] = {"frames_per_s": nmc / mdt, "info_mbps": nmc * K / mdt / 1e6, "frames": nmc, "fer": int(mc[1]) / int(mc[0]),
                                "counters": [int(v) for v in mc], "kernel": code.last_launch()["kernel"]}
                if not staged:  # the whole loop is one kernel: its roofline from the wall time of the call (no HBM traffic but the counters)
                    extras[name]["roofline"] = leg_roofline(code.last_launch()["kernel"], mdt * 1e3, nmc, N, K, hbm_bytes_per_frame=0.0)
            finally:
                os.environ.pop("POLAR_MC_KERNEL", None)
                os.environ.pop("POLAR_EXPERIMENT", None)
        out["extras"] = extras

    if rank == 0 and world == 1 and not a.no_cpu:
        threads = min(16, os.cpu_count() or 1)
        sample = llr[: 8192 * threads].cpu().numpy()
        fps_cpu, secs, done = cpu_baseline(mask, sample, threads)
        out["cpu_baseline"] = {"value": fps_cpu * K / 1e6, "unit": "Mb/s", "cores": threads, "kind": "port",
                               "sample": f"{done} frame decodes ({sample.shape[0]} frames of the same LLR batch, repeated) by the oracle SC decoder "
                                         f"(C restatement of scDecoder, fp32), {secs:.1f} s on {threads} threads",
                               "frames_per_s": fps_cpu,
                               "reference_python_1core_frames_per_s": 26.4}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
