"""Monte-Carlo driver: sharding, stopping rule and the counter all-reduce, exercised on CPU with the
oracle as the frame runner (single process and a 2-rank gloo job)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import functions as pcf
import simulator
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeCode:
    def __init__(self, N, K):
        self.codewordLength, self.infoLen = N, K
        self.polarMask = pcf.degaBuild(N, K, 2.0)
        self.decoderType, self.L, self.snrType, self.modu = "sc", 1, "SNRb", "BPSK"


def oracle_runner(pc, log=None):
    def run(N0, seed, frame0, nframes):
        if log is not None:
            log.append((seed, frame0, nframes))
        return oracle.mc_run(pc.polarMask, 0, 1, 0, N0, seed, frame0, nframes)

    return run


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 100, 10 ** 7 + 3):
        for world in (1, 2, 3, 8):
            spans = [simulator.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_fixed_frames_single_rank_and_batch_invariance():
    pc = FakeCode(64, 32)
    log = []
    r1 = simulator.simulate(pc, [1.0, 3.0], frames=700, batch=256, seed=5, runner=oracle_runner(pc, log), verbose=False)
    assert r1.frames == [700, 700]
    assert [c[2] for c in log[:3]] == [256, 256, 188] and [c[0] for c in log[:4]] == [5, 5, 5, 6]
    r2 = simulator.simulate(pc, [1.0, 3.0], frames=700, batch=700, seed=5, runner=oracle_runner(pc), verbose=False)
    assert r1.frameErrors == r2.frameErrors and r1.bitErrors == r2.bitErrors
    assert r1.fer[0] > r1.fer[1] > 0
    assert r1.ber[0] == r1.bitErrors[0] / (700 * 32)


def test_stop_rule_matches_reference_condition():
    """fer >= errCnt and t >= minFrames-1, tested after every round (cn/simulator.py:257)."""
    pc = FakeCode(64, 32)
    r = simulator.simulate(pc, [0.0], frames=None, errCnt=30, minFrames=500, batch=100, seed=1, runner=oracle_runner(pc), verbose=False)
    assert r.frames[0] >= 500 and r.frameErrors[0] >= 30 and r.frames[0] % 100 == 0
    # a rare-error point keeps running until enough errors are seen (bounded by maxFrames)
    r = simulator.simulate(pc, [9.0], frames=None, errCnt=5, minFrames=100, maxFrames=400, batch=100, seed=1, runner=oracle_runner(pc), verbose=False)
    assert r.frames[0] == 400


def test_rejects_unknown_decoder():
    pc = FakeCode(64, 32)
    pc.decoderType = "fano"
    with pytest.raises(NotImplementedError):
        simulator.simulate(pc, [1.0], frames=10, runner=oracle_runner(pc), verbose=False)


WORKER = r'''
import os, sys, json
sys.path[:0] = [%(root)r, %(root)r + "/polar-codes_amd", %(root)r + "/tests"]
import torch.distributed as dist
import simulator
from test_simulator import FakeCode, oracle_runner
dist.init_process_group("gloo")
pc = FakeCode(64, 32)
res = simulator.simulate(pc, [1.0, 2.5], frames=901, batch=200, seed=11, runner=oracle_runner(pc), verbose=False)
stop = simulator.simulate(pc, [0.0], frames=None, errCnt=40, minFrames=300, batch=100, seed=3, runner=oracle_runner(pc), verbose=False)
open(os.path.join(%(out)r, "rank%%d.json" %% dist.get_rank()), "w").write(json.dumps([res.frames, res.frameErrors, res.bitErrors, stop.frames, stop.frameErrors]))
dist.destroy_process_group()
'''


def test_two_rank_gloo_job_gives_single_rank_totals(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r0, r1 = (json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in (0, 1))
    assert r0 == r1, "every rank reports the all-reduced totals"
    pc = FakeCode(64, 32)
    single = simulator.simulate(pc, [1.0, 2.5], frames=901, batch=901, seed=11, runner=oracle_runner(pc), verbose=False)
    assert r0[0] == single.frames and r0[1] == single.frameErrors and r0[2] == single.bitErrors
    # stop-rule mode: both ranks ran the same number of rounds; totals are multiples of world*batch
    assert r0[3][0] % 200 == 0 and r0[3][0] >= 300 and r0[4][0] >= 40


RESUME_WORKER = r'''
import os, sys, json
sys.path[:0] = [%(root)r, %(root)r + "/polar-codes_amd", %(root)r + "/tests"]
import torch.distributed as dist
import simulator
from test_simulator import FakeCode, oracle_runner
dist.init_process_group("gloo")
rank = dist.get_rank()
pc = FakeCode(64, 32)
# each rank is given its own path: only rank 0's file exists, as on ranks without a shared file system
ck = os.path.join(%(out)r, "ck_rank%%d.json" %% rank)
kw = dict(frames=None, errCnt=30, minFrames=800, batch=100, seed=4, verbose=False)
phase = %(phase)r
if phase == "first":
    calls = [0]
    inner = oracle_runner(pc)
    def dying(N0, seed, f0, n):
        calls[0] += 1
        if calls[0] > 5:          # both ranks die in the same round of the second point
            raise RuntimeError("killed")
        return inner(N0, seed, f0, n)
    try:
        simulator.simulate(pc, [0.0, 1.0], runner=dying, checkpoint=ck, **kw)
    except RuntimeError:
        pass
else:
    log = []
    res = simulator.simulate(pc, [0.0, 1.0], runner=oracle_runner(pc, log), checkpoint=ck, **kw)
    open(os.path.join(%(out)r, "resume%%d.json" %% rank), "w").write(json.dumps([res.frames, res.frameErrors, res.bitErrors, len(log)]))
dist.destroy_process_group()
'''


def test_two_rank_resume_takes_the_checkpoint_from_rank_0(tmp_path):
    """Only rank 0 can see the checkpoint file (every rank is handed a path of its own, and only rank 0 ever writes one):
    the state found by rank 0 is broadcast, so both ranks skip the finished point, continue the interrupted one at the same
    round, enter the same all-reduces and end with the totals of an uninterrupted single-rank sweep."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for phase, port in (("first", "29541"), ("second", "29542")):
        script = tmp_path / ("resume_%s.py" % phase)
        script.write_text(RESUME_WORKER % {"root": ROOT, "out": str(tmp_path), "phase": phase})
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                              "127.0.0.1", "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert (tmp_path / "ck_rank0.json").exists() and not (tmp_path / "ck_rank1.json").exists()
    r0, r1 = (json.loads((tmp_path / ("resume%d.json" % r)).read_text()) for r in (0, 1))
    assert r0[:3] == r1[:3], "every rank reports the all-reduced totals"
    assert r0[3] == r1[3], "both ranks ran the same number of batches after the resume"
    pc = FakeCode(64, 32)
    log = []
    # the uninterrupted sweep with the same frame layout: one rank, rounds of world * batch frames
    single = simulator.simulate(pc, [0.0, 1.0], frames=None, errCnt=30, minFrames=800, batch=200, seed=4, runner=oracle_runner(pc, log),
                                verbose=False)
    assert r0[0] == single.frames and r0[1] == single.frameErrors and r0[2] == single.bitErrors
    assert 1 <= r0[3] < len(log)  # fewer rounds than a sweep from the start


def test_bench_gpus_flag_starts_that_many_ranks():
    """`bench.py --gpus 2` without a launcher starts two ranks itself, before any GPU call of its own; without a GPU every
    rank stops with the same message and the exit code is passed on."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (the GPU run is test_bench_two_rank_rehearsal)")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu",
                          "--no-extras"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") == 2, out.stderr[-2000:]


# ---- GPU: the real frame runner (polar_mc_run) under torch.distributed ----------------------------------------------
NATIVE_WORKER = r'''
import os, sys, json
sys.path[:0] = [%(root)r, %(root)r + "/polar-codes_amd"]
import numpy as np
import torch
import torch.distributed as dist
os.environ["POLAR_DEVICE"] = "0"          # every rank of this rehearsal shares GPU 0
backend = %(backend)r
if backend == "nccl":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
import simulator
from polarCode import PolarCode
out = {}
for name, N, K, dec, L, frames in (("sc", 1024, 512, "sc", 1, 20001), ("scl", 256, 128, "scl", 4, 3001)):
    pc = PolarCode(N, K, "dega", 2.0, listSize=L)
    pc.decoderType, pc.snrType, pc.modu = dec, "SNRb", "BPSK"
    res = simulator.simulate(pc, [1.0, 2.0], frames=frames, batch=4096, seed=7, verbose=False)
    out[name] = [res.frames, res.frameErrors, res.bitErrors, pc.native.last_launch()["kernel"]]
pc = PolarCode(128, 64, "dega", 2.0)
pc.decoderType, pc.snrType, pc.modu = "sc", "SNRb", "BPSK"
stop = simulator.simulate(pc, [1.0], frames=None, errCnt=50, minFrames=3000, batch=1000, seed=3, verbose=False)
out["stop"] = [stop.frames, stop.frameErrors]
out["backend"] = dist.get_backend()
open(os.path.join(%(out)r, "rank%%d.json" %% dist.get_rank()), "w").write(json.dumps(out))
dist.destroy_process_group()
'''


def _run_native_job(tmp_path, nproc, backend, port):
    import json

    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER % {"root": ROOT, "out": str(tmp_path), "backend": backend})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(nproc)]


def _single_process_totals():
    from polarCode import PolarCode

    out = {}
    for name, N, K, dec, L, frames in (("sc", 1024, 512, "sc", 1, 20001), ("scl", 256, 128, "scl", 4, 3001)):
        pc = PolarCode(N, K, "dega", 2.0, listSize=L)
        pc.decoderType, pc.snrType, pc.modu = dec, "SNRb", "BPSK"
        res = simulator.simulate(pc, [1.0, 2.0], frames=frames, batch=frames, seed=7, verbose=False)
        out[name] = [res.frames, res.frameErrors, res.bitErrors]
    return out


@pytest.mark.gpu
def test_rccl_one_rank_job_runs_the_device_all_reduce(tmp_path):
    """A fresh process initialises RCCL (backend "nccl", world size 1) and runs simulate() with the GPU library as the
    frame runner: the counters go through a device-tensor all-reduce and equal the run without torch.distributed."""
    (r0,) = _run_native_job(tmp_path, 1, "nccl", 29541)
    assert r0["backend"] == "nccl"
    single = _single_process_totals()
    for name in ("sc", "scl"):
        assert r0[name][:3] == single[name], name
    assert r0["sc"][3].startswith("sc_static_mc_kernel")


@pytest.mark.gpu
def test_two_rank_job_with_the_gpu_frame_runner(tmp_path):
    """Two ranks (both on GPU 0, counters reduced over gloo) shard the frame range and run polar_mc_run on their shares:
    the totals equal the single-process run, whatever the split and the batch size."""
    r0, r1 = _run_native_job(tmp_path, 2, "gloo", 29543)
    assert r0 == r1
    single = _single_process_totals()
    for name in ("sc", "scl"):
        assert r0[name][:3] == single[name], name
    assert r0["stop"][0][0] % 2000 == 0 and r0["stop"][0][0] >= 3000 and r0["stop"][1][0] >= 50


@pytest.mark.gpu
def test_bench_two_rank_rehearsal():
    """bench.py --gpus 2 starts its two ranks itself; on a one-GPU box they share the GPU and reduce over gloo."""
    import json

    import torch

    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["POLAR_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames",
                          "200000", "--no-cpu", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["counters"][0] == 400000 and line["value"] > 0


def test_checkpoint_resume_gives_the_uninterrupted_totals(tmp_path):
    """simulate(checkpoint=...): a sweep that dies in the middle of a point (here: the frame runner raises) continues from
    the file with the rounds it has not run, and ends with the totals of an uninterrupted sweep; finished points are not
    run again; a file of another sweep is refused."""
    pc = FakeCode(64, 32)
    ck = str(tmp_path / "sweep.json")
    kw = dict(frames=None, errCnt=60, minFrames=900, batch=100, seed=4, verbose=False)
    whole = simulator.simulate(pc, [0.0, 1.0], runner=oracle_runner(pc), **kw)

    calls = []

    def dying(N0, seed, frame0, nframes):
        calls.append((seed, frame0))
        if len(calls) == 14:
            raise RuntimeError("power cut")
        return oracle.mc_run(pc.polarMask, 0, 1, 0, N0, seed, frame0, nframes)

    with pytest.raises(RuntimeError):
        simulator.simulate(pc, [0.0, 1.0], runner=dying, checkpoint=ck, **kw)
    log = []
    again = simulator.simulate(pc, [0.0, 1.0], runner=oracle_runner(pc, log), checkpoint=ck, **kw)
    assert again.frames == whole.frames and again.frameErrors == whole.frameErrors and again.bitErrors == whole.bitErrors
    assert len(log) + 13 == sum(f // 100 for f in whole.frames), "only the rounds that were missing have run"
    log.clear()
    third = simulator.simulate(pc, [0.0, 1.0], runner=oracle_runner(pc, log), checkpoint=ck, **kw)
    assert not log and third.frameErrors == whole.frameErrors
    with pytest.raises(ValueError):
        simulator.simulate(pc, [0.0, 1.0], runner=oracle_runner(pc), checkpoint=ck, **dict(kw, seed=5))


@pytest.mark.gpu
def test_c_abi_counter_allreduce_over_rccl():
    """polar_comm_init_all / polar_allreduce_counters: RCCL communicators made inside one process over the visible GPUs
    (one on the test box), counters of polar_count_errors summed in place."""
    import torch

    import _native

    ndev = torch.cuda.device_count()
    comm = _native.Comm(list(range(ndev)))
    bufs = []
    for d in range(ndev):
        with torch.cuda.device(d):
            bufs.append(torch.tensor([100 + d, 7 * (d + 1), 3], dtype=torch.int64, device="cuda:%d" % d))
    torch.cuda.synchronize()
    comm.allreduce_counters([b.data_ptr() for b in bufs], 3)
    want = [sum(100 + d for d in range(ndev)), sum(7 * (d + 1) for d in range(ndev)), 3 * ndev]
    for b in bufs:
        assert b.cpu().tolist() == want
    comm.close()
    with pytest.raises(_native.PolarError):
        _native.Comm(list(range(ndev + 1)))
    with pytest.raises(_native.PolarError):
        _native.Comm([0, 0][: max(2, ndev)]) if ndev >= 2 else _native.Comm([5])
