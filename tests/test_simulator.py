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
