"""Error returns that happen half-way through a call leave nothing behind: each path is triggered once through the
library's failure-injection switch (POLAR_FAULT_AT, read only with POLAR_EXPERIMENT=1) and the next ordinary call must work
and give the result it gave before.  One process, one trigger per path."""
import numpy as np
import pytest
import torch

import _native
import functions as pcf
from channel import Channel

pytestmark = pytest.mark.gpu


@pytest.fixture
def fault(monkeypatch):
    monkeypatch.setenv("POLAR_EXPERIMENT", "1")

    def at(site):
        if site is None:
            monkeypatch.delenv("POLAR_FAULT_AT", raising=False)
        else:
            monkeypatch.setenv("POLAR_FAULT_AT", site)

    yield at
    monkeypatch.delenv("POLAR_FAULT_AT", raising=False)


def test_ctx_create_failure_gives_back_what_it_made(fault):
    fault("ctx_create")
    with pytest.raises(_native.PolarError):
        _native.Context(0)
    fault(None)
    ctx = _native.Context(0)  # and the device is still usable
    assert ctx.num_cus > 0
    ctx.close()


def test_comm_init_and_allreduce_failures_leave_rccl_usable(fault):
    ndev = torch.cuda.device_count()
    fault("comm_init")
    with pytest.raises(_native.PolarError):
        _native.Comm(list(range(ndev)))
    fault(None)
    comm = _native.Comm(list(range(ndev)))
    bufs = [torch.tensor([5 + d, 1, 2], dtype=torch.int64, device="cuda:%d" % d) for d in range(ndev)]
    torch.cuda.synchronize()
    fault("allreduce")
    with pytest.raises(_native.PolarError):
        comm.allreduce_counters([b.data_ptr() for b in bufs], 3)
    fault(None)
    # the group opened by the failed call was closed: this collective runs and sums
    comm.allreduce_counters([b.data_ptr() for b in bufs], 3)
    assert bufs[0].cpu().tolist() == [sum(5 + d for d in range(ndev)), ndev, 2 * ndev]
    comm.close()


def test_mc_run_failure_leaves_nothing_in_flight(fault, monkeypatch):
    ctx = _native.default_ctx(0)
    code = _native.Code(ctx, pcf.degaBuild(256, 128, 2.0))
    n0 = Channel("BPSK", 2.0, "SNRb", 0.5).noisePower
    monkeypatch.setenv("POLAR_MC_CHUNK", "2048")  # several chunks, two buffers, two streams
    want = code.mc_run(_native.DEC_SCL, 4, np.float32, n0, 11, 0, 10000)
    fault("mc_run")
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SCL, 4, np.float32, n0, 11, 0, 10000)
    fault(None)
    again = code.mc_run(_native.DEC_SCL, 4, np.float32, n0, 11, 0, 10000)
    assert again.tolist() == want.tolist()


def test_list_decoder_without_a_free_slab_reports_instead_of_spinning(fault):
    """The (1024, L <= 8, fp32) register list decoder keeps a wide level in slabs of global memory that its workgroups
    claim from a bitmap.  With fewer slabs than resident workgroups (injected) the search gives up after a bounded number
    of steps, and the next synchronising call returns the error."""
    ctx = _native.default_ctx(0)
    N, K, L, B = 1024, 512, 8, 8192
    code = _native.Code(ctx, pcf.degaBuild(N, K, 2.0))
    llr = torch.empty((B, N), dtype=torch.float32, device="cuda")
    info = torch.empty((B, code.Kw), dtype=torch.int32, device="cuda")
    code.gen_frames_dev(info.data_ptr(), llr.data_ptr(), np.float32, Channel("BPSK", 2.0, "SNRb", 0.5).noisePower, 3, 0, B)
    good = torch.zeros((B, code.Kw), dtype=torch.int32, device="cuda")
    code.scl_decode_dev(llr.data_ptr(), np.float32, L, good.data_ptr(), B)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel<10>"), code.last_launch()
    ctx.sync()
    fault("scl_slab")
    hat = torch.zeros((B, code.Kw), dtype=torch.int32, device="cuda")
    code.scl_decode_dev(llr.data_ptr(), np.float32, L, hat.data_ptr(), B)
    with pytest.raises(_native.PolarError, match="slab"):
        ctx.sync()
    fault(None)
    code.scl_decode_dev(llr.data_ptr(), np.float32, L, hat.data_ptr(), B)
    ctx.sync()
    assert torch.equal(hat, good)
