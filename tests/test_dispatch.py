"""The dispatch table of the library (csrc/host_tables.cpp): every decoder entry point asks it which kernel family serves a
call.  Host-only: the table is read through polar_dispatch_select / polar_dispatch_table, no GPU is needed.  Checked here:
the five benchmark configurations map to the register kernels that were written for them, in both arithmetics and for
every output; holes are refused with POLAR_EUNSUPPORTED and a message; the experiment switches move a call to the kernel
they name."""
import numpy as np
import pytest

import _native
from _native import (DQ_CRC, DQ_FAST, DQ_FLIP_LIST, DQ_FORCE_GENERIC, DQ_FORCE_LDS, DQ_SPECIALISED, DQ_SPLIT, OP_MC_SC, OP_PBF, OP_SC, OP_SCL,
                     OUT_DECLLR, OUT_INFO, OUT_PM, OUT_UHAT)

F32, F64 = np.float32, np.float64


def kernel(*a, **kw):
    return _native.dispatch_select(*a, **kw)[1]


def test_table_is_listed_and_ordered():
    tab = _native.dispatch_table()
    assert len(tab) >= 15
    assert {r["op"] for r in tab} == {"SC", "SCL", "PBF", "MC_SC"}
    # within an operation the register kernels come before the LDS-staged catch-all
    for op in ("SC", "SCL", "PBF"):
        names = [r["kernel"] for r in tab if r["op"] == op]
        assert any("decode_kernel" in n for n in names[-2:]), names
        assert not any("decode_kernel" in n for n in names[:2]), names


# (config, op, N, K, dtype, L, outputs, flags) -> kernel prefix
CONFIGS = [
    # C1 (128,64) SC: the reference's CPU-runnable case; fp64 with u_hat and decision LLRs is what PolarCode.scDecoder runs
    ("C1 fp64 scDecoder", OP_SC, 128, 64, F64, 0, OUT_INFO | OUT_UHAT | OUT_DECLLR, 0, "sc_reg_full_kernel<double"),
    ("C1 fp32 fast", OP_SC, 128, 64, F32, 0, OUT_INFO, DQ_FAST, "sc_reg_kernel"),
    ("C1 fp32 specialised", OP_SC, 128, 64, F32, 0, OUT_INFO, DQ_FAST | DQ_SPECIALISED, "sc_static_kernel"),
    # C2 (1024,512) SC, 10^7 frames: the headline
    ("C2 fp32 specialised", OP_SC, 1024, 512, F32, 0, OUT_INFO, DQ_FAST | DQ_SPECIALISED, "sc_static_kernel"),
    ("C2 fp32 generic", OP_SC, 1024, 512, F32, 0, OUT_INFO, DQ_FAST, "sc_reg_kernel"),
    ("C2 fp32 every leaf", OP_SC, 1024, 512, F32, 0, OUT_INFO, 0, "sc_reg_kernel"),
    ("C2 fp32 u_hat", OP_SC, 1024, 512, F32, 0, OUT_INFO | OUT_UHAT, 0, "sc_reg_full_kernel<float"),
    ("C2 fp64", OP_SC, 1024, 512, F64, 0, OUT_INFO, 0, "sc_reg_full_kernel<double"),
    ("C2 fp64 decision LLRs", OP_SC, 1024, 512, F64, 0, OUT_INFO | OUT_DECLLR, 0, "sc_reg_full_kernel<double"),
    ("C2 Monte-Carlo loop", OP_MC_SC, 1024, 512, F32, 0, 0, DQ_FAST | DQ_SPECIALISED, "sc_static_mc_kernel"),
    ("C2 Monte-Carlo loop, not specialised", OP_MC_SC, 1024, 512, F32, 0, 0, DQ_FAST, "gen_frames_kernel"),
    # C3 (1024,512) SCL L=8
    ("C3 fp32", OP_SCL, 1024, 512, F32, 8, OUT_INFO, 0, "scl_reg_kernel"),
    ("C3 fp32 CRC", OP_SCL, 1024, 536, F32, 8, OUT_INFO, DQ_CRC, "scl_reg_kernel"),
    ("C3 fp64 whole list", OP_SCL, 1024, 512, F64, 8, OUT_INFO | OUT_UHAT | OUT_PM, 0, "scl_reg_kernel"),
    ("C3 fp32 whole list", OP_SCL, 1024, 512, F32, 8, OUT_INFO | OUT_UHAT | OUT_PM, 0, "scl_reg_kernel"),
    # C4 N=512 rate-profiled, L=32
    ("C4 fp32", OP_SCL, 512, 256, F32, 32, OUT_INFO, 0, "scl_wide_kernel"),
    ("C4 fp64 whole list", OP_SCL, 512, 256, F64, 32, OUT_INFO | OUT_UHAT | OUT_PM, 0, "scl_wide_kernel"),
    ("C4 fp64 L=16", OP_SCL, 512, 256, F64, 16, OUT_INFO | OUT_PM, 0, "scl_wide_kernel"),
    # C5 (16384,8192) SC
    ("C5 fp32 fast", OP_SC, 16384, 8192, F32, 0, OUT_INFO, DQ_FAST, "sc_big_kernel"),
    ("C5 fp32 every leaf", OP_SC, 16384, 8192, F32, 0, OUT_INFO, 0, "sc_decode_kernel<float"),
    ("C5 fp64", OP_SC, 16384, 8192, F64, 0, OUT_INFO | OUT_DECLLR, 0, "sc_decode_kernel<double"),
    # N = 2048, 4096: lists of 8 and more on the two-lanes-per-path kernel, shorter lists LDS-staged
    ("SCL-8 at 2048", OP_SCL, 2048, 1024, F32, 8, OUT_INFO, 0, "scl_wide_kernel"),
    ("SCL-32 at 2048, fp64 lists", OP_SCL, 2048, 1024, F64, 32, OUT_INFO | OUT_UHAT | OUT_PM, 0, "scl_wide_kernel"),
    ("SCL-4 at 2048", OP_SCL, 2048, 1024, F32, 4, OUT_INFO, 0, "scl_wide_kernel"),
    ("SCL-8 at 4096", OP_SCL, 4096, 2048, F32, 8, OUT_INFO, 0, "scl_wide_kernel"),
    ("SCL-4 at 4096", OP_SCL, 4096, 2048, F64, 4, OUT_INFO, 0, "scl_wide_kernel"),
    ("SCL-1 at 4096", OP_SCL, 4096, 2048, F64, 1, OUT_INFO, 0, "scl_decode_kernel"),
    # flip search on the configuration cn/simulator.py sets (512 message bits + CRC-24)
    ("PBF one flip", OP_PBF, 1024, 536, F32, 1, OUT_INFO, DQ_CRC, "pbf_reg_kernel"),
    ("PBF two flips in pieces", OP_PBF, 1024, 536, F64, 2, OUT_INFO, DQ_CRC | DQ_SPLIT, "pbf_reg_kernel"),
    ("PBF two flips, no workspace", OP_PBF, 1024, 536, F32, 2, OUT_INFO, DQ_CRC, "pbf_decode_kernel"),
    ("PBF three flips in pieces", OP_PBF, 1024, 536, F32, 3, OUT_INFO, DQ_CRC | DQ_SPLIT, "pbf_reg_kernel"),
    ("PBF three flips, no workspace", OP_PBF, 1024, 536, F32, 3, OUT_INFO, DQ_CRC, "pbf_decode_kernel"),
    ("PBF three flips, long code", OP_PBF, 2048, 1048, F32, 3, OUT_INFO, DQ_CRC | DQ_SPLIT, "pbf_decode_kernel"),
    ("scDecoder(sequenceU)", OP_PBF, 1024, 536, F64, 1, OUT_INFO | OUT_UHAT | OUT_DECLLR, DQ_FLIP_LIST, "pbf_decode_kernel"),
]


@pytest.mark.parametrize("what,op,N,K,dtype,L,outputs,flags,want", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_benchmark_configurations_map_to_their_kernels(what, op, N, K, dtype, L, outputs, flags, want):
    got = kernel(op, N, K, dtype, L, outputs, flags)
    assert got.startswith(want), (what, got)


def test_experiment_switches_move_a_call():
    assert kernel(OP_SC, 1024, 512, F32, 0, OUT_INFO, DQ_FAST | DQ_SPECIALISED | DQ_FORCE_GENERIC).startswith("sc_reg_kernel")
    assert kernel(OP_SC, 1024, 512, F32, 0, OUT_INFO, DQ_FAST | DQ_SPECIALISED | DQ_FORCE_LDS).startswith("sc_decode_kernel")
    assert kernel(OP_SCL, 1024, 512, F32, 8, OUT_INFO, DQ_FORCE_LDS).startswith("scl_decode_kernel")
    assert kernel(OP_SCL, 512, 256, F64, 32, OUT_INFO, DQ_FORCE_LDS).startswith("scl_decode_kernel")
    assert kernel(OP_PBF, 1024, 536, F32, 1, OUT_INFO, DQ_CRC | DQ_FORCE_LDS).startswith("pbf_decode_kernel")
    assert kernel(OP_MC_SC, 1024, 512, F32, 0, 0, DQ_FAST | DQ_SPECIALISED | DQ_FORCE_GENERIC).startswith("gen_frames_kernel")


def test_edges_of_the_register_kernels():
    # K = 0 and K = N: the root is a rate-0 / rate-1 node, which the fp32 information-bit kernels do not take
    assert kernel(OP_SC, 256, 0, F32, 0, OUT_INFO, DQ_FAST).startswith("sc_decode_kernel")
    assert kernel(OP_SC, 256, 256, F32, 0, OUT_INFO, DQ_FAST).startswith("sc_decode_kernel")
    assert kernel(OP_SC, 256, 256, F64, 0, OUT_INFO, 0).startswith("sc_reg_full_kernel")
    # lengths below the register layouts
    assert kernel(OP_SC, 16, 8, F32, 0, OUT_INFO, DQ_FAST).startswith("sc_decode_kernel")
    assert kernel(OP_SCL, 32, 16, F32, 8, OUT_INFO, 0).startswith("scl_decode_kernel")
    # too few information bits to fill the list before the last leaves
    assert kernel(OP_SCL, 64, 2, F32, 8, OUT_INFO, 0).startswith("scl_decode_kernel")
    assert kernel(OP_SCL, 64, 4, F64, 32, OUT_INFO, 0).startswith("scl_decode_kernel")
    # list of one path
    assert kernel(OP_SCL, 1024, 512, F32, 1, OUT_INFO, 0).startswith("scl_decode_kernel")
    # above N = 1024 every list of two paths and more is on the two-lanes-per-path kernel, with or without the list as output
    assert kernel(OP_SCL, 4096, 2048, F32, 4, OUT_INFO, 0).startswith("scl_wide_kernel")
    assert kernel(OP_SCL, 2048, 1024, F32, 2, OUT_INFO, 0).startswith("scl_wide_kernel")
    assert kernel(OP_SCL, 2048, 1024, F32, 8, OUT_INFO | OUT_PM, 0).startswith("scl_wide_kernel")
    # the Monte-Carlo kernel exists from N = 128 on
    assert kernel(OP_MC_SC, 64, 32, F32, 0, 0, DQ_FAST | DQ_SPECIALISED).startswith("gen_frames_kernel")


HOLES = [
    ("SC fp64 beyond 32768", OP_SC, 65536, 30000, F64, 0, OUT_INFO, 0),
    ("SC beyond 65536", OP_SC, 1 << 17, 1000, F32, 0, OUT_INFO, DQ_FAST),
    ("list beyond 4096", OP_SCL, 8192, 4096, F32, 8, OUT_INFO, 0),
    ("list size 3", OP_SCL, 1024, 512, F32, 3, OUT_INFO, 0),
    ("list size 64", OP_SCL, 1024, 512, F32, 64, OUT_INFO, 0),
    ("flip search beyond 4096", OP_PBF, 8192, 4096, F32, 1, OUT_INFO, DQ_CRC),
    ("unknown arithmetic", OP_SC, 1024, 512, np.float16, 0, OUT_INFO, 0),
]


@pytest.mark.parametrize("what,op,N,K,dtype,L,outputs,flags", HOLES, ids=[h[0] for h in HOLES])
def test_holes_are_refused_with_a_message(what, op, N, K, dtype, L, outputs, flags):
    code = 7 if dtype is np.float16 else _native._dt(dtype)
    rc = _native.lib().polar_dispatch_select(op, N, K, code, L, outputs, flags, None, 0)
    assert rc < 0, what
    assert b"no kernel serves" in _native.lib().polar_last_error()


def test_not_a_power_of_two():
    assert _native.lib().polar_dispatch_select(OP_SC, 1000, 500, 0, 0, OUT_INFO, 0, None, 0) < 0
    assert b"power of two" in _native.lib().polar_last_error()
