"""Bit flipping: scDecoder(sequenceU), generateCriticalSet / modifyCriticalSet / constructTree and
progressiveBitFlipping against vectors captured from the reference (tests/golden/pbf.npz, tools/gen_golden.py
section `pbf`).  The stored frames record how many SC runs the reference made and the flips of its last run."""
import numpy as np
import pytest

import _native
import functions as pcf
from frames import make_frame, n0_ebn0, pack
from oracle import oracle
from test_cascl import ca_frame

CRIT = [("dega", 128, 72), ("dega", 256, 136), ("dega", 1024, 536), ("rmpolar", 512, 256), ("dega", 64, 20)]
PBF = [(128, 72, 0x107, 8, 1, 1.5), (128, 72, 0x107, 8, 2, 1.5), (256, 136, 0x107, 8, 1, 1.5), (128, 72, 0x107, 8, 3, 2.0),
       (128, 72, 0x107, 8, 3, 1.0), (1024, 536, 0x1800063, 24, 1, 1.75)]


def build_mask(cons, N, K):
    return pcf.degaBuild(N, K, 2.0) if cons == "dega" else pcf.rmPolarBuild(N, K, 2.0)


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("cons,N,K", CRIT)
def test_critical_sets_match_reference(golden, cons, N, K):
    g = golden.pbf
    mask = build_mask(cons, N, K)
    want = g[f"crit_{cons}_{N}_{K}"]
    assert np.array_equal(pcf.generateCriticalSet(1 - mask), want)
    assert np.array_equal(oracle.critical_set(1 - mask), want)
    assert np.array_equal(_native.critical_set(mask), want)          # the table builder of the GPU library (host only)
    for key in g.files:
        if key.startswith(f"crit_{cons}_{N}_{K}_after_"):
            i = int(key.rsplit("_", 1)[1])
            assert np.array_equal(pcf.modifyCriticalSet(mask, i), g[key])
            assert np.array_equal(oracle.modify_critical_set(mask, i), g[key])
            assert np.array_equal(_native.critical_set(mask, i), g[key])


def test_construct_tree_matches_reference(golden):
    g = golden.pbf
    tree = pcf.constructTree(pcf.degaBuild(64, 20, 2.0), 2)
    assert np.array_equal(np.array([nd.sequenceU for nd in tree[0]]), g["tree_64_20_level0"])
    assert np.array_equal(np.array([nd.sequenceU for nd in tree[1]]), g["tree_64_20_level1"])


def test_oracle_sc_with_flips_matches_reference(golden):
    g = golden.pbf
    mask = pcf.degaBuild(128, 64, 2.0)
    for f, seed in enumerate(g["flip_128_64_seeds"]):
        msg, x, llr = make_frame(mask, 1.0, int(seed), oracle.polar_encode)
        seq = [int(v) for v in g["flip_128_64_seq"][f] if v >= 0]
        uh, dl = oracle.sc_decode(llr, mask, want_llrs=True, flips=seq)
        assert np.array_equal(pack(uh), g["flip_128_64_uhat"][f])
        assert np.array_equal(dl, g["flip_128_64_decllr"][f])


@pytest.mark.parametrize("N,K,poly,w,T,snr", PBF)
def test_oracle_pbf_matches_reference_vectors(golden, N, K, poly, w, T, snr):
    g = golden.pbf
    tag = f"{N}_{K}_w{w}_T{T}_{snr}"
    mask, mu = pcf.degaBuild(N, K, 2.0), pcf.GA(N, K, 2.0)
    for f, seed in enumerate(g[f"{tag}_seeds"]):
        msg, word, x, llr = ca_frame(mask, w, poly, snr, int(seed))
        wd, runs, last = oracle.pbf_decode(llr, mask, poly, T, mu)
        assert runs == g[f"{tag}_runs"][f]
        assert np.array_equal(pack(wd[:K - w]), g[f"{tag}_dec"][f])
        assert int(np.sum(wd[:K - w] != msg)) == int(g[f"{tag}_biterr"][f])
        assert bool(pcf.crcCheck(wd, poly)) == bool(g[f"{tag}_crc_ok"][f])
        if runs > 1:
            assert np.array_equal(last, g[f"{tag}_last_seq"][f][:T])
    assert (g[f"{tag}_runs"] > 1).any()


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_sc_decoder_with_sequenceU(golden):
    from polarCode import PolarCode

    g = golden.pbf
    pc = PolarCode(128, 64, "dega", 2.0)
    fr = [make_frame(pc.polarMask, 1.0, int(s), oracle.polar_encode) for s in g["flip_128_64_seeds"]]
    llr = np.stack([f[2] for f in fr])
    # reference-style, one frame and one python list
    for f in range(4):
        seq = [int(v) for v in g["flip_128_64_seq"][f] if v >= 0]
        dec = pc.scDecoder(llr[f], seq)
        assert np.array_equal(pack(pc.uHat), g["flip_128_64_uhat"][f])
        assert np.array_equal(pc.llrs, g["flip_128_64_decllr"][f])
        assert np.array_equal(dec, pc.uHat[pc.polarMask == 1])
    # batched, one row of flips per frame
    dec = pc.scDecoder(llr, g["flip_128_64_seq"])
    assert np.array_equal(pack(pc.uHat), g["flip_128_64_uhat"])
    assert np.array_equal(pc.llrs, g["flip_128_64_decllr"])
    # fp32 against the fp32 oracle
    code = pc.native
    info, uh, dl = code.sc_decode_flip(llr.astype(np.float32), g["flip_128_64_seq"])
    for f in range(len(fr)):
        seq = [int(v) for v in g["flip_128_64_seq"][f] if v >= 0]
        ruh, rdl = oracle.sc_decode(llr[f].astype(np.float32), pc.polarMask, want_llrs=True, flips=seq)
        assert np.array_equal(uh[f], ruh) and np.array_equal(dl[f], rdl)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,w,T,snr", PBF)
def test_gpu_pbf_fp64_matches_reference_vectors(golden, N, K, poly, w, T, snr):
    from polarCode import PolarCode

    g = golden.pbf
    tag = f"{N}_{K}_w{w}_T{T}_{snr}"
    pc = PolarCode(N, K, "dega", 2.0)
    pc.crcPoly, pc.crcWidth, pc.maxLevel = poly, w, T
    fr = [ca_frame(pc.polarMask, w, poly, snr, int(s)) for s in g[f"{tag}_seeds"]]
    msgs = np.stack([f[0] for f in fr])
    llr = np.stack([f[3] for f in fr])
    dec = pc.progressiveBitFlipping(llr)
    assert np.array_equal(pc.scRuns, g[f"{tag}_runs"])
    assert np.array_equal(pack(dec), g[f"{tag}_dec"])
    assert np.array_equal(np.sum(dec != msgs, axis=1), g[f"{tag}_biterr"])
    assert np.array_equal(pc.crcPass.astype(int), g[f"{tag}_crc_ok"])
    flipped = pc.scRuns > 1
    assert np.array_equal(pc.lastSequenceU[flipped], g[f"{tag}_last_seq"][flipped])
    one = pc.progressiveBitFlipping(llr[1])
    assert one.shape == (K - w,) and np.array_equal(one, dec[1]) and pc.scRuns == g[f"{tag}_runs"][1]


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,T,nfr,dtype", [(128, 72, 0x107, 1, 3000, 1), (128, 72, 0x107, 2, 600, 1), (256, 140, 0x1621, 1, 1500, 0),
                                                   (1024, 536, 0x1800063, 1, 400, 0), (64, 30, 0x107, 3, 300, 0)])
def test_gpu_pbf_equals_oracle_on_random_frames(N, K, poly, T, nfr, dtype):
    """Philox frames (CRC appended on the device), fp64 and fp32: runs, flips and bits equal the oracle's, and the
    fused Monte-Carlo loop counts the same errors."""
    import torch

    mask, mu = pcf.degaBuild(N, K, 2.0), pcf.GA(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(2.0, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    code.set_reliability(mu)
    npdt = np.float64 if dtype else np.float32
    dev = torch.device("cuda:0")
    info_t = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
    llr_t = torch.empty((nfr, N), dtype=torch.float64 if dtype else torch.float32, device=dev)
    code.gen_frames_dev(info_t.data_ptr(), llr_t.data_ptr(), npdt, n0, 77, 10, nfr, crc=True)
    code.ctx.sync()
    llr = llr_t.cpu().numpy()
    sent = np.unpackbits(info_t.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :K]
    info, runs, ok, last = code.pbf_decode(llr, T)
    fe = be = 0
    for f in range(nfr):
        wd, r, ls = oracle.pbf_decode(llr[f], mask, poly, T, mu)
        assert r == runs[f], (f, r, runs[f])
        assert np.array_equal(wd, info[f]), f
        assert bool(ok[f]) == bool(pcf.crcCheck(wd, poly))
        if r > 1:
            assert np.array_equal(ls, last[f][:T])
        e = int(np.sum(wd[:K - w] != sent[f, :K - w]))
        fe += e > 0
        be += e
    assert (runs > 1).any()
    got = code.mc_run(_native.DEC_PBF, T, npdt, n0, 77, 10, nfr, crc=True)
    assert got.tolist() == [nfr, fe, be]
    plain = code.mc_run(_native.DEC_SC, 1, npdt, n0, 77, 10, nfr, fast=True)
    assert got[1] <= plain[1]
    if not dtype and N >= 32:
        code.specialize()  # the first SC run of every frame then comes from the decoder compiled for this code
        assert code.mc_run(_native.DEC_PBF, T, npdt, n0, 77, 10, nfr, crc=True).tolist() == got.tolist()
        assert code.mc_run(_native.DEC_SC, 1, npdt, n0, 77, 10, nfr, fast=True).tolist() == plain.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,nfr,snr,dtype", [(1024, 536, 0x1800063, 6000, 2.0, 0), (128, 72, 0x107, (1 << 18) + 777, 3.0, 0),
                                                     (256, 140, 0x1621, 4000, 1.5, 1)])
def test_gpu_pbf_two_flips_in_pieces_equals_whole_frame_search(N, K, poly, nfr, snr, dtype, monkeypatch):
    """Two flips: the search worked on in (frame, first flip) pieces by many wavefronts gives the runs, flips, CRC
    flags and bits of the search one wavefront makes per frame (which the oracle tests pin), also when the batch
    takes more than one pass of the workspace."""
    import torch

    mask, mu = pcf.degaBuild(N, K, 2.0), pcf.GA(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(snr, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    code.set_reliability(mu)
    npdt = np.float64 if dtype else np.float32
    dev = torch.device("cuda:0")
    info_t = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
    llr_t = torch.empty((nfr, N), dtype=torch.float64 if dtype else torch.float32, device=dev)
    code.gen_frames_dev(info_t.data_ptr(), llr_t.data_ptr(), npdt, n0, 5, 0, nfr, crc=True)

    def search():
        hat = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
        runs = torch.zeros(nfr, dtype=torch.int32, device=dev)
        ok = torch.zeros(nfr, dtype=torch.uint8, device=dev)
        last = torch.zeros((nfr, 3), dtype=torch.int32, device=dev)
        code.pbf_decode_dev(llr_t.data_ptr(), npdt, 2, hat.data_ptr(), nfr, runs_ptr=runs.data_ptr(), ok_ptr=ok.data_ptr(),
                            last_ptr=last.data_ptr())
        code.ctx.sync()
        return hat.cpu().numpy(), runs.cpu().numpy(), ok.cpu().numpy(), last.cpu().numpy(), code.last_launch()["kernel"]

    pieces = search()
    monkeypatch.setenv("POLAR_PBF_SPLIT", "0")
    whole = search()
    assert pieces[4].endswith("+pairs") and not whole[4].endswith("+pairs")
    for a, b in zip(pieces[:4], whole[:4]):
        assert np.array_equal(a, b)
    assert (whole[3][:, 1] >= 0).any() and (whole[2] == 0).any()   # some frames needed two flips, some found none


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,nfr,snr,dtype,family", [(64, 30, 0x107, 3000, 1.0, 0, "reg"), (128, 72, 0x107, 1500, 1.5, 1, "reg"),
                                                            (256, 140, 0x1621, 300, 1.5, 0, "reg"), (32, 20, 0x107, 2000, 0.0, 1, "reg"),
                                                            (64, 30, 0x107, 3000, 1.0, 1, "lds"), (128, 72, 0x107, 800, 0.5, 0, "lds")])
def test_gpu_pbf_three_flips_in_pieces_equals_whole_frame_search(N, K, poly, nfr, snr, dtype, family, monkeypatch):
    """Three flips: the search worked on in (frame, first flip) pieces by many wavefronts (SC runs in registers, or LDS-staged)
    gives the runs, flips, CRC flags and bits of the depth-first search one wavefront makes per frame (which the oracle tests and
    the reference's vectors pin): frames that pass with one, two and three flips and frames that never pass."""
    import torch

    mask, mu = pcf.degaBuild(N, K, 2.0), pcf.GA(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(snr, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    code.set_reliability(mu)
    npdt = np.float64 if dtype else np.float32
    dev = torch.device("cuda:0")
    info_t = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
    llr_t = torch.empty((nfr, N), dtype=torch.float64 if dtype else torch.float32, device=dev)
    code.gen_frames_dev(info_t.data_ptr(), llr_t.data_ptr(), npdt, n0, 21, 0, nfr, crc=True)

    def search():
        hat = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
        runs = torch.zeros(nfr, dtype=torch.int32, device=dev)
        ok = torch.zeros(nfr, dtype=torch.uint8, device=dev)
        last = torch.zeros((nfr, 3), dtype=torch.int32, device=dev)
        code.pbf_decode_dev(llr_t.data_ptr(), npdt, 3, hat.data_ptr(), nfr, runs_ptr=runs.data_ptr(), ok_ptr=ok.data_ptr(),
                            last_ptr=last.data_ptr())
        code.ctx.sync()
        return hat.cpu().numpy(), runs.cpu().numpy(), ok.cpu().numpy(), last.cpu().numpy(), code.last_launch()["kernel"]

    if family == "lds":  # the LDS-staged search (what codes longer than 1024 get) has its own kernels for the pieces
        monkeypatch.setenv("POLAR_PBF_KERNEL", "lds")
    pieces = search()
    monkeypatch.setenv("POLAR_PBF_SPLIT", "0")
    whole = search()
    assert pieces[4].startswith("pbf_reg_kernel" if family == "reg" else "pbf_decode_kernel") and pieces[4].endswith("+pairs+triples")
    assert whole[4].startswith("pbf_decode_kernel") and not whole[4].endswith("triples")
    for name, a, b in zip(("bits", "runs", "crc", "flips"), pieces[:4], whole[:4]):
        assert np.array_equal(a, b), (name, np.flatnonzero((a != b).reshape(nfr, -1).any(axis=1))[:5])
    ok, last = whole[2], whole[3]
    assert (ok[last[:, 2] >= 0] == 1).any()      # some frames needed all three flips
    assert ((last[:, 1] >= 0) & (last[:, 2] < 0)).any() and (ok == 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,T,nfr,snr,dtype", [(32, 20, 0x107, 1, 3000, 1.0, 0), (32, 20, 0x107, 2, 2000, 1.0, 1), (64, 30, 0x107, 2, 2000, 1.0, 0),
                                                       (512, 280, 0x1800063, 1, 3000, 1.5, 1), (512, 280, 0x1800063, 2, 3000, 2.0, 0),
                                                       (1024, 536, 0x1800063, 1, 4000, 1.5, 1)])
def test_gpu_pbf_register_search_equals_lds_search(N, K, poly, T, nfr, snr, dtype, monkeypatch):
    """The search with its SC runs in registers (four candidates per wavefront) against the LDS-staged search with one
    run per wavefront: runs, flips, CRC flags and bits, every length the register kernel serves, fp32 and fp64."""
    import torch

    mask, mu = pcf.degaBuild(N, K, 2.0), pcf.GA(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(snr, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    code.set_reliability(mu)
    npdt = np.float64 if dtype else np.float32
    dev = torch.device("cuda:0")
    info_t = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
    llr_t = torch.empty((nfr, N), dtype=torch.float64 if dtype else torch.float32, device=dev)
    code.gen_frames_dev(info_t.data_ptr(), llr_t.data_ptr(), npdt, n0, 9, 0, nfr, crc=True)

    def search():
        hat = torch.zeros((nfr, code.Kw), dtype=torch.int32, device=dev)
        runs = torch.zeros(nfr, dtype=torch.int32, device=dev)
        ok = torch.zeros(nfr, dtype=torch.uint8, device=dev)
        last = torch.zeros((nfr, 3), dtype=torch.int32, device=dev)
        code.pbf_decode_dev(llr_t.data_ptr(), npdt, T, hat.data_ptr(), nfr, runs_ptr=runs.data_ptr(), ok_ptr=ok.data_ptr(),
                            last_ptr=last.data_ptr())
        code.ctx.sync()
        return hat.cpu().numpy(), runs.cpu().numpy(), ok.cpu().numpy(), last.cpu().numpy(), code.last_launch()["kernel"]

    reg = search()
    monkeypatch.setenv("POLAR_PBF_KERNEL", "lds")
    lds = search()
    assert reg[4].startswith("pbf_reg_kernel") and lds[4].startswith("pbf_decode_kernel")
    for a, b in zip(reg[:4], lds[:4]):
        assert np.array_equal(a, b)
    assert (lds[1] > 1).any() and (lds[2] == 0).any()


@pytest.mark.gpu
def test_gpu_pbf_argument_errors():
    code = _native.Code(_native.default_ctx(), pcf.degaBuild(64, 32, 2.0))
    llr = np.ones((2, 64), dtype=np.float32)
    with pytest.raises(_native.PolarError):
        code.pbf_decode(llr, 1)                    # no CRC
    code.set_crc(0x107)
    with pytest.raises(_native.PolarError):
        code.pbf_decode(llr, 1)                    # no reliabilities
    info, runs, ok, last = code.pbf_decode(llr, 0)  # level 0: SC + CRC test only
    assert runs.tolist() == [1, 1] and ok.tolist() == [1, 1] and not info.any()
    code.set_reliability(pcf.GA(64, 32, 2.0))
    with pytest.raises(_native.PolarError):
        code.pbf_decode(llr, 4)
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_PBF, 1, np.float32, 1.0, 1, 0, 10)   # needs the CRC flag
