"""CRC-aided list decoding (sclDecoder(isCRC=True)) against vectors captured from the reference.

tests/golden/cascl.npz holds (tools/gen_golden.py, section `cascl`): crcEncode / standardForm / getGC outputs,
and for three codes the frames the reference decoded with isCRC=True (codeword of CRCPolarEncode2, decoded
message, bit errors with and without the CRC choice).  A frame is reproducible from its seed: the legacy
MT19937 stream yields the K - width message bits, then the N noise samples; Eb/N0 refers to the message rate.
"""
import numpy as np
import pytest

import functions as pcf
from frames import n0_ebn0, pack, unpack
from oracle import oracle

POLYS = [(0x107, "crc8"), (0x1800063, "crc24"), (0x1621, "crc12")]
CFGS = [(128, 64, 0x107, 8, 4, 1.0), (256, 128, 0x107, 8, 8, 1.0), (1024, 536, 0x1800063, 24, 8, 1.0)]


def ca_frame(mask, width, poly, snr_db, seed):
    """(message, info word with CRC, codeword, fp64 LLRs) of one stored frame."""
    mask = np.asarray(mask)
    N, K = mask.size, int(mask.sum())
    rs = np.random.RandomState(seed)
    msg = rs.randint(0, 2, size=K - width, dtype=int)
    z = rs.standard_normal(N)
    word = oracle.crc_encode(msg, poly)
    x = oracle.polar_encode(word, mask).astype(np.int64)
    n0 = n0_ebn0(snr_db, (K - width) / N)
    llr = (4 / n0) * ((1 - 2 * x) + np.sqrt(n0 / 2) * z)
    return msg.astype(np.uint8), word, x.astype(np.uint8), llr


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("poly,name", POLYS)
def test_crc_functions_match_reference(golden, poly, name):
    g = golden.cascl
    msg = unpack(g[f"{name}_msg"], 40)
    w = poly.bit_length() - 1
    enc = pcf.crcEncode(msg, poly)
    assert enc.shape == (6, 40 + w)
    assert np.array_equal(pack(enc), g[f"{name}_enc"])
    assert np.array_equal(pcf.crcEncode(msg[0], poly), enc[0])
    assert np.array_equal(pcf.standardForm(poly)[0], g[f"{name}_std"])
    assert pcf.crcCheck(enc, poly).all() and pcf.crcCheck(enc[0], poly) is True
    bad = enc.copy()
    bad[:, 3] ^= 1
    assert not pcf.crcCheck(bad, poly).any()
    for r in range(6):  # the oracle's restatement, bit-array register and all
        assert np.array_equal(oracle.crc_encode(msg[r], poly), enc[r])


def test_standard_form_strings():
    assert pcf.standardForm(0x107)[1] == "0x07"
    assert pcf.standardForm(0x1800063)[1] == "0x800063"


def test_getGC_matches_reference(golden):
    g = [1, 0, 0, 0, 0, 0, 1, 1, 1]
    GC = pcf.getGC(20, g)
    assert np.array_equal(GC, golden.cascl["getGC_20_crc8"])
    rs = np.random.RandomState(3)
    m = rs.randint(0, 2, size=(5, 20))
    assert np.array_equal(m.dot(GC) % 2, pcf.crcEncode(m, 0x107))  # the two CRC encoders of the reference agree
    assert np.array_equal(pcf.polynomialMod([1, 0, 0, 0, 0, 0, 0, 0, 0, 0], g), [0, 0, 0, 0, 1, 1, 1, 0])


@pytest.mark.parametrize("N,K,poly,w,L,snr", CFGS)
def test_oracle_ca_scl_matches_reference_vectors(golden, N, K, poly, w, L, snr):
    g = golden.cascl
    tag = f"{N}_{K}_w{w}_L{L}"
    mask = pcf.degaBuild(N, K, 2.0)
    changed = 0
    for f, seed in enumerate(g[f"{tag}_seeds"]):
        msg, word, x, llr = ca_frame(mask, w, poly, snr, int(seed))
        assert np.array_equal(pack(x), g[f"{tag}_x"][f])
        dec = oracle.ca_scl_decode(llr, mask, L, poly)
        assert np.array_equal(pack(dec), g[f"{tag}_dec"][f]), (tag, f)
        assert int(np.sum(dec != msg)) == int(g[f"{tag}_biterr"][f])
        changed += int(g[f"{tag}_biterr"][f] != g[f"{tag}_biterr_first_path"][f])
    assert changed > 0  # the stored frames do exercise the CRC choice


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,w,L,snr", CFGS)
def test_gpu_ca_scl_fp64_matches_reference_vectors(golden, N, K, poly, w, L, snr):
    from polarCode import PolarCode

    g = golden.cascl
    tag = f"{N}_{K}_w{w}_L{L}"
    pc = PolarCode(N, K, "dega", 2.0, listSize=L)
    pc.crcPoly, pc.crcWidth = poly, w
    seeds = g[f"{tag}_seeds"]
    fr = [ca_frame(pc.polarMask, w, poly, snr, int(s)) for s in seeds]
    msgs = np.stack([f[0] for f in fr])
    llr = np.stack([f[3] for f in fr])
    x = pc.CRCPolarEncode2(msgs)
    assert np.array_equal(pack(x), g[f"{tag}_x"])
    assert np.array_equal(pc.CRCPolarEncode2(msgs[0]), x[0])
    dec = pc.sclDecoder(llr, isCRC=True)
    assert dec.shape == (len(seeds), K - w)
    assert np.array_equal(pack(dec), g[f"{tag}_dec"])
    assert np.array_equal(np.sum(dec != msgs, axis=1), g[f"{tag}_biterr"])
    # the same choice made on the host from the list the GPU returned
    words = np.stack([[p.uHat[pc.polarMask == 1] for p in pl] for pl in pc.pathList])
    ok = pcf.crcCheck(words.reshape(-1, K), poly).reshape(words.shape[:2])
    pick = np.where(ok.any(axis=1), ok.argmax(axis=1), 0)
    assert np.array_equal(dec, words[np.arange(len(seeds)), pick][:, :K - w])
    assert (pick > 0).any()
    one = pc.sclDecoder(llr[0], isCRC=True)
    assert one.shape == (K - w,) and np.array_equal(one, dec[0])
    # without isCRC the best path comes back, CRC bits included
    plain = pc.sclDecoder(llr)
    assert np.array_equal(np.sum(plain[:, :K - w] != msgs, axis=1), g[f"{tag}_biterr_first_path"])


@pytest.mark.gpu
def test_gpu_crc_generator_matrix_encoder():
    from polarCode import PolarCode

    pc = PolarCode(128, 64, "dega", 2.0, listSize=4)
    pc.crcPoly, pc.crcWidth, pc.crcPolyArrayForm = 0x107, 8, np.array([1, 0, 0, 0, 0, 0, 1, 1, 1])
    m = np.random.RandomState(5).randint(0, 2, size=(7, 56))
    assert np.array_equal(pc.CRCPolarEncode(m), pc.CRCPolarEncode2(m))


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,L,nfr", [(128, 64, 0x107, 4, 3000), (256, 128, 0x1621, 8, 1500), (64, 30, 0x107, 16, 2000),
                                             (1024, 536, 0x1800063, 8, 300), (2048, 1040, 0x1800063, 2, 200)])
def test_gpu_mc_run_crc_fp64_equals_oracle(N, K, poly, L, nfr):
    """Whole CRC-aided loop on the GPU (frames with CRC, list decoding, CRC choice, message-bit counting)
    against the oracle's loop: identical counters."""
    import _native

    mask = pcf.degaBuild(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(1.5, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    assert code.crc_width == w
    ref = oracle.mc_run_crc(mask, L, 1, n0, poly, 99, 500, nfr)
    got = code.mc_run(_native.DEC_SCL, L, np.float64, n0, 99, 500, nfr, crc=True)
    assert got.tolist() == ref.tolist()
    assert 0 < ref[1] < nfr
    nocrc = code.mc_run(_native.DEC_SCL, L, np.float64, n0, 99, 500, nfr)
    assert nocrc[0] == nfr


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly", [(256, 128, 0x107), (1024, 536, 0x1800063), (64, 32, 0x1621), (512, 300, 0x107)])
def test_gpu_ca_register_kernel_equals_lds_kernel(monkeypatch, N, K, poly):
    """fp32, L = 8: the register-resident kernel and the LDS-staged kernel make the same CRC-aided choice."""
    import _native

    mask = pcf.degaBuild(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(1.0, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    B = 6000
    rs = np.random.RandomState(8)
    word = pcf.crcEncode(rs.randint(0, 2, size=(B, K - w)), poly)
    x = oracle.polar_encode(word, mask)
    llr = code.channel_llr(x, n0, 5, 0, np.float32)
    a, _, _ = code.scl_decode(llr, 8, crc=True)
    assert code.last_launch()["kernel"].startswith("scl_reg_kernel"), code.last_launch()
    plain, _, _ = code.scl_decode(llr, 8)
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, uh = code.scl_decode(llr, 8, want_list=True, crc=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), int(np.any(a != b, axis=1).sum())
    assert np.any(a != plain)  # the choice differs from the best path on some frames
    # and both agree with the choice made on the host from the LDS kernel's list
    words = uh[:, :, mask == 1]
    ok = pcf.crcCheck(words.reshape(-1, K), poly).reshape(B, 8)
    pick = np.where(ok.any(axis=1), ok.argmax(axis=1), 0)
    assert np.array_equal(b, words[np.arange(B), pick])
    # fewer frame errors with the CRC than without
    fe_crc = int(np.any(a[:, :K - w] != word[:, :K - w], axis=1).sum())
    fe_plain = int(np.any(plain[:, :K - w] != word[:, :K - w], axis=1).sum())
    assert fe_crc < fe_plain


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,poly,L", [(256, 128, 0x107, 32), (1024, 536, 0x1800063, 16), (64, 32, 0x1621, 32), (512, 300, 0x107, 16),
                                          (512, 280, 0x1800063, 32), (2048, 1048, 0x1800063, 8), (2048, 1048, 0x1800063, 16)])
def test_gpu_ca_wide_kernel_equals_lds_kernel(monkeypatch, N, K, poly, L):
    """fp32, L = 16 / 32 (and 8 at N = 2048): the two-lanes-per-path register kernel and the LDS-staged kernel make the same
    CRC-aided choice, and it is the choice the host makes from the LDS kernel's list."""
    import _native

    mask = pcf.degaBuild(N, K, 2.0)
    w = poly.bit_length() - 1
    n0 = n0_ebn0(0.5, (K - w) / N)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    B = 2000 if N <= 1024 else 500
    rs = np.random.RandomState(8)
    word = pcf.crcEncode(rs.randint(0, 2, size=(B, K - w)), poly)
    x = oracle.polar_encode(word, mask)
    llr = code.channel_llr(x, n0, 5, 0, np.float32)
    a, _, _ = code.scl_decode(llr, L, crc=True)
    assert code.last_launch()["kernel"].startswith("scl_wide_kernel"), code.last_launch()
    plain, _, _ = code.scl_decode(llr, L)
    monkeypatch.setenv("POLAR_SCL_KERNEL", "lds")
    b, _, uh = code.scl_decode(llr, L, want_list=True, crc=True)
    assert code.last_launch()["kernel"].startswith("scl_decode_kernel")
    assert np.array_equal(a, b), int(np.any(a != b, axis=1).sum())
    assert np.any(a != plain)
    words = uh[:, :, mask == 1]
    ok = pcf.crcCheck(words.reshape(-1, K), poly).reshape(B, L)
    pick = np.where(ok.any(axis=1), ok.argmax(axis=1), 0)
    assert np.array_equal(b, words[np.arange(B), pick])
    assert (pick > 0).any() and (~ok.any(axis=1)).any()   # deeper list entries get chosen, and some frames have no passing entry


@pytest.mark.gpu
def test_gpu_gen_frames_crc_and_counting():
    """polar_gen_frames with POLAR_CRC: message bits are those of the CRC-less frame, the tail is their CRC;
    polar_count_errors with POLAR_CRC ignores the CRC bits."""
    import torch

    import _native

    N, K, poly = 512, 200, 0x1800063
    w = 24
    mask = pcf.degaBuild(N, K, 2.0)
    code = _native.Code(_native.default_ctx(), mask)
    code.set_crc(poly)
    B = 777
    dev = torch.device("cuda:0")
    info_a = torch.zeros((B, code.Kw), dtype=torch.int32, device=dev)
    info_b = torch.zeros_like(info_a)
    llr_a = torch.empty((B, N), dtype=torch.float64, device=dev)
    llr_b = torch.empty_like(llr_a)
    code.gen_frames_dev(info_a.data_ptr(), llr_a.data_ptr(), np.float64, 0.8, 4, 100, B)
    code.gen_frames_dev(info_b.data_ptr(), llr_b.data_ptr(), np.float64, 0.8, 4, 100, B, crc=True)
    code.ctx.sync()
    ia = np.unpackbits(info_a.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :K]
    ib = np.unpackbits(info_b.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    assert not ib[:, K:].any()
    ib = ib[:, :K]
    assert np.array_equal(ia[:, :K - w], ib[:, :K - w])
    assert np.array_equal(ib, pcf.crcEncode(ib[:, :K - w], poly))
    for f in (0, 5, B - 1):
        msg = np.empty(K, dtype=np.uint8)
        oracle.lib().orc_gen_info(4, 100 + f, K, msg)
        assert np.array_equal(msg[:K - w], ib[f, :K - w])
        x = oracle.polar_encode(ib[f], mask).astype(np.float64)
        z = oracle.gen_noise(4, 100 + f, N)
        ref = (4 / 0.8) * ((1 - 2 * x) + np.sqrt(0.8 / 2) * z)
        np.testing.assert_allclose(llr_b[f].cpu().numpy(), ref, rtol=0, atol=1e-12)
    # counting: flip one CRC bit and one message bit in some frames
    hat = ib.copy()
    hat[::3, K - 1] ^= 1          # CRC bit: must not count with the flag
    hat[::5, 7] ^= 1              # message bit
    hat[::7, K - w - 1] ^= 1      # last message bit
    hw = np.zeros((B, code.Kw * 32), dtype=np.uint8)
    hw[:, :K] = hat
    hat_t = torch.from_numpy(np.packbits(hw, axis=1, bitorder="little").view(np.int32)).to(dev)
    cnt = torch.zeros(3, dtype=torch.int64, device=dev)
    code.count_errors_dev(info_b.data_ptr(), hat_t.data_ptr(), B, cnt.data_ptr(), crc=True)
    code.ctx.sync()
    diff = hat[:, :K - w] != ib[:, :K - w]
    assert cnt.cpu().tolist() == [B, int(diff.any(axis=1).sum()), int(diff.sum())]
    cnt.zero_()
    code.count_errors_dev(info_b.data_ptr(), hat_t.data_ptr(), B, cnt.data_ptr())
    code.ctx.sync()
    diff = hat != ib
    assert cnt.cpu().tolist() == [B, int(diff.any(axis=1).sum()), int(diff.sum())]


@pytest.mark.gpu
def test_gpu_crc_argument_errors():
    import _native

    code = _native.Code(_native.default_ctx(), pcf.degaBuild(64, 8, 2.0))
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SCL, 4, np.float32, 1.0, 1, 0, 10, crc=True)   # no CRC attached
    with pytest.raises(_native.PolarError):
        code.set_crc(0x1800063)                                                 # wider than K
    with pytest.raises(_native.PolarError):
        code.set_crc(1 << 40)                                                   # wider than 32
    code.set_crc(0x107)
    with pytest.raises(_native.PolarError):
        code.mc_run(_native.DEC_SC, 1, np.float32, 1.0, 1, 0, 10, crc=True)    # CRC goes with the list decoder
    code.set_crc(0)
    assert code.crc_width == 0
