"""PolarCode: construction on the host, encode / SC / SCL decode on the GPU.

Keeps the names, argument meaning and return types of the reference's cn/polarCode.py for the
encode -> decode path (constructor attributes, rateProfile, mul_matrix, polarEncode, extract, decode,
scDecoder, sclDecoder).  Every method also accepts a leading batch axis [B, ...].  The arithmetic runs
in libpolar_hip.so through ctypes; there is no CPU fallback.
"""
import math

import numpy as np

import _native
import functions as pcfun


class Path:
    """Read-only view of one surviving SCL path (cn/polarCode.py:14-28 keeps full decoder state per
    path; after decoding only uHat and pathMetric are meaningful, and those are what is exposed)."""

    def __init__(self, uHat, pathMetric):
        self.uHat = uHat
        self.pathMetric = pathMetric


class PolarCode:
    def __init__(self, N, K, construct, dSNR, stackSize=0, listSize=0, gen=[]):
        """Same signature as cn/polarCode.py:116.  `construct` in {'dega', 'rmpolar'} (case-insensitive);
        the reference's 'ws' and 'my' constructions belong to the PAC code family and are not provided."""
        self.codewordLength = N
        self.infoLen = K
        self.n = int(math.log2(N))
        self.designSNR = dSNR
        self.indicesRev = [pcfun.bitreversed(i, self.n) for i in range(N)]
        c = construct.lower()
        if c == "dega":
            self.polarMask = pcfun.degaBuild(N, K, dSNR)
        elif c == "rmpolar":
            self.polarMask = pcfun.rmPolarBuild(N, K, dSNR)
        elif c == "rm":
            self.polarMask = pcfun.rmBuild(N, K)
        else:
            raise NotImplementedError("construction %r is not provided (use 'dega', 'rmpolar' or 'rm')" % (construct,))
        self._init_common(stackSize, listSize, gen)

    @classmethod
    def from_mask(cls, mask, dSNR=0.0, listSize=0):
        """Code object around an explicit rate profile (1 = information bit)."""
        self = cls.__new__(cls)
        mask = np.asarray(mask).astype(int)
        self.codewordLength = int(mask.size)
        self.infoLen = int(mask.sum())
        self.n = int(math.log2(mask.size))
        self.designSNR = dSNR
        self.indicesRev = [pcfun.bitreversed(i, self.n) for i in range(mask.size)]
        self.polarMask = mask
        self._init_common(0, listSize, [])
        return self

    def _init_common(self, stackSize, listSize, gen):
        N = self.codewordLength
        self.frozenMask = (self.polarMask + 1) % 2
        self.A = np.array(range(N), dtype=int)[self.polarMask == 1]
        self.Fn = np.array([[1, 0], [1, 1]])
        self.gen = gen
        self.m = len(gen) - 1
        self.snrType = ""
        self.modu = ""
        self.D = stackSize
        self.L = listSize
        self.isCRC = False
        self.crcPoly = 0x0
        self.crcPolyArrayForm = []
        self.crcWidth = 0
        self.maxLevel = 1
        self.criticalSet = pcfun.generateCriticalSet(self.frozenMask)
        self.decoderType = "sc"
        self.dtype = np.float64  # arithmetic type of the decoders: float64 = parity with the reference
        # bitReversed = True reproduces the ordering of the reference's stand-alone script cn/main.py:233,255-273:
        # the encoder is fed u[indicesRev] and step i of the decoder decides uHat[bitreversed(i)] with
        # polarMask[bitreversed(i)] -- i.e. the natural-order code with the bit-reversed rate profile.
        self.bitReversed = False
        self._code = None
        self._code_rev = None

    # ---- lazily built pieces ------------------------------------------------------------------------
    @property
    def GN(self):
        """Dense generator matrix (cn/polarCode.py:131).  O(N^2): only built when asked for."""
        if "_GN" not in self.__dict__:
            self._GN = pcfun.GN(self.codewordLength, self.Fn)
        return self._GN

    @property
    def native(self):
        """Device-side code object; created on first use (raises if the GPU library is unavailable)."""
        if self._code is None:
            self._code = _native.Code(_native.default_ctx(), self.polarMask)
        return self._code

    @property
    def _rev(self):
        return np.asarray(self.indicesRev)

    @property
    def _active(self):
        """Device code object of the ordering in force (natural, or bit-reversed rate profile)."""
        if not self.bitReversed:
            return self.native
        if self._code_rev is None:
            self._code_rev = _native.Code(_native.default_ctx(), self.polarMask[self._rev])
        return self._code_rev

    # ---- encode -------------------------------------------------------------------------------------
    def rateProfile(self, info):
        """u = 0; u[polarMask == 1] = info (cn/polarCode.py:225-230)."""
        info = np.asarray(info)
        u = np.zeros(info.shape[:-1] + (self.codewordLength,), dtype=int)
        u[..., self.polarMask == 1] = info
        return u

    def mul_matrix(self, u):
        """x = u F^{(x)n} mod 2, natural order; like the reference (cn/polarCode.py:232-248) the result is
        written back into `u` when it is an integer ndarray, and returned."""
        arr = np.asarray(u)
        x = self.native.mul_matrix(arr.reshape(-1, self.codewordLength)).reshape(arr.shape)
        if isinstance(u, np.ndarray) and np.issubdtype(u.dtype, np.integer):
            u[...] = x
            return u
        return x.astype(int)

    def polarEncode(self, info):
        """mul_matrix(rateProfile(info)) in one GPU call (cn/polarCode.py:250-255)."""
        info = np.asarray(info)
        if self.bitReversed:  # cn/main.py:228-233: x = mul_matrix(u[indicesRev])
            v = self.rateProfile(info)[..., self._rev]
            return self.native.mul_matrix(v.reshape(-1, self.codewordLength)).reshape(v.shape).astype(int)
        x = self.native.encode(info.reshape(int(np.prod(info.shape[:-1], dtype=np.int64)), self.infoLen))
        return x.reshape(info.shape[:-1] + (self.codewordLength,)).astype(int)

    def CRCPolarEncode2(self, info):
        """Append the CRC of self.crcPoly to the message, then polar-encode (cn/polarCode.py:265-271)."""
        return self.polarEncode(pcfun.crcEncode(info, self.crcPoly))

    def CRCPolarEncode(self, info):
        """Systematic CRC through the generator matrix of self.crcPolyArrayForm (cn/polarCode.py:273-280)."""
        GC = pcfun.getGC(self.infoLen - self.crcWidth, self.crcPolyArrayForm)
        return self.polarEncode(np.dot(np.asarray(info), GC) % 2)

    def extract(self, uHat):
        """uHat[polarMask == 1] in index order (cn/polarCode.py:305-312)."""
        return np.asarray(uHat)[..., self.polarMask == 1].astype(int)

    # ---- decode -------------------------------------------------------------------------------------
    def decode(self, softMess, y=[], LMax=0):
        """String dispatch on self.decoderType like cn/polarCode.py:316-350; 'sc' and 'scl' are provided."""
        t = self.decoderType.lower()
        if t == "sc":
            return self.scDecoder(softMess)
        if t == "scl":
            return self.sclDecoder(softMess)
        raise NotImplementedError("decoder type %r is not provided (the reference blocks on input() here)" % (self.decoderType,))

    def _llr(self, softMess):
        a = np.asarray(softMess)
        if a.dtype != np.float32 or np.dtype(self.dtype) == np.float64:
            a = a.astype(self.dtype)
        return a

    def scDecoder(self, softMess, sequenceU=[]):
        """Successive cancellation (cn/polarCode.py:1525-1571).  Sets self.uHat, self.llrs (decision LLRs),
        self.ANV = N like the reference; returns the decoded information bits."""
        llr = self._llr(softMess)
        batch = llr.shape[:-1]
        if len(sequenceU):
            # the flip hook (cn/polarCode.py:1555-1558): decisions at these positions are inverted.  One list for
            # every frame, or one row per frame ([B, <=3], -1 = unused).
            if self.bitReversed:
                raise NotImplementedError("sequenceU together with the bit-reversed ordering is not provided")
            seq = np.asarray(sequenceU, dtype=np.int32)
            nfr = int(np.prod(batch, dtype=np.int64)) if batch else 1
            seq = np.broadcast_to(seq, (nfr, seq.shape[-1]))
            info, uh, dl = self.native.sc_decode_flip(llr.reshape(-1, self.codewordLength), seq)
        else:
            info, uh, dl = self._active.sc_decode(llr.reshape(-1, self.codewordLength), want_uhat=True, want_llrs=True)
        if self.bitReversed:  # decoder step i decided uHat[bitreversed(i)] (cn/main.py:256-273)
            uh = uh[:, self._rev]
            info = uh[:, self.polarMask == 1]
        self.uHat = uh.reshape(batch + (self.codewordLength,)).astype(int)
        self.llrs = dl.reshape(batch + (self.codewordLength,))  # in decoding-step order, like main.py's llrOb
        self.ANV = self.codewordLength
        self.mT = -6
        self.predictIdx = []
        return info.reshape(batch + (self.infoLen,)).astype(int)

    def progressiveBitFlipping(self, softMess):
        """Progressive bit flipping (cn/polarCode.py:1573-1628) with self.maxLevel simultaneous flips at most,
        self.crcPoly / self.crcWidth and the Gaussian-approximation reliabilities of (N, infoLen, designSNR).
        Returns the message bits of the last SC run made (the reference returns them whether or not the CRC
        holds); sets self.scRuns (SC runs per frame), self.crcPass and self.lastSequenceU."""
        if self.bitReversed:
            raise NotImplementedError("bit flipping together with the bit-reversed ordering is not provided")
        llr = self._llr(softMess)
        batch = llr.shape[:-1]
        self._attach_crc()
        if getattr(self.native, "_mu_for", None) != (self.infoLen, self.designSNR):
            self.native.set_reliability(pcfun.GA(self.codewordLength, self.infoLen, self.designSNR))
            self.native._mu_for = (self.infoLen, self.designSNR)
        info, runs, ok, last = self.native.pbf_decode(llr.reshape(-1, self.codewordLength), int(self.maxLevel))
        self.scRuns = runs.reshape(batch) if batch else int(runs[0])
        self.crcPass = ok.reshape(batch).astype(bool) if batch else bool(ok[0])
        self.lastSequenceU = last.reshape(batch + (3,))
        km = self.infoLen - self.crcWidth
        return info[:, :km].reshape(batch + (km,)).astype(int)

    def sclDecoderOracle(self, softMess, isCRC=False, isRCPP=False):
        """Genie-aided list decoding (cn/polarCode.py:1267-1317), the decoder the reference's driver calls
        (cn/simulator.py:156): sclDecoder's list, then the first entry equal to self.message (the transmitted
        u, set by the caller as in cn/simulator.py:140-141; [N] or [B, N]) is returned, the best one if the list
        does not hold it.  isCRC is accepted and ignored like in the reference."""
        return self.sclDecoder(softMess, isRCPP=isRCPP, _genie=True)

    def sclDecoder(self, softMess, isCRC=False, isRCPP=False, _genie=False):
        """List decoding with list size self.L (cn/polarCode.py:2032-2102, 969-1005).  Sets self.pathList
        (survivors sorted by metric, descending, ties in the reference's stable order), self.ANV,
        self.sortNum, self.unitCal; returns the information bits of the best path."""
        if isRCPP:
            raise NotImplementedError("puncturing is not provided")
        L = int(self.L)
        llr = self._llr(softMess)
        batch = llr.shape[:-1]
        crc_dev = bool(isCRC) and not self.bitReversed
        if crc_dev:
            self._attach_crc()
        if _genie:
            msg = np.broadcast_to(np.asarray(self.message), batch + (self.codewordLength,)).reshape(-1, self.codewordLength)
            truth = msg[:, self.polarMask == 1]
            # a "message" with ones on frozen positions can never equal a decoded word
            possible = ~(msg[:, self.polarMask == 0] != 0).any(axis=1)
        if _genie and not self.bitReversed:
            info, pm, uh = self._active.scl_decode_genie(llr.reshape(-1, self.codewordLength), L, truth, want_list=True)
            info = np.where(possible[:, None], info, uh[:, 0][:, self.polarMask == 1])
        else:
            info, pm, uh = self._active.scl_decode(llr.reshape(-1, self.codewordLength), L, want_list=True, crc=crc_dev)
        if self.bitReversed:
            uh = uh[:, :, self._rev]
            info = uh[:, 0][:, self.polarMask == 1]
        nsurv = min(L, 2 ** self.infoLen) if self.infoLen < 6 else L
        if batch == ():
            self.pathList = [Path(uh[0, j].astype(int), pm[0, j]) for j in range(nsurv)]
        else:
            self.pathList = [[Path(uh[b, j].astype(int), pm[b, j]) for j in range(nsurv)] for b in range(uh.shape[0])]
        self.pathMetrics = pm.reshape(batch + (L,))
        self.ANV, self.sortNum, self.unitCal = self._scl_counts(L)
        if _genie and self.bitReversed:
            # bit-reversed ordering: the genie's choice made here from the list the GPU returned
            words = uh[:, :, self.polarMask == 1]
            hit = (words == truth[:, None, :]).all(axis=2) & possible[:, None]
            hit[:, nsurv:] = False
            pick = np.where(hit.any(axis=1), hit.argmax(axis=1), 0)
            info = words[np.arange(words.shape[0]), pick]
        if crc_dev:
            # the GPU picked the first survivor, in list order, whose information word passes the CRC, else the
            # best one; the CRC bits are dropped from the result (cn/polarCode.py:2079-2097)
            return info[:, :self.infoLen - self.crcWidth].reshape(batch + (self.infoLen - self.crcWidth,)).astype(int)
        if isCRC:
            # bit-reversed ordering: the same choice made here from the list the GPU returned
            words = uh[:, :, self.polarMask == 1]                       # [B, L, K]
            ok = pcfun.crcCheck(words.reshape(-1, self.infoLen), self.crcPoly).reshape(words.shape[:2])
            ok[:, nsurv:] = False
            pick = np.where(ok.any(axis=1), ok.argmax(axis=1), 0)
            chosen = words[np.arange(words.shape[0]), pick][:, :self.infoLen - self.crcWidth]
            return chosen.reshape(batch + (self.infoLen - self.crcWidth,)).astype(int)
        return info.reshape(batch + (self.infoLen,)).astype(int)

    def _attach_crc(self):
        """Hand self.crcPoly to the device code object (once per polynomial)."""
        if pcfun.standardForm(self.crcPoly)[0].size != self.crcWidth:
            raise ValueError("crcWidth %d does not match crcPoly %#x" % (self.crcWidth, self.crcPoly))
        if getattr(self.native, "_crc_poly", None) != self.crcPoly:
            self.native.set_crc(self.crcPoly)
            self.native._crc_poly = self.crcPoly

    def _scl_counts(self, L):
        """Complexity counters of the reference's sclDecoder in closed form: they depend only on the
        rate profile (list size doubles per information bit until it reaches L)."""
        cur, anv, sorts, unit = 1, 0, 0, 0
        N, n = self.codewordLength, self.n
        for i in range(N):
            per_path = N - 1 if i == 0 else (1 << ((i & -i).bit_length())) - 1
            unit += per_path * cur
            anv += cur
            if self.polarMask[i] == 1:
                if 2 * cur <= L:
                    cur *= 2
                else:
                    sorts += 1
                    cur = L
        return anv, sorts + 1, unit
