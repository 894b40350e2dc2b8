"""BPSK / AWGN channel with the call surface of the reference's cn/channel.py.

Single-frame calls keep the reference's arithmetic and its use of numpy's global generator, so a
script seeded with np.random.seed() sees the same noise.  The batched Monte-Carlo path
(`llrBatch`) generates Philox noise and LLRs on the GPU through the HIP library.
"""
import numpy as np


class Channel:
    def __init__(self, modulation, snrdB, snrType, Rc):
        self.modulation = modulation
        self.M = 2 if modulation.lower() == "bpsk" else None
        if self.M is None:
            raise NotImplementedError("only BPSK is supported (cn/channel.py:6 leaves other modulations undefined)")
        self.noisePower = self.calc_N0(snrdB, snrType, Rc)

    def calc_N0(self, snrdB, snrType, Rc):
        """N0 from Es/N0 ('SNR') or Eb/N0 (anything else, e.g. 'SNRb') -- cn/channel.py:9-14."""
        if snrType.lower() == "snr":
            return 1 / pow(10, snrdB / 10)
        return 1 / (np.log2(self.M) * Rc * pow(10, snrdB / 10))

    # -- cn/channel.py:16-31 -------------------------------------------------------------------
    def modulate(self, m):
        m = np.asarray(m)
        if m.ndim == 1:
            return [int(1 - 2 * x) for x in m]  # the reference returns a Python list
        return 1 - 2 * m.astype(np.int64)

    def modulateMultiple(self, ms):
        return [self.modulate(m) for m in ms]

    # -- cn/channel.py:33-51 -------------------------------------------------------------------
    def addNoise(self, signal):
        signal = np.asarray(signal)
        return signal + np.sqrt(self.noisePower / 2) * np.random.standard_normal(signal.shape[-1] if signal.ndim == 1 else signal.shape)

    def addNoise2(self, signal1, signal2):
        noise = np.sqrt(self.noisePower / 2) * np.random.standard_normal(len(signal1))
        return signal1 + noise, signal2 + noise

    def addNoiseMultiple(self, signals):
        noise = np.sqrt(self.noisePower / 2) * np.random.standard_normal(len(signals[0]))
        for i in range(len(signals)):
            signals[i] = signals[i] + noise
        return signals

    # -- cn/channel.py:53-69 -------------------------------------------------------------------
    def calcLLR(self, c):
        return (4 / self.noisePower) * np.asarray(c, dtype=np.float64)

    def calcLLRMultiple(self, cs):
        return [self.calcLLR(c) for c in cs]

    # -- batched device path ---------------------------------------------------------------------
    def llrBatch(self, x, seed=2024, frame0=0, dtype=np.float32):
        """x: [B, N] code bits (0/1).  Returns LLRs [B, N] computed on the GPU: BPSK, Philox4x32-10
        Gaussian noise keyed by (seed, frame0 + row), llr = 4/N0 * y.  Raises if the HIP library or a
        GPU is missing."""
        import _native

        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.uint8)
        return _native.scratch_code(x.shape[1]).channel_llr(x, self.noisePower, seed, frame0, dtype)
