"""Stand-alone SC demo / sweep with the flow of the reference's cn/main.py, on the GPU.

cn/main.py:214-286 builds a (1024,512) dega code, encodes one random message as mul_matrix(u[indicesRev]),
adds AWGN at 1 dB Eb/N0, SC-decodes in bit-reversed order and prints the outcome.  This script does the same
for any (N, K), one frame or many, in the natural ordering of polarCode.py or (--bitreversed) in main.py's.

    python main.py --N 128 --K 64 --snr 1 1.5 2 2.5 3 --frames 100000
"""
import argparse

import numpy as np

from channel import Channel
from polarCode import PolarCode


def run(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--K", type=int, default=512)
    ap.add_argument("--dsnr", type=float, default=2.5, help="design SNR of the dega construction (cn/main.py:222 uses 2.5)")
    ap.add_argument("--snr", type=float, nargs="+", default=[1.0], help="Eb/N0 points in dB (cn/main.py:239 uses 1)")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--bitreversed", action="store_true", help="cn/main.py's ordering: encode u[indicesRev], decide uHat[bitreversed(i)]")
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args(argv)
    if a.seed is not None:
        np.random.seed(a.seed)
    pc = PolarCode(a.N, a.K, "dega", a.dsnr)
    pc.bitReversed = a.bitreversed
    pc.decoderType = "sc"
    results = []
    for snr in a.snr:
        ch = Channel("BPSK", snr, "SNRb", a.K / a.N)
        message = np.random.randint(0, 2, size=(a.frames, a.K), dtype=int)
        x = pc.polarEncode(message)
        y = ch.addNoise(ch.modulate(x))
        llr = ch.calcLLR(y)
        decoded = pc.decode(llr)
        if a.frames == 1:  # the prints of cn/main.py:281-286
            u = pc.rateProfile(message[0])
            print(f"decoded =   {decoded[0]}")
            print(f"message =   {message[0]}")
            print(f"polarMask = {pc.polarMask}")
            print(f"u =         {u}")
            print(f"uHat =      {pc.uHat[0]}")
            print((message[0] == decoded[0]).all())
        errs = np.sum(message != decoded, axis=1)
        fer, ber = float(np.mean(errs > 0)), float(np.sum(errs) / (a.frames * a.K))
        if a.frames > 1:
            print(f"SNR = {snr} dB  frames = {a.frames}  FER = {fer:0.2e}  BER = {ber:0.2e}")
        results.append((snr, fer, ber))
    return results


if __name__ == "__main__":
    run()
