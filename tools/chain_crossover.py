#!/usr/bin/env python3
"""Where does a residual chain as ONE launch (conv_rb16) beat the per-op launches?  Times the encoder and the vocoder of vctk_v1
at several stream counts with chains on / off on the same objects (adk_set_option chain_max_channels), one HIP stream.

  python tools/chain_crossover.py [B ...]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ADK_SPLIT16"] = "1"
os.environ["ADK_VOCODER_STAGES"] = "1"
import contextlib, io
import numpy as np
import torch
from audiodec_amd import native, synth
from audiodec_amd.audiodec import AudioDec, assign_model


def timed(fn, n=60, w=15):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 192, 256]
    tmp = tempfile.mkdtemp()
    synth.write_model(tmp, "vctk_v1", 1337)
    os.chdir(tmp)
    sr, enc, dec = assign_model("vctk_v1")
    print("B  enc_chain_ms enc_perop_ms  voc_chain_ms voc_perop_ms")
    for B in Bs:
        ad = AudioDec(tx_device="cuda:0", rx_device="cuda:0", num_streams=B, max_frames=1, guard=False)
        with contextlib.redirect_stdout(io.StringIO()):
            ad.load_transmitter(enc); ad.load_receiver(enc, dec)
        x = torch.from_numpy(np.stack([synth.synth_audio(1337, s, 300) for s in range(B)]))[:, None, :].to("cuda:0")
        zq = ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x)))
        res = []
        for maxc in (128, 0):
            native.set_option("chain_max_channels", maxc)
            native.set_option("chain_min_blocks", 0)
            res.append((timed(lambda: ad.tx_encoder.encode(x)), timed(lambda: ad.decoder.decode(zq))))
        print(f"{B:4d}  {res[0][0]:.4f} {res[1][0]:.4f}   {res[0][1]:.4f} {res[1][1]:.4f}")
        del ad
    native.set_option("chain_max_channels", 128)


if __name__ == "__main__":
    main()
