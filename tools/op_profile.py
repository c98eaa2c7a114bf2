#!/usr/bin/env python3
"""Per-op HIP-event times of one model at a given (streams, frames per call) -- tuning aid.
usage: op_profile.py <model> <streams> <frames>   (ADK_SPLIT16=1 for the split kernels)"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_bench as cb

def main():
    model, B, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dev = "cuda:0"
    with tempfile.TemporaryDirectory() as root, torch.no_grad():
        ad = cb.load(root, model, dev, B, F)
        hop = ad.tx_encoder.hop
        x = cb.audio(dev, B, F * hop)
        progs = {"enc": ad.tx_encoder._encoder()}
        dec = ad.decoder
        stages = dec._decoder_stages() if hasattr(dec, "_decoder_stages") else [dec._decoder()]
        for i, p in enumerate(stages):
            progs[f"dec{i}"] = p
        for p in progs.values():
            p.set_profiling(True)
        acc = {k: np.zeros(p.n_ops) for k, p in progs.items()}
        n = 5
        for it in range(n + 2):
            y = ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x))))
            if it >= 2:
                for k, p in progs.items():
                    acc[k] += np.asarray(p.last_op_ms())
        tot = 0.0
        for k, p in progs.items():
            for i in range(p.n_ops):
                ms = acc[k][i] / n
                tot += ms
                print(f"{k:5s} {p.op_names[i]:44s} {p.describe_op(i, F):18s} {1e3 * ms:9.1f} us")
        print("total", round(tot, 3), "ms")

main()
