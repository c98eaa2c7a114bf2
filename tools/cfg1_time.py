#!/usr/bin/env python3
"""Per-call wall times of the config-1 round trip (libritts_sym, one stream, 80 frames per call) -- tuning aid."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_bench as cb

def main():
    dev = "cuda:0"
    with tempfile.TemporaryDirectory() as root, torch.no_grad():
        ad = cb.load(root, "libritts_sym", dev, 1, 80)
        x = cb.audio(dev, 1, 24000)
        ts = []
        for i in range(int(os.environ.get("N_ITER", "12"))):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            z = ad.tx_encoder.encode(x); torch.cuda.synchronize(); t1 = time.perf_counter()
            idx = ad.tx_encoder.quantize(z); torch.cuda.synchronize(); t2 = time.perf_counter()
            zq = ad.rx_encoder.lookup(idx); torch.cuda.synchronize(); t3 = time.perf_counter()
            y = ad.decoder.decode(zq); torch.cuda.synchronize(); t4 = time.perf_counter()
            ts.append([1e3 * (b - a) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))])
        for t in ts:
            print("encode %.3f  quantize %.3f  lookup %.3f  decode %.3f ms" % tuple(t))

main()
