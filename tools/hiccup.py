#!/usr/bin/env python3
"""Per-step host-synchronised times of a streaming loop: where do multi-millisecond stalls come from?  (tuning aid)
usage: hiccup.py <model> <streams> <steps>"""
import os, sys, time, tempfile, gc
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_bench as cb

def main():
    model, B, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dev = "cuda:0"
    with tempfile.TemporaryDirectory() as root, torch.no_grad():
        ad = cb.load(root, model, dev, B, 1)
        x = cb.audio(dev, B, ad.tx_encoder.hop)
        if os.environ.get("NOGC"):
            gc.disable()
        ts = []
        if os.environ.get("SIDE"):                      # a non-default stream (HIP graphs cannot be captured on the legacy default one)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            torch.cuda.set_stream(side)
        for i in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x))))
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        ts = np.asarray(ts)
        st = [p.graph_stats() for p in [ad.tx_encoder._encoder()]] if hasattr(ad.tx_encoder._encoder(), "graph_stats") else None
        print("graph stats (encoder):", st)
        print(f"{model} B={B}: median {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f}, max {ts.max():.3f}; steps over 3x the median:",
              [(int(i), round(float(ts[i]), 2)) for i in np.nonzero(ts > 3 * np.median(ts))[0]][:20])

main()
