#!/usr/bin/env python3
"""How long does the HOST need to enqueue one pipeline step (tuning aid)?  Issues N steps without synchronising and
reports host time per step next to the synchronised time per step."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from audiodec_amd import synth

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    stages = sys.argv[2] if len(sys.argv) > 2 else "2"
    os.environ["ADK_SPLIT16"] = "1"; os.environ["ADK_VOCODER_STAGES"] = stages
    dev = "cuda:0"
    tmp = tempfile.TemporaryDirectory()
    synth.write_model(tmp.name, bench.MODEL, bench.SEED)
    ad = bench.build_audiodec(tmp.name, dev, B, 1)
    xs = [torch.from_numpy(np.stack([synth.synth_audio(bench.SEED + j, s, 300) for s in range(B)]))[:, None, :].to(dev) for j in range(4)]
    pipe = bench.TxRxPipeline(ad, dev)
    with torch.no_grad():
        pipe.enter()
        for i in range(20): pipe.step(xs[i % 4])
        pipe.exit(); torch.cuda.synchronize()
        for n in (20, 100):
            t0 = time.perf_counter()
            pipe.enter()
            for i in range(n): pipe.step(xs[i % 4])
            pipe.exit()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"B={B} stages={stages} n={n}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")

main()
