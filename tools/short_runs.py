#!/usr/bin/env python3
"""What a SHORT timed region of bench.py costs beyond K steady-state steps (the driver runs --steps 20 --warmup 5).

  python tools/short_runs.py            (on the GPU box; same models, streams, pipeline object as bench.py)

Prints (1) the first eight 20-step regions after a 5-step warm-up, (2) elapsed(K) for K = 1..100 (median of 5 regions each) with
the host's enqueue time, (3) 20-step regions after 1 s of idle + W warm-up steps.  Results: profiles/r3_short_runs.md."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ADK_SPLIT16", "1"); os.environ.setdefault("ADK_VOCODER_STAGES", "2")
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
import bench
from audiodec_amd import synth, configs

dev = "cuda:0"; torch.cuda.set_device(0)
sr, enc_tag, tx_steps, dec_tag, rx_steps = configs.alias(bench.MODEL)
tmp = tempfile.TemporaryDirectory()
for tag, st in ((enc_tag, tx_steps), (dec_tag, rx_steps)):
    synth.write_experiment(tmp.name, tag, st, bench.SEED, sd=synth.synth_state_dict(tag, bench.SEED))
B = 256
ad = bench.build_audiodec(tmp.name, dev, B, 1)
xs = [torch.from_numpy(np.stack([synth.synth_audio(bench.SEED + j, s, bench.HOP) for s in range(B)]))[:, None, :].to(dev) for j in range(8)]
pipe = bench.TxRxPipeline(ad, dev)


def region(K):
    """K pipeline steps between two device synchronisations, as bench.py times them: (elapsed ms, host enqueue ms)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.enter()
    for i in range(K):
        pipe.step(xs[i % 8])
    t_host = time.perf_counter() - t0
    pipe.exit()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, t_host * 1e3


with torch.no_grad():
    region(5)
    for j in range(8):
        e, h = region(20)
        print(f"region {j} of 20 steps after a 5-step warm-up: {e:.3f} ms ({e / 20:.4f} per step)")
    for K in (1, 2, 3, 4, 6, 10, 20, 40, 100):
        r = [region(K) for _ in range(5)]
        e = sorted(x[0] for x in r)[2]; h = sorted(x[1] for x in r)[2]
        print(f"K={K:4d}  elapsed {e:8.3f} ms  ({e / K:6.3f} per step)   host enqueue {h:8.3f} ms ({h / K:6.3f} per step)")
    for W in (5, 50, 300, 5, 50, 300, 5, 50, 300):
        time.sleep(1.0)
        region(W)
        e, h = region(20)
        print(f"idle 1 s, {W:3d} warm-up steps, then 20 timed: {e / 20:.4f} ms per step = {256 * 20 / e:.1f} k frames/s")
