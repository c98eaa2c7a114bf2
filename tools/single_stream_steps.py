#!/usr/bin/env python3
"""N host-synchronised single-stream steps of vctk_v1 (the facade's default lowering; guard: ADK_SS_GUARD = off (default) | lazy | sync) between two bench.py-style marker launches:
run under `rocprofv3 --kernel-trace` and summarise with tools/trace_summary.py to split a step's latency into kernel time and gaps.
usage: single_stream_steps.py [streams=1] [steps=40]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ADK_VOCODER_STAGES", "1")
import numpy as np
import torch
import bench
from audiodec_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
root = tempfile.mkdtemp()
synth.write_model(root, bench.MODEL, bench.SEED)
G = os.environ.get("ADK_SS_GUARD", "off")
if G == "sync":
    os.environ["ADK_GUARD_MODE"] = "sync"
ad = bench.build_audiodec(root, dev, B, 1, guard=(False if G == "off" else True))
x = torch.from_numpy(np.stack([synth.synth_audio(5, s, bench.HOP) for s in range(B)]))[:, None, :].to(dev)
with torch.no_grad():
    for _ in range(20):
        bench.step(ad, x)
    torch.cuda.synchronize()
    torch.arange(bench.PMC_MARKER_N, device=dev); torch.cuda.synchronize()
    lat = []
    for _ in range(N):
        t0 = time.perf_counter()
        bench.step(ad, x)
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t0))
    torch.arange(bench.PMC_MARKER_N, device=dev); torch.cuda.synchronize()
print(f"guard {G}: ", end="")
print(f"streams {B}: median {np.median(lat):.4f} ms, min {np.min(lat):.4f} ms per host-synchronised step ({N} steps)")
