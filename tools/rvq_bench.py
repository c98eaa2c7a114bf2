#!/usr/bin/env python3
"""Time adk_rvq_encode / adk_rvq_lookup alone (tuning aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audiodec_amd import layers
g = torch.Generator().manual_seed(0)
embeds = [torch.randn(64, 1024, generator=g) * 0.8 ** i for i in range(8)]
rvq = layers.ResidualVQ(embeds)
rvq.initial()
for n in (1, 32, 256, 2048):
    x = torch.randn(n, 1, 64, generator=g).cuda()
    for _ in range(3):
        q, idx = rvq.forward_index(x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        q, idx = rvq.forward_index(x, True)
    e1.record(); torch.cuda.synchronize()
    print(f"rvq_encode rows={n}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us")
