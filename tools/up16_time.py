#!/usr/bin/env python3
"""The last up-sampling stage (conv_up16: LeakyReLU -> ConvTranspose1d 64 -> 32, K 6, stride 3, + bias) by itself: back-to-back launches timed by HIP
events inside the library (layers.time_kernel), for streams x frames per call given as B:frames pairs.  `hot` in front of the pairs: two seconds of
matrix-core load (torch.mm, bf16 8192^3) right before every timing -- the state bench.py's `transposed_conv_alone` leg finds the GPU in.
usage: up16_time.py [hot] 256:5 256:1 64:5 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiodec_amd import layers, native

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
w = torch.randn(64, 32, 6, generator=g) / 128 ** 0.5
bias = torch.randn(32, generator=g) * 0.1
args = sys.argv[1:]
hot = bool(args) and args[0] == "hot"
if hot:
    args = args[1:]
    ha = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def load(seconds):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            torch.mm(ha, ha)
        torch.cuda.synchronize()


for spec in args or ["256:5"]:
    B, fps = (int(v) for v in spec.split(":"))
    t_in = 100 * fps
    m = layers.CausalConvTranspose1d(64, 32, 6, 3, device=dev, batch=B, max_len=t_in).load(w, bias)
    m.set_activation("LeakyReLU", 0.1)
    m.impl = native.IMPL_SPLIT16
    m.inference(torch.randn(B, 64, t_in, generator=g))
    torch.cuda.synchronize()
    us = []
    for _ in range(5):
        if hot:
            load(2.0)
        us.append(m.time_kernel(t_in, 300))
    us.sort()
    byts = 4.0 * (64 * (t_in + 1) + 32 * 3 * t_in) * B + 4.0 * 64 * 32 * 6
    print(f"{'after 2 s of matrix-core load, ' if hot else ''}{B} streams x {fps} frames: {m.last_kernel} median {us[2]:.2f} us (min {us[0]:.2f}) per launch, "
          f"{byts / 1e6:.1f} MB -> {byts / us[2] / 1e3:.0f} GB/s = {byts / us[2] / 1e3 / 8000:.3f} of 8 TB/s")
