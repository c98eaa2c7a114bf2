#!/usr/bin/env python3
"""Micro-benchmark of one fused causal conv launch (tuning aid, not part of the product).

usage: conv_bench.py [--shape s0|s1|s2|s3|...] [--cfg -1..5] [--batch 256] [--iters 50]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audiodec_amd import native  # noqa: E402
from audiodec_amd.native import ConvDesc, RingView  # noqa: E402
from audiodec_amd.program import pack_mfma, pack_split16  # noqa: E402

SHAPES = {  # cin_g, cout_g, groups, taps, stride, dil, t_out, up, act
    "s0": (256, 256, 3, 11, 1, 5, 5, 1, 2), "s1": (128, 128, 3, 11, 1, 5, 25, 1, 2),
    "s2": (64, 64, 3, 11, 1, 5, 100, 1, 2), "s3": (32, 32, 3, 11, 1, 5, 300, 1, 2),
    "e3": (256, 256, 1, 7, 1, 9, 5, 1, 1), "e2": (128, 128, 1, 7, 1, 9, 25, 1, 1),
    "e1": (64, 64, 1, 7, 1, 9, 100, 1, 1), "e0": (32, 32, 1, 7, 1, 9, 300, 1, 1),
    "up0": (512, 1280, 1, 2, 1, 1, 1, 5, 2), "up3": (64, 96, 1, 2, 1, 1, 100, 3, 2), "up3s": (64, 96, 1, 2, 1, 1, 100, 3, 0), "up3x5": (64, 96, 1, 2, 1, 1, 500, 3, 2), "up3x20": (64, 96, 1, 2, 1, 1, 2000, 3, 2),
    "r0": (32, 32, 1, 1, 1, 1, 300, 1, 1), "r1": (64, 64, 1, 1, 1, 1, 100, 1, 1),
    "d3": (256, 512, 1, 10, 5, 1, 1, 1, 0), "p": (512, 64, 1, 3, 1, 1, 1, 1, 0),
}


def view(t, rows, ch, cur):
    v = RingView()
    v.base, v.rows, v.channels, v.cursor, v.ch_off = (t.data_ptr() if t is not None else None), rows, ch, cur, 0
    return v


def run(shape, cfg, B, iters, impl=native.IMPL_MFMA, check=False):
    lib = native.lib()
    cin_g, cout_g, groups, taps, stride, dil, t_out, up, act = SHAPES[shape]
    dev = "cuda:0"
    hist = (taps - 1) * dil
    rows = hist + t_out * stride
    cin_t = cin_g * groups
    M = cout_g * groups
    cout_real = M // up
    ring = torch.randn(B, rows, cin_t, device=dev)
    w = torch.randn(M, taps * cin_g, device=dev) / (taps * cin_g) ** 0.5
    bias = torch.randn(M, device=dev)
    out = torch.empty(B, t_out * up, cout_real, device=dev)
    d = ConvDesc()
    d.cin_g, d.cout_g, d.groups, d.taps, d.stride, d.dilation, d.hist = cin_g, cout_g, groups, taps, stride, dil, hist
    d.up, d.cout_real, d.in_group_stride, d.res_group_stride = up, cout_real, cin_g, cout_g
    d.act_in, d.act_in_slope, d.act_out = act, 0.1, 0
    wf = (pack_split16(w.cpu(), groups) if impl in (native.IMPL_SPLIT16, native.IMPL_SPLIT16_ROWS, native.IMPL_SPLIT16_SK) else pack_mfma(w.cpu(), groups)).to(dev)
    d.w, d.w_frag, d.bias = w.data_ptr(), wf.data_ptr(), bias.data_ptr()
    lib.adk_set_conv_cfg(cfg)
    st = native.current_stream(dev)
    vin, vout, vres = view(ring, rows, cin_t, hist), view(out, t_out * up, cout_real, 0), view(None, 0, 0, 0)

    def go():
        native.check(lib.adk_causal_conv(C.byref(d), vin, vout, vres, B, t_out, impl, st), "conv")
    if check:
        go()
        got = out.clone()
        wf2 = pack_mfma(w.cpu(), groups).to(dev)
        d.w_frag = wf2.data_ptr()
        native.check(lib.adk_causal_conv(C.byref(d), vin, vout, vres, B, t_out, native.IMPL_MFMA, st), "conv")
        print(f"   max|impl {impl} - stream-K f32| = {float((got - out).abs().max()):.3e}  (|out|max {float(out.abs().max()):.2f})")
        d.w_frag = wf.data_ptr()
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    flops = 2.0 * M * taps * cin_g * t_out * B
    return us, flops / us / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="s0,s1,s2,s3")
    ap.add_argument("--cfg", default="-1")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--impl", type=int, default=native.IMPL_MFMA, help="0 auto, 2 stream-K, 3 rows-in-LDS; split-f16: 4 auto, 5 rows-in-LDS, 6 stream-K")
    ap.add_argument("--check", action="store_true", help="compare the output with the f32 stream-K kernel")
    a = ap.parse_args()
    for sh in a.shape.split(","):
        for cfg in a.cfg.split(","):
            us, tf = run(sh, int(cfg), a.batch, a.iters, a.impl, a.check)
            print(f"{sh:5s} impl {a.impl} cfg {cfg:>2s}  B={a.batch}  {us:9.1f} us  {tf:7.1f} TFLOP/s", flush=True)
