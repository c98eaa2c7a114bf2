#!/usr/bin/env python3
"""The six wide grouped convs of the v1 vocoder's first stage (blocks.0.convs1/2.*) and every other stream-K launch of a vctk_v1 step, serial, by
per-op HIP events -- with the wide-tile kernel (conv_wt16) off / on (128 rows) / on (256 rows), alternating in one process.
usage: wt16_bench.py [streams=256] [rounds=2]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ADK_VOCODER_STAGES", "2")
import numpy as np
import torch
import bench
from audiodec_amd import synth, native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
root = tempfile.mkdtemp()
synth.write_model(root, bench.MODEL, bench.SEED)
ad = bench.build_audiodec(root, dev, B, 1, guard=False)
xs = [torch.from_numpy(np.stack([synth.synth_audio(5 + j, s, bench.HOP) for s in range(B)]))[:, None, :].to(dev) for j in range(4)]
for r in range(R):
    for mode, rows, nb in ((0, 128, 2), (1, 128, 2), (1, 128, 3), (1, 128, 4), (1, 256, 2), (1, 256, 3), (2, 256, 3)):
        native.set_option("wt16", mode); native.set_option("wt16_rows", rows); native.set_option("wt16_buffers", nb)
        with torch.no_grad():
            rows_ = bench.op_profile(ad, xs, B, 10, 1)
        st0 = [q for q in rows_ if q["name"].startswith("blocks.0.convs")]
        sk = [q for q in rows_ if q["kernel"].startswith(("conv_sk16", "conv_wt16")) and not q["name"].startswith("blocks.0.convs")]
        print(f"round {r} wt16={mode} rows={rows} buffers={nb}: stage-0 convs {1e3 * sum(q['ms'] for q in st0):7.1f} us ({', '.join(f'{1e3 * q[chr(109) + chr(115)]:.1f}' for q in st0)}) [{st0[0]['kernel']}]; "
              f"other stream-K launches {1e3 * sum(q['ms'] for q in sk):7.1f} us; whole step {1e3 * sum(q['ms'] for q in rows_):7.1f} us (each figure contains one event record)")
        if mode == 2:
            print("   ", ", ".join(f"{q['name']}={1e3 * q['ms']:.1f}[{q['kernel'][5:9]}]" for q in sk))
native.set_option("wt16", 1); native.set_option("wt16_rows", 128); native.set_option("wt16_buffers", 2)
