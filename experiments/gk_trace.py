#!/usr/bin/env python3
"""Phase timeline of conv_gk16 (csrc/conv_mfma.hip) from the product library: ADK_GK16_DBG=16 makes wave 0 of every workgroup stamp
s_memrealtime at its phase boundaries; this runs a few serial steps of the vctk_v1 vocoder at 256 streams and prints, for the LAST
conv_gk16 launch (blocks.0.convs2.2), the launch span and per phase the median / max over workgroups.   python tools/gk_trace.py [streams] [extra dbg bits]"""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ["ADK_GK16_DBG"] = str(16 | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
os.environ["ADK_SPLIT16"] = "1"
os.environ["ADK_GK16"] = "1"                 # (the kernel is opt-in)
os.environ["ADK_VOCODER_STAGES"] = "1"
import numpy as np
import torch
from audiodec_amd import native, synth
from audiodec_amd.audiodec import AudioDec, assign_model

tmp = tempfile.mkdtemp()
synth.write_model(tmp, "vctk_v1", 1337)
os.chdir(tmp)
sr, enc, dec = assign_model("vctk_v1")
ad = AudioDec(tx_device="cuda:0", rx_device="cuda:0", num_streams=B, max_frames=1, guard=False)
ad.load_transmitter(enc); ad.load_receiver(enc, dec)
g = torch.Generator().manual_seed(1)
idx = (torch.randint(0, 1024, (8, B, 1), generator=g) + 1024 * torch.arange(8).view(8, 1, 1)).to("cuda:0")
zq = ad.rx_encoder.lookup(idx)
for _ in range(6):
    ad.decoder.decode(zq)
torch.cuda.synchronize()
n = 512 * 8
buf = (C.c_ulonglong * n)()
lib = native.lib()
lib.adk_debug_gk_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.adk_debug_gk_trace(buf, n) == 0
t = np.array(buf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
live = t[:, 0] > 0
t = t[live]
print("workgroups", len(t), "launch span %.1f us, start spread %.1f us" % ((t[:, 6].max() - t[:, 0].min()) / 100.0, (t[:, 0].max() - t[:, 0].min()) / 100.0))
names = ["entry -> prologue issued", "prologue issued -> first chunk landed", "first chunk landed -> loop done",
         "loop done -> slabs published, all parts arrived", "own slab reduced + epilogue stores issued", "-> exit (counters)"]
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print(f"{nm:50s} median {np.median(d):7.2f}  max {d.max():7.2f} us")
d = (t[:, 3] - t[:, 0].min()) / 100.0
print("loop done at (from launch start): median %.2f, first %.2f, last %.2f us" % (np.median(d), d.min(), d.max()))
