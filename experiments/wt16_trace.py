#!/usr/bin/env python3
"""Phase timeline of the wide-tile kernel (csrc/conv_wt16.hip) from a debug build (-DADK_WT16_DBG=1): per-workgroup wall-clock stamps of wave 0 of the
LAST conv_wt16 launch of a vctk_v1 decoder step (blocks.0.convs2.2).
  python tools/wt16_trace.py --build          (where hipcc is: tools/dbg/wt1/libaudiodec_hip.so)
  python tools/wt16_trace.py [streams=256] [rows=128] [buffers=2]   (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "dbg", "wt1", "libaudiodec_hip.so")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        return subprocess.check_call(["bash", os.path.join(ROOT, "tools", "alt_build.sh"), "wt1", "conv_wt16", "-DADK_WT16_DBG=1"])
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    os.environ["ADK_LIB_PATH"] = LIB
    os.environ["ADK_SPLIT16"] = "1"
    os.environ["ADK_VOCODER_STAGES"] = "2"
    import contextlib, io
    import numpy as np
    import torch
    from audiodec_amd import native, synth
    from audiodec_amd.audiodec import AudioDec, assign_model
    tmp = tempfile.mkdtemp()
    synth.write_model(tmp, "vctk_v1", 1337)
    os.chdir(tmp)
    sr, enc, dec = assign_model("vctk_v1")
    ad = AudioDec(tx_device="cuda:0", rx_device="cuda:0", num_streams=B, max_frames=1, guard=False)
    with contextlib.redirect_stdout(io.StringIO()):
        ad.load_transmitter(enc); ad.load_receiver(enc, dec)
    native.set_option("wt16", 1); native.set_option("wt16_rows", rows); native.set_option("wt16_buffers", nb)
    x = torch.from_numpy(np.stack([synth.synth_audio(1337, s, 300) for s in range(B)]))[:, None, :].to("cuda:0")
    zq = ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x)))
    for _ in range(6):
        ad.decoder.decode_stage(0, zq)
    torch.cuda.synchronize()
    fn = native.lib().adk_debug_wt_trace
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    buf = (C.c_uint64 * (1024 * 8))()
    assert fn(buf, 1024 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    a = a[a[:, 0] > 0]
    a = a[a[:, 0] > a[:, 0].max() - 20000]          # the stamps of the LAST launch (earlier launches with more workgroups leave theirs behind)
    a = a[:, [0, 1, 2, 3, 7, 4, 5, 6]]              # in time order: ... 3 loop done, 7 slabs published, 4 siblings arrived ...
    t0 = a[:, 0].min()
    print(f"streams {B}, tile rows {rows}, {nb} stage buffers: {len(a)} workgroups; launch span (first entry -> last exit) {(a[:, 7].max() - t0) / 100.0:.2f} us; entry spread {(a[:, 0].max() - t0) / 100.0:.2f}; median workgroup duration {np.median(a[:, 7] - a[:, 0]) / 100.0:.2f}")
    names = ["prologue: addresses, first stages' copies issued", "first stage landed (first barrier)", "K loop", "tile -> LDS halves, slabs published (stores drained)",
             "wait for the siblings", "own slab reduced + finished (stores issued)", "counters, exit"]
    for i, n in enumerate(names):
        d = (a[:, i + 1] - a[:, i]) / 100.0
        print(f"  {n:60s} median {np.median(d):6.2f}   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")


if __name__ == "__main__":
    main()
