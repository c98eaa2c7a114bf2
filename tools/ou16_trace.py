#!/usr/bin/env python3
"""Phase timeline of the fused conv_out + up-sampler kernel (csrc/conv_ou16.hip) from a debug build (-DADK_OU16_DBG=1).

  python tools/ou16_trace.py --build        (where hipcc is: tools/dbg/ou1/libaudiodec_hip.so)
  python tools/ou16_trace.py [streams=256]  (on the GPU box)"""
import ctypes as C
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("ADK_OU16_TRACE_LIB") or os.path.join(ROOT, "tools", "dbg", "ou1", "libaudiodec_hip.so")


EXTRA = os.environ.get("ADK_OU16_TRACE_FLAGS", "").split()


def build():
    out = os.path.dirname(LIB)
    os.makedirs(out, exist_ok=True)
    objs = []
    for s in sorted(glob.glob(os.path.join(ROOT, "audiodec_amd", "csrc", "*.hip"))):
        base = os.path.basename(s)[:-4]
        if base != "conv_ou16":
            objs.append(os.path.join(ROOT, "audiodec_amd", "csrc", ".obj", base + ".o"))
            continue
        o = os.path.join(out, base + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DADK_OU16_DBG=1"] + EXTRA + [ "-I", os.path.join(ROOT, "include"), "-c", s, "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs)
    print(LIB)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        return build()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    os.environ["ADK_LIB_PATH"] = LIB
    os.environ["ADK_SPLIT16"] = "1"
    os.environ["ADK_VOCODER_STAGES"] = "1"
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
    x = torch.from_numpy(np.stack([synth.synth_audio(1337, s, 300) for s in range(B)]))[:, None, :].to("cuda:0")
    zq = ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x)))
    for _ in range(6):
        ad.decoder.decode(zq)
    torch.cuda.synchronize()
    fn = native.lib().adk_debug_ou_trace
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    buf = (C.c_uint64 * (1024 * 8))()
    assert fn(buf, 1024 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:B, :6]
    t0 = a[:, 0].min()
    print(f"{B} workgroups (one per stream); times in us (10 ns ticks); launch span (first entry -> last exit) {(a[:, 5].max() - t0) / 100.0:.2f}")
    print(f"workgroup entry spread {(a[:, 0].max() - t0) / 100.0:.2f}; median workgroup duration {np.median(a[:, 5] - a[:, 0]) / 100.0:.2f}")
    names = ["issue: 12 LDS-DMA instructions per wave (W1, column blocks 0-2 of the tile; wave 0 also biases + c[-1])", "wait: {biases, W1, block 0} landed, barrier",
             "GEMM 1 (72 MFMAs of 16 x 16 x 32 per wave; blocks 1-5 stream in beneath it; f16 split of the fragments)", "epilogue 1: act(c), split -> LDS, second barrier",
             "GEMM 2 (72 MFMAs per wave) + 6 x 16 B stores per lane issued"]
    for i, n in enumerate(names):
        d = (a[:, i + 1] - a[:, i]) / 100.0
        print(f"  {n:95s} median {np.median(d):5.2f}   p10 {np.percentile(d, 10):5.2f}   p90 {np.percentile(d, 90):5.2f}")


if __name__ == "__main__":
    main()
