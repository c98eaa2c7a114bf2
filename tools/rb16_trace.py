#!/usr/bin/env python3
"""Phase timeline of the residual-chain kernel (csrc/conv_rb16.hip) from a DEBUG build of the library.

  python tools/rb16_trace.py [streams=256] [dbg flags=1]        (--build f1 f2 ...: only build the debug libraries)

Builds csrc/conv_rb16.hip with -DADK_RB16_DBG=<flags> (1 = stamps; +2 no weight loads, +4 no MFMAs, +8 one B read: knock-outs)
into tools/dbg/rb<flags>/libaudiodec_hip.so (the product library is untouched),
loads vctk_v1 at `streams` streams, runs a few steps and prints, for the chain launches of the LAST step, the median over
workgroups of the time wave 0 spent in each phase (s_memrealtime stamps, 10 ns ticks -> us)."""
import ctypes as C
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_debug(flags=1):
    """Debug build of the library with -DADK_RB16_DBG=<flags> under tools/dbg/ (not tracked; it travels to the GPU box with the
    snapshot, so build it where the compiler is -- `python tools/rb16_trace.py --build 1 3 5 9` -- and run it there)."""
    out = os.path.join(ROOT, "tools", "dbg", f"rb{flags}")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libaudiodec_hip.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "audiodec_amd", "csrc", "*.hip")))
    objs, procs = [], []
    for s in srcs:
        base = os.path.basename(s)[:-4]
        if base != "conv_rb16":                     # every other object is the product build's
            objs.append(os.path.join(ROOT, "audiodec_amd", "csrc", ".obj", base + ".o"))
            continue
        o = os.path.join(out, base + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DADK_RB16_DBG={flags}",
                                       "-I", os.path.join(ROOT, "include"), "-c", s, "-o", o]))
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs)
    return lib


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        for f in sys.argv[2:]:
            print(build_debug(int(f)))
        return
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib_path = os.path.join(ROOT, "tools", "dbg", f"rb{flags}", "libaudiodec_hip.so")
    if "ADK_TRACE_LIB" in os.environ:               # any other debug build (tools/alt_build.sh <name> conv_rb16 -DADK_RB16_DBG=1 ...)
        os.environ["ADK_LIB_PATH"] = os.environ["ADK_TRACE_LIB"]
    else:
        os.environ["ADK_LIB_PATH"] = lib_path if os.path.exists(lib_path) else build_debug(flags)
    os.environ["ADK_SPLIT16"] = "1"
    os.environ.setdefault("ADK_VOCODER_STAGES", "1")
    import numpy as np
    import torch
    from audiodec_amd import native, synth
    from audiodec_amd.audiodec import AudioDec, assign_model
    tmp = tempfile.mkdtemp()
    synth.write_model(tmp, "vctk_v1", 1337)
    os.chdir(tmp)
    sr, enc, dec = assign_model("vctk_v1")
    import contextlib, io
    ad = AudioDec(tx_device="cuda:0", rx_device="cuda:0", num_streams=B, max_frames=1)
    with contextlib.redirect_stdout(io.StringIO()):
        ad.load_transmitter(enc); ad.load_receiver(enc, dec)
    x = torch.from_numpy(np.stack([synth.synth_audio(1337, s, 300) for s in range(B)]))[:, None, :].to("cuda:0")
    lib = native.lib()
    fn = lib.adk_debug_rb_trace
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    n = 16 * 1024 * 32
    buf = (C.c_uint64 * n)()
    for it in range(5):
        z = ad.tx_encoder.encode(x)
        y = ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(z)))
    torch.cuda.synchronize()
    assert fn(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 1024, 32).astype(np.int64)
    # the last 6 launches (slots are assigned round-robin): order within a step = encoder blocks 0,1,2, vocoder stages 1,2,3
    names = ["enc.block0 C32 K7+1x1", "enc.block1 C64", "enc.block2 C128", "voc.stage1 C128 K11 g3", "voc.stage2 C64", "voc.stage3 C32"]
    # number of launches so far = 6 per step * steps (incl. warm-up); find slots by recency of stamp 0
    last = a[:, :, 0].max(axis=1)
    order = np.argsort(last)[-6:]
    for slot, name in zip(order, names):
        t = a[slot]
        wgs = (t[:, 0] > 0) & (t[:, 0] >= t[:, 0].max() - 100000)          # workgroups of this launch (within 1 ms)
        t = t[wgs]
        t0 = t[:, 0].min()
        print(f"== {name}: {wgs.sum()} workgroups traced; launch span {(t.max() - t0) / 100.0:.1f} us; "
              f"workgroup start spread {(t[:, 0].max() - t0) / 100.0:.1f} us; median workgroup duration {np.median(t.max(axis=1) - t[:, 0]) / 100.0:.1f} us")
        print("   stage-in %.2f us" % (np.median(t[:, 1] - t[:, 0]) / 100.0))
        prev = t[:, 1]
        for k in range(6):
            st = [t[:, 2 + 4 * k + i] for i in range(4)]
            if (st[0] <= 0).all():
                break
            seg = [np.median(st[0] - prev) / 100.0, np.median(st[1] - st[0]) / 100.0]
            if (st[3] > 0).any():
                seg += [np.median(st[2] - st[1]) / 100.0, np.median(st[3] - st[2]) / 100.0]
                prev = st[3]
            print(f"   conv {k}: mfma loop {seg[0]:.2f}  epilogue(regs, ring stores) {seg[1]:.2f}" + (f"  wait at barrier {seg[2]:.2f}  lds write + history + barrier {seg[3]:.2f}" if len(seg) > 2 else ""))


if __name__ == "__main__":
    main()
