#!/usr/bin/env python3
"""The four smaller measurement configurations of SURVEY.md section 8(d) / BASELINE.json (configs 1-4) on one
MI355X (the CPU side of config 1 is timed by `bench.py --cpu-cfg1`, the only place allowed to run the oracle).  The headline
(config 5 per-GPU share) is bench.py; these are reported in profiles/ only.

    python tools/config_bench.py [--steps 200] [--warmup 20] [--out gpurun_out/configs.json]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED, HOP = 1337, 300


def load(root, model, dev, streams, max_frames):
    from audiodec_amd import synth
    from audiodec_amd.audiodec import AudioDec, assign_model
    synth.write_model(root, model, SEED)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        sr, enc, dec = assign_model(model)
        ad = AudioDec(tx_device=dev, rx_device=dev, num_streams=streams, max_frames=max_frames)
        with contextlib.redirect_stdout(io.StringIO()):
            ad.load_transmitter(enc)
            ad.load_receiver(enc, dec)
    finally:
        os.chdir(cwd)
    return ad


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def audio(dev, streams, length):
    from audiodec_amd import synth
    return torch.from_numpy(np.stack([synth.synth_audio(SEED, s, length) for s in range(streams)]))[:, None, :].to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--out", type=str, default=None)
    a = ap.parse_args()
    dev = "cuda:0"
    res = {}
    with tempfile.TemporaryDirectory() as root, torch.no_grad():
        # cfg-1: libritts_sym, one 24000-sample file, one-shot encode -> quantize -> lookup -> decode (B = 1)
        ad = load(root, "libritts_sym", dev, 1, 80)
        x = audio(dev, 1, 24000)
        f = lambda: ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x))))
        t = timed(f, max(20, a.steps // 4), 5)
        res["cfg1_libritts_sym_file_24000_B1"] = {"ms": round(1e3 * t, 3), "frames_per_s": round(80 / t, 1), "rtf": round(t / 1.0, 5)}
        del ad
        # cfg-2: symAD vctk encoder + projector + RVQ, B = 32, one hop per step
        ad = load(root, "vctk_sym", dev, 32, 1)
        x = audio(dev, 32, HOP)
        t = timed(lambda: ad.tx_encoder.quantize(ad.tx_encoder.encode(x)), a.steps, a.warmup)
        res["cfg2_vctk_encoder_rvq_B32"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(32 / t, 1)}
        del ad
        # cfg-3: encoder + RVQ + lookup + symAD decoder, B = 64
        ad = load(root, "vctk_sym", dev, 64, 1)
        x = audio(dev, 64, HOP)
        f = lambda: ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x))))
        t = timed(f, a.steps, a.warmup)
        res["cfg3_vctk_sym_full_B64"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(64 / t, 1)}
        del ad
        # cfg-4: AD-v1 vocoder only, B = 256, zq from uniformly random codes through the lookup
        ad = load(root, "vctk_v1", dev, 256, 1)
        g = torch.Generator().manual_seed(SEED)
        idx = (torch.randint(0, 1024, (8, 256, 1), generator=g) + 1024 * torch.arange(8).view(8, 1, 1)).to(dev)
        zq = ad.rx_encoder.lookup(idx)
        t = timed(lambda: ad.decoder.decode(zq), a.steps, a.warmup)
        flops = 596.8e6 * 256
        res["cfg4_v1_vocoder_B256"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(256 / t, 1),
                                       "tflops": round(flops / t / 1e12, 2)}
    print(json.dumps(res, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
