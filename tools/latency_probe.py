"""Where a few-stream step spends its time: host enqueue vs device.  `python tools/latency_probe.py [streams ...]`
Per stream count: wall time until the last launch is enqueued (no synchronisation), wall time until the device is done, the number of
kernel launches of a step (adk ops; from the programs' own descriptions).  Run under `rocprofv3 --kernel-trace --stats` for the sum
of kernel durations (the difference to the wall time is launch gaps)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from audiodec_amd import synth


def main():
    dev = torch.device("cuda:0")
    counts = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]
    root = tempfile.mkdtemp()
    synth.write_model(root, bench.MODEL, bench.SEED)
    from audiodec_amd import native
    knobs = [int(v) for v in os.environ.get("PROBE_CHAIN_MIN_BLOCKS", "160").split(",")]
    for B in [(b, k) for b in counts for k in knobs]:
        B, k = B
        native.set_option("chain_min_blocks", k)
        ad = bench.build_audiodec(root, dev, B, 1)
        x = torch.from_numpy(np.stack([synth.synth_audio(5, s, bench.HOP) for s in range(B)]))[:, None, :].to(dev)
        progs = [ad.tx_encoder._encoder()] + list(ad.decoder._decoder_stages())
        launches = sum(1 for pr in progs for i in range(pr.n_ops) if "fused into" not in pr.describe_op(i, 1) and "in the launch of" not in pr.describe_op(i, 1)) + 2   # + RVQ encode, lookup
        with torch.no_grad():
            for _ in range(10):
                bench.step(ad, x)
            torch.cuda.synchronize()
            enq, tot = [], []
            for _ in range(50):
                t0 = time.perf_counter()
                bench.step(ad, x)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
        print(f"streams {B:4d} chain_min_blocks {k:3d}: launches/step {launches}  enqueue {np.median(enq):.3f} ms  device-complete {np.median(tot):.3f} ms (min {np.min(tot):.3f})", flush=True)


if __name__ == "__main__":
    main()
