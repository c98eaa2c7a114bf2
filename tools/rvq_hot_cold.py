"""Why two clocks disagree about the residual-VQ search at 256 rows: HIP events over back-to-back launches said 31.6 us, the
rocprofv3 dispatch duration inside a bench step 44.8 us.  Measured here with events around ONE launch: (a) back to back -- the 2 MB of codes are in
every XCD's L2 from the launch before; (b) after a pass over a buffer much larger than the L2s and the Infinity Cache's share (what the rest of a
pipeline step does between two searches: ~0.7 GB of ring and weight traffic); (c) after a 64 MB pass (evicts the L2s, not the Infinity Cache)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from audiodec_amd import layers

def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    embeds = [torch.randn(64, 1024, generator=g) * (0.8 ** i) for i in range(8)]
    rvq = layers.ResidualVQ(embeds, device=dev)
    big = torch.empty(512 * 1024 * 1024 // 4, device=dev)         # 512 MB
    mid = torch.empty(64 * 1024 * 1024 // 4, device=dev)          # 64 MB
    for n in (1, 32, 256):
        x = torch.randn(1, n, 64, generator=g).to(dev)
        for _ in range(20):
            rvq.forward_index(x, flatten_idx=True)
        torch.cuda.synchronize()
        res = {}
        for name, evict in (("back to back", None), ("after a 64 MB pass", mid), ("after a 512 MB pass", big)):
            ts = []
            for _ in range(30):
                if evict is not None:
                    evict.add_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); rvq.forward_index(x, flatten_idx=True); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[name] = float(np.median(ts))
        print(f"rows {n:4d}: " + "   ".join(f"{k}: {v:.1f} us" for k, v in res.items()), flush=True)

if __name__ == "__main__":
    main()
