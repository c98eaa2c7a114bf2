"""Launch time of the residual-VQ search by rows per workgroup ("rvq_rows" 0 / 2 / 4) and row count.  HIP events over 200 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiodec_amd import layers, native

def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    embeds = [torch.randn(64, 1024, generator=g) * (0.8 ** i) for i in range(8)]
    rvq = layers.ResidualVQ(embeds, device=dev)
    for n in (64, 128, 192, 256, 512, 1024):
        x = torch.randn(1, n, 64, generator=g).to(dev)
        line = [f"rows {n:5d}"]
        for r in (0, 2, 4):
            native.set_option("rvq_rows", r); native.set_option("rvq_v4_min", 1)
            for _ in range(20):
                rvq.forward_index(x, flatten_idx=True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                rvq.forward_index(x, flatten_idx=True)
            b.record(); torch.cuda.synchronize()
            line.append(f"rows/wg {r or 1}: {a.elapsed_time(b) * 5:.1f} us")
        print("  ".join(line), flush=True)

if __name__ == "__main__":
    main()
