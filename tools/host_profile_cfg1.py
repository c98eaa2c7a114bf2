import os, sys, tempfile, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_bench as cb
dev = "cuda:0"
with tempfile.TemporaryDirectory() as root, torch.no_grad():
    ad = cb.load(root, "libritts_sym", dev, 1, 80)
    x = cb.audio(dev, 1, 24000)
    f = lambda: ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x))))
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host {1e3*(t1-t0)/50:.3f} ms/iter, total {1e3*(t2-t0)/50:.3f} ms/iter")
    for name, g in (("encode", lambda: ad.tx_encoder.encode(x)),):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): z = g()
        torch.cuda.synchronize(); print(name, f"{1e3*(time.perf_counter()-t0)/50:.3f} ms")
    z = ad.tx_encoder.encode(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): idx = ad.tx_encoder.quantize(z)
    torch.cuda.synchronize(); print("quantize", f"{1e3*(time.perf_counter()-t0)/50:.3f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): zq = ad.rx_encoder.lookup(idx)
    torch.cuda.synchronize(); print("lookup", f"{1e3*(time.perf_counter()-t0)/50:.3f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): y = ad.decoder.decode(zq)
    torch.cuda.synchronize(); print("decode", f"{1e3*(time.perf_counter()-t0)/50:.3f} ms")
