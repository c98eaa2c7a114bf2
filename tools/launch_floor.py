#!/usr/bin/env python3
"""Floor of a dependent kernel chain on this box: N tiny launches on one stream (tuning aid)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audiodec_amd import native
from audiodec_amd.native import RingView
lib = native.lib()
dev = "cuda:0"
src = torch.zeros(1, 4, 4, device=dev); ring = torch.zeros(1, 8, 4, device=dev)
v = RingView(); v.base, v.rows, v.channels, v.cursor, v.ch_off = ring.data_ptr(), 8, 4, 0, 0
st = native.current_stream(dev)
def chain(n):
    for _ in range(n):
        lib.adk_ring_write(C.c_void_p(src.data_ptr()), v, None, None, 1, 4, st)
chain(100); torch.cuda.synchronize()
for n in (100, 1000):
    t0 = time.perf_counter(); chain(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n} tiny kernels: host {1e6*(t1-t0)/n:.2f} us/launch, total {1e6*(t2-t0)/n:.2f} us/launch")
# same through a captured HIP graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st2 = native.current_stream(dev)
    g.capture_begin()
    for _ in range(100):
        lib.adk_ring_write(C.c_void_p(src.data_ptr()), v, None, None, 1, 4, st2)
    g.capture_end()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graph of 100 tiny kernels: {1e6*(t2-t0)/1000:.2f} us/kernel")
