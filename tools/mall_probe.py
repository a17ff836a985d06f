"""Dev probe: does a tensor just WRITTEN by one kernel wait in the Infinity Cache / L2 for the next kernel's reads?
Times a streaming read (gyre_op_copy_probe) of an N-MB buffer: (a) again right after reading it (warm), (b) after a
cache-evicting 512 MB copy (cold), (c) right after another kernel wrote it."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gyre_amd import _lib

L = _lib.lib()
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fl_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
fl_b = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


for mb in (16, 42, 84, 126, 200):
    n = mb << 20
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    read = lambda: L.gyre_op_copy_probe(st, C.c_void_p(x.data_ptr()), C.c_void_p(dst.data_ptr()), n)
    write = lambda: L.gyre_op_copy_probe(st, C.c_void_p(src.data_ptr()), C.c_void_p(x.data_ptr()), n)
    res = {"warm": [], "cold": [], "after_write": []}
    for _ in range(5):
        read(); torch.cuda.synchronize()
        res["warm"].append(timed(read))
        fl_a.copy_(fl_b); torch.cuda.synchronize()
        res["cold"].append(timed(read))
        fl_a.copy_(fl_b); write(); torch.cuda.synchronize()
        res["after_write"].append(timed(read))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{mb:4d} MB read+copy: warm {med['warm']:7.1f} us  cold {med['cold']:7.1f} us  right after being written {med['after_write']:7.1f} us"
          f"   (GB/s of 2x bytes: {2 * n / med['warm'] / 1e3:.0f} / {2 * n / med['cold'] / 1e3:.0f} / {2 * n / med['after_write'] / 1e3:.0f})")
