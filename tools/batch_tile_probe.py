"""Dev probe: does running a chain of 64x64-level layers on two batch halves (each half's tensors fit the Infinity Cache between
producer and consumer) beat running every layer on the whole batch?  Chain = LN -> linear 320->960 -> linear 320->320 (+res) -> LN
-> GEGLU linear 320->2560 -> linear 1280->320 (+res), M = 16 x 4096 rows."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gyre_amd import _lib
from gpu_util import repack_linear

L = _lib.lib()
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
M, Cc = 16 * 4096, 320
p = lambda t: C.c_void_p(t.data_ptr())
h = torch.randn(M, Cc, generator=g).to(torch.bfloat16).to(dev)
w_qkv = repack_linear(torch.randn(960, 320, generator=g) / 18)
w_o = repack_linear(torch.randn(320, 320, generator=g) / 18)
w_ff1 = repack_linear(torch.randn(2560, 320, generator=g) / 18, geglu=True)
w_ff2 = repack_linear(torch.randn(320, 1280, generator=g) / 36)
gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
n1, qkv, h2, n2, ff, h3 = (torch.empty(M, c, dtype=torch.bfloat16, device=dev) for c in (320, 960, 320, 320, 1280, 320))
fl_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
fl_b = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def chain(r0, rows):
    s = lambda t: C.c_void_p(t.data_ptr() + r0 * t.shape[1] * 2)
    L.gyre_op_layernorm(st, s(h), rows, 320, p(gam), p(bet), 1e-5, s(n1))
    L.gyre_op_linear(st, s(n1), rows, 320, p(w_qkv), 960, None, None, 0, s(qkv))
    L.gyre_op_linear(st, s(n1), rows, 320, p(w_o), 320, None, s(h), 0, s(h2))          # stands in for attention + to_out
    L.gyre_op_layernorm(st, s(h2), rows, 320, p(gam), p(bet), 1e-5, s(n2))
    L.gyre_op_linear(st, s(n2), rows, 320, p(w_ff1), 1280, None, None, 1, s(ff))
    L.gyre_op_linear(st, s(ff), rows, 1280, p(w_ff2), 320, None, s(h2), 0, s(h3))


def timed(parts):
    ts = []
    for _ in range(7):
        fl_a.copy_(fl_b)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(parts):
            chain(i * (M // parts), M // parts)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


for parts in (1, 2, 4, 1, 2, 4):
    print(f"{parts} batch part(s): {timed(parts):8.1f} us for the chain over all {M} rows")
