"""Dev tool: time the attention-backward kernels (gyre_op_attention_bwd) at the SD1.5 self-attention shapes, batch 8."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gyre_amd import _lib

L = _lib.lib()
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, N, D) in ((8, 8, 4096, 40), (8, 8, 1024, 80), (8, 8, 256, 160), (32, 1, 784, 512)):
    Cc = H * D
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda: torch.randn(B, N, Cc, device=dev, generator=g).to(torch.bfloat16)
    q, k, v, o, do = mk(), mk() * 0.2, mk(), mk(), mk()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    need = L.gyre_op_attention_bwd_workspace(B, H, N, N, D)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    call = lambda: L.gyre_op_attention_bwd(st, p(q), Cc, p(k), Cc, p(v), Cc, p(o), Cc, p(do), Cc, B, H, N, N, D, 1, p(ws), need,
                                           p(dq), Cc, p(dk), Cc, p(dv), Cc)
    for _ in range(2):
        assert call() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    fl = 5 * 2.0 * B * H * N * N * D          # S, dP, dQ, dK, dV (the recomputation of S / dP in the second kernel not counted)
    print(f"attn bwd B={B} H={H} N={N} D={D}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s (5 GEMMs, incl. 3 transposes)")
