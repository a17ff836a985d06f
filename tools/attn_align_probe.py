"""Does the pipelined attention kernel's time depend on where its operands sit?  16 x 8 heads x 4096^2 x D = 40 (the 64x64 self-attention
of SD1.5, Q|K interleaved with row stride 2C as the UNet lays them out), with the Q|K / V^T / output buffers placed at
different offsets inside one big allocation."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gyre_amd import _lib
from gpu_util import DEV, st
L = _lib.lib()
B, H, N, D = 16, 8, 4096, 40
C = H * D
g = torch.Generator(device=DEV).manual_seed(0)
qk_b, vt_b, o_b = B * N * 2 * C * 2, B * C * N * 2, B * N * C * 2
big = torch.empty(qk_b + vt_b + o_b + (64 << 20), dtype=torch.uint8, device=DEV)
base = big.data_ptr()
print(f"base % 2MB = {base % (2 << 20)}")
def place(off, nbytes, init):
    t = big[off: off + nbytes].view(torch.bfloat16)
    if init is not None:
        t.copy_((torch.randn(nbytes // 2, device=DEV, generator=g) * init).to(torch.bfloat16))
    return base + off
def run_at(o_qk, o_vt, o_o):
    q = place(o_qk, qk_b, 0.5); vt = place(o_vt, vt_b, 1.0); o = place(o_o, o_b, None)
    vpq, vpk = ctypes.c_void_p(q), ctypes.c_void_p(q + C * 2)
    run = lambda: _lib.check(L.gyre_op_attention_ex(st(), vpq, 2 * C, vpk, 2 * C, ctypes.c_void_p(vt), N, B, H, N, N, D, ctypes.c_void_p(o), C, 1))
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
A = 2 << 20
up = lambda v: (v + A - 1) // A * A
for rep in range(4):
    for name, (dq, dv, do) in {"all 2MB aligned": (0, 0, 0), "+256": (256, 256, 256), "+4K": (4096, 4096, 4096), "+64K+256": (65792, 65792, 65792),
                               "q+256 only": (256, 0, 0), "vt+256 only": (0, 256, 0), "o+256 only": (0, 0, 256),
                               "q+1MB": (1 << 20, 0, 0), "vt +1MB": (0, 1 << 20, 0), "packed tight": (-1, 0, 0)}.items():
        if name == "packed tight":
            o1, o2, o3 = 0, qk_b, qk_b + vt_b
        else:
            o1 = dq; o2 = up(o1 + qk_b) + dv; o3 = up(o2 + vt_b) + do
        print(f"{name:18s}: {run_at(o1, o2, o3):7.1f} us", flush=True)
