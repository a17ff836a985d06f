"""Two host threads, each launching the pipelined attention kernel on its own stream over and over (64x64 self-attention of SD1.5 at
batch 2): is every result bit-equal to the serial one?  ATTN=7 selects the always-checked pass."""
import ctypes as C, math, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr())
B, H, N, D = 2, int(os.environ.get("H", "8")), int(os.environ.get("N", "4096")), int(os.environ.get("D", "40"))
NK = int(os.environ.get("NK", "0")) or N
Cc = H * D
VAR = int(os.environ.get("ATTN", "0"))
def mk(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    q = torch.randn(B, N, Cc, device=DEV, generator=g).to(torch.bfloat16)
    if os.environ.get("INTERLEAVED") == "1" and NK == N:           # the UNet's layout: Q | K in one [B, N, 2C] tensor
        qk = torch.empty(B, N, 2 * Cc, dtype=torch.bfloat16, device=DEV)
        qk[..., :Cc] = q
        qk[..., Cc:] = (torch.randn(B, N, Cc, device=DEV, generator=g) * (1.4426950408889634 / math.sqrt(D))).to(torch.bfloat16)
        vt_ = torch.randn(B, Cc, N, device=DEV, generator=g).to(torch.bfloat16)
        return qk, None, vt_
    k = (torch.randn(B, NK, Cc, device=DEV, generator=g) * (1.4426950408889634 / math.sqrt(D))).to(torch.bfloat16)
    LDV = (NK + 7) // 8 * 8
    vt = torch.zeros(B, Cc, LDV, dtype=torch.bfloat16, device=DEV); vt[:, :, :NK] = torch.randn(B, Cc, NK, device=DEV, generator=g).to(torch.bfloat16)
    return q, k, vt
data = [mk(1), mk(2)]
def run(k, stream):
    q, kk, vt = data[k]
    o = torch.empty(B, N, Cc, dtype=torch.bfloat16, device=DEV)
    L.gyre_debug_force_attn_variant(VAR)
    if kk is None:
        rc = L.gyre_op_attention_ex(C.c_void_p(stream.cuda_stream), vp(q), 2 * Cc, C.c_void_p(q.data_ptr() + 2 * Cc), 2 * Cc, vp(vt), vt.shape[2], B, H, N, NK, D, vp(o), Cc, 1)
        assert rc == 0
        return o
    rc = L.gyre_op_attention_ex(C.c_void_p(stream.cuda_stream), vp(q), Cc, vp(kk), Cc, vp(vt), vt.shape[2], B, H, N, NK, D, vp(o), Cc, 1)
    assert rc == 0
    return o
ref = []
for k in range(2):
    s = torch.cuda.Stream()
    o = run(k, s); s.synchronize(); ref.append(o)
ITERS = int(os.environ.get("ITERS", "300"))
bad = [0, 0]
def worker(k):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(ITERS):
            o = run(k, s)
            s.synchronize()
            if not torch.equal(o, ref[k]):
                bad[k] += 1
ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"ATTN={VAR} N={N} NK={NK} D={D} H={H}: mismatching launches per thread {bad} of {ITERS}")
