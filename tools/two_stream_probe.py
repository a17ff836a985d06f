"""Dev probe: one SD1.5 UNet evaluation of batch 16 as ONE call vs as P independent batch parts on P HIP streams (one handle
per part; the samples of a batch never interact, so the halves of a CFG batch are independent work).  Question: do the dispatch
gaps, the tile-quantisation tails and the latency-bound 16x16 / 8x8 levels of one part hide behind the other part's kernels?

Usage (GPU box): python tools/two_stream_probe.py [B=16] [parts=2,4]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
PARTS = [int(p) for p in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def make():
    net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    net._invalidate()
    return net


nets = [make() for _ in range(max(PARTS))]
x = torch.randn(B, 4, 64, 64, device=dev)
t = torch.full((B,), 500, device=dev)
ctx = torch.randn(B, 77, 768, device=dev)
streams = [torch.cuda.Stream() for _ in range(max(PARTS))]


def whole():
    nets[0](x, t, encoder_hidden_states=ctx)


def split(P):
    n = B // P
    xs = [x[i * n:(i + 1) * n].contiguous() for i in range(P)]
    ts = [t[i * n:(i + 1) * n].contiguous() for i in range(P)]
    cs = [ctx[i * n:(i + 1) * n].contiguous() for i in range(P)]

    def run():
        cur = torch.cuda.current_stream()
        for i in range(P):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                nets[i](xs[i], ts[i], encoder_hidden_states=cs[i])
        for i in range(P):
            cur.wait_stream(streams[i])
    return run


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f"one call, batch {B}: {timeit(whole):.2f} ms")
for P in PARTS:
    print(f"{P} parts of {B // P} on {P} streams: {timeit(split(P)):.2f} ms")
    n = B // P
    one = lambda: nets[0](x[:n].contiguous(), t[:n].contiguous(), encoder_hidden_states=ctx[:n].contiguous())
    print(f"   (one part alone: {timeit(one):.2f} ms)")
print(f"one call again: {timeit(whole):.2f} ms")
