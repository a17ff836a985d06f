"""Dev experiment: does replaying the UNet forward from a HIP graph shrink the inter-kernel gaps?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in net.named_parameters():
        if p.ndim > 1:
            p.copy_(torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
        elif k.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
net._invalidate()
x = torch.randn(B, 4, 64, 64, device=dev)
t = torch.full((B,), 500, device=dev)
ctx = torch.randn(B, 77, 768, device=dev)
ref = net(x, t, encoder_hidden_states=ctx).sample
torch.cuda.synchronize()


def timeit(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


eager = timeit(lambda: net(x, t, encoder_hidden_states=ctx).sample)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        net(x, t, encoder_hidden_states=ctx)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = net(x, t, encoder_hidden_states=ctx).sample
graph.replay(); torch.cuda.synchronize()
print("graph output equal:", torch.equal(out, ref))
gt = timeit(graph.replay)
print(f"B={B}: eager {eager:.3f} ms, graph {gt:.3f} ms  ({(eager/gt-1)*100:+.1f}%)")
