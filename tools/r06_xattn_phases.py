"""Cycle stamps of the fused cross-attention kernel's phase boundaries (one SD1.5 UNet call at batch 16: 5 launches of 512 workgroups)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from gyre_amd import _lib, config as gcfg
from gyre_amd.modules import GyreHipUNet
import bench
dev = "cuda:0"; L = _lib.lib()
net = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); bench.fill_synthetic_on_device(net, 0)
B = 16
x = torch.randn(B, 4, 64, 64, device=dev); t = torch.full((B,), 500, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
for _ in range(2): net(x, t, encoder_hidden_states=ctx)
buf = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
L.gyre_debug_xattn_stamps(C.c_void_p(buf.data_ptr()))
net(x, t, encoder_hidden_states=ctx); torch.cuda.synchronize()
L.gyre_debug_xattn_stamps(None)
s = buf.view(512, 8).cpu().double()
t0 = s[:, 0].min()
names = ["x block issued", "ln stats read", "to_q K loop", "Q -> XO + barrier", "attention (8 heads)", "to_out K loop", "epilogue -> XO + copy out", "row statistics"]
d = s[:, 1:] - s[:, :-1]
print("per-workgroup cycles between stamps (median over 512 workgroups of the LAST launch; 100 MHz counter? see total):")
for i, n in enumerate(names[1:]):
    print(f"  {n:28s} {float(d[:, i].median()):10.0f}")
print(f"  workgroup total            {float((s[:, 7] - s[:, 0]).median()):10.0f}")
print(f"  first start .. last end    {float(s[:, 7].max() - t0):10.0f}   starts spread {float(s[:, 0].max() - t0):10.0f}")

