import sys, os, threading
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_gpu_threads as T
from gyre_amd import config as gcfg, _lib
cfg = gcfg.sd15_unet()
bits = int(os.environ.get("BITS", "0"), 0)
_lib.lib().gyre_debug_attn_redo_count()          # (the counter is always on since round 6)
nets = [T._unet(cfg, 0), T._unet(cfg, 1)]
ins = [[T._inputs(cfg, 2, 64, 10 * k + i) for i in range(2)] for k in range(2)]
def job(k):
    def f(i):
        _lib.lib().gyre_debug_gemm_ablation(bits)
        _lib.lib().gyre_debug_force_attn_variant(int(os.environ.get("ATTN", "0")))
        x, t, ctx = ins[k][i % 2]
        return nets[k](x, t, encoder_hidden_states=ctx).sample
    return f
for rep in range(int(os.environ.get("REPS", "3"))):
    serial, threaded = T._run_threads([job(0), job(1)], 6)
    bad = [(k, i, float((serial[k][i].float() - threaded[k][i].float()).abs().max())) for k in range(2) for i in range(6) if not torch.equal(serial[k][i], threaded[k][i])]
    print("bits", hex(bits), "attn", os.environ.get("ATTN", "0"), "rep", rep, "mismatches", bad[:6], flush=True)
print("attention redo count:", _lib.lib().gyre_debug_attn_redo_count())
