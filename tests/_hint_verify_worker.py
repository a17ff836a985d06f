"""Worker of tests/test_gpu_properties.py::test_hint_verifier (own process: GYRE_VERIFY_HINTS is read once per process)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gyre_amd import config as gcfg, weights  # noqa: E402
from gyre_amd.modules import GyreHipUNet, cfg_pairs  # noqa: E402

dev = "cuda:0"
ucfg = gcfg.tiny_unet()
unet = GyreHipUNet(ucfg)
unet.load_state_dict(weights.synthetic_state_dict(weights.unet_param_shapes(ucfg)))
unet = unet.to(dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 4, 16, 16, generator=g).to(dev)
ctx = torch.randn(4, 77, ucfg.cross_attention_dim, generator=g).to(dev)
pair = torch.cat([x, x])                               # what CFGUNet_Parallel builds: halves identical
with cfg_pairs():
    good = unet(pair, 10, encoder_hidden_states=ctx).sample        # scalar timestep -> uniform-timestep hint too
plain = unet(pair, torch.tensor([10, 10, 10, 10], device=dev), encoder_hidden_states=ctx).sample
assert torch.isfinite(good).all() and float((good - plain).abs().max()) < 0.1
wrong = torch.cat([x, x + 1e-3])                       # a caller that sets the hint on a batch that is NOT a pair batch
try:
    with cfg_pairs():
        unet(wrong, 10, encoder_hidden_states=ctx)
    print("NOT_CAUGHT")
except ValueError as e:
    print("CAUGHT:" + str(e)[:160])
# and the same call goes through once the hint is not given
unet(wrong, 10, encoder_hidden_states=ctx)
print("DONE")
