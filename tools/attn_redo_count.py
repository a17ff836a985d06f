"""How often does the pipelined attention kernel repeat its pass (overflow of the optimistic first pass) in the real sampling
loop?  One bench-like request (SD1.5 512^2, CFG, N steps) (the counter is always on since round 6)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
sys.argv = [sys.argv[0]]
import bench
from gyre_amd import _lib, config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline
from gyre_amd.text import ClipTextEncoder, empty_prompt_ids, synthetic_prompt_ids
dev = torch.device("cuda", 0)
L = _lib.lib()
print("counter before:", L.gyre_debug_attn_redo_count())
unet = GyreHipUNet(gcfg.sd15_unet()).to(torch.bfloat16).to(dev); bench.fill_synthetic_on_device(unet, 0)
vae = GyreHipVAE(gcfg.sd15_vae()).to(torch.bfloat16).to(dev); bench.fill_synthetic_on_device(vae, 1)
clip = ClipTextEncoder.synthetic(dev, torch.bfloat16, seed=2)
pipe = GyrePipeline(unet, vae, clip, device=dev)
B = 8
ids, neg = synthetic_prompt_ids(B, seed=1234).to(dev), empty_prompt_ids(B).to(dev)
for steps in (10, 50):
    c0 = L.gyre_debug_attn_redo_count()
    pipe(seeds=[420420420 + j for j in range(B)], height=512, width=512, num_inference_steps=steps, guidance_scale=7.5, sampler="dpmpp_2m",
         output_type="latent", input_ids=ids, negative_ids=neg)
    torch.cuda.synchronize()
    c1 = L.gyre_debug_attn_redo_count()
    print(f"{steps} steps: {c1 - c0} workgroup redos over {pipe.last_unet_evals} UNet evaluations "
          f"(64x64 self-attention: 4096 workgroups per launch, 5 launches per evaluation)")
