"""Minimal end-to-end use of the native path without Gyre: text ids -> CLIP (host PyTorch) -> native UNet loop -> native VAE
-> PNG.  With --model it loads a diffusers-layout folder (model_index.json, unet/, vae/, text_encoder/, tokenizer/);
without it, random-init SD1.5-shaped weights are used (the output is noise-like, but every kernel runs at full size).

    python examples/txt2img.py --out /tmp/out.png --steps 20 --sampler dpmpp_2m --seeds 1 2
    python examples/txt2img.py --model /models/stable-diffusion-v1-5 --prompt "a (red:1.3) fox" --out fox.png
    python examples/txt2img.py --clip-guidance 0.3 --tome 1024        # CLIP-guided (native reverse sweeps) + token merging
    python examples/txt2img.py --model ... --clip-model /models/clip-vit-base-patch32 --clip-guidance 0.3
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gyre_amd import config as gcfg
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline
from gyre_amd.text import ClipTextEncoder, LPWTextEmbedder, empty_prompt_ids, synthetic_prompt_ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None, help="diffusers-layout model folder (optional)")
    ap.add_argument("--prompt", default="a photograph of an astronaut riding a horse")
    ap.add_argument("--negative", default="")
    ap.add_argument("--seeds", type=int, nargs="+", default=[420420420])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--sampler", default="dpmpp_2m")
    ap.add_argument("--size", type=int, nargs=2, default=[512, 512], metavar=("H", "W"))
    ap.add_argument("--cfg", type=float, default=7.5)
    ap.add_argument("--out", default="out.png")
    ap.add_argument("--clip-guidance", type=float, default=None, help="clip_guidance_scale (reference ClipGuidedMode)")
    ap.add_argument("--clip-model", default=None, help="transformers CLIPModel folder (default: random-init ViT-B/32)")
    ap.add_argument("--tome", type=int, default=0, help="ToMe: keys / values merged per self-attention")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = len(args.seeds)
    if args.model:
        pipe = GyrePipeline.from_pretrained(args.model, torch_dtype=torch.bfloat16, device=dev)
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(os.path.join(args.model, "tokenizer"))
        lpw = LPWTextEmbedder(pipe.text_encoder, lambda s: tok(s, add_special_tokens=False).input_ids)
        text, uncond = lpw.get_embeddings([args.prompt] * B, [args.negative] * B)
        kw = dict(text_embeddings=text, uncond_embeddings=uncond)
    else:
        unet = GyreHipUNet(gcfg.sd15_unet()).load_synthetic(0).to(torch.bfloat16).to(dev)
        vae = GyreHipVAE(gcfg.sd15_vae()).load_synthetic(1).to(torch.bfloat16).to(dev)
        pipe = GyrePipeline(unet, vae, ClipTextEncoder.synthetic(dev, torch.bfloat16, seed=2), device=dev)
        kw = dict(input_ids=synthetic_prompt_ids(B), negative_ids=empty_prompt_ids(B))
    if args.tome:
        pipe.unet.set_tome(args.tome)
    if args.clip_guidance:
        from types import SimpleNamespace
        from transformers import CLIPConfig, CLIPModel
        from gyre_amd.clipguided import patch_embedding_as_matmul
        clip = CLIPModel.from_pretrained(args.clip_model) if args.clip_model else CLIPModel(CLIPConfig(projection_dim=512))
        pipe.clip_model = patch_embedding_as_matmul(clip.eval().to(dev).requires_grad_(False))
        pipe.feature_extractor = SimpleNamespace(image_mean=[0.48145466, 0.4578275, 0.40821073],
                                                 image_std=[0.26862954, 0.26130258, 0.27577711], size={"shortest_edge": 224})
        if args.model:
            ids = tok([args.prompt] * B, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        else:
            ids = synthetic_prompt_ids(B, seed=7)
        kw.update(clip_guidance_scale=args.clip_guidance, clip_input_ids=ids)
    t0 = time.time()
    img = pipe(seeds=args.seeds, height=args.size[0], width=args.size[1], num_inference_steps=args.steps,
               guidance_scale=args.cfg, sampler=args.sampler, **kw)
    torch.cuda.synchronize()
    print(f"{B} image(s) {args.size[0]}x{args.size[1]}, {args.steps} steps {args.sampler}: {time.time() - t0:.2f} s "
          f"({pipe.last_unet_evals} UNet evaluations)")
    from PIL import Image
    arr = (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    for i, a in enumerate(arr):
        path = args.out if B == 1 else args.out.replace(".png", f"_{i}.png")
        Image.fromarray(a).save(path)
        print("wrote", path)


if __name__ == "__main__":
    main()
