"""Build-container probe: the REFERENCE's own request path over the native engine class, on CPU with oracle models.

  level 1  gyre/pipeline/pipeline_wrapper.py DiffusionPipelineWrapper (sampler injection, generators from seeds, kwargs
           filtering, progress / cancellation) -> gyre_amd.engine.GyreUnifiedPipeline
  level 2  gyre/services/generate.py GenerationServiceServicer.Generate (Stability-API protobuf Request -> stream of
           Answer artifacts) over a stand-in manager that hands out that wrapper

Absent third-party roots are stubbed as in make_golden.py; PNG encoding (cv2 / torchvision in the reference) is swapped
for PIL.  Prints one JSON object.  Run by tests/test_reference_service_path.py; nothing here ships."""
import contextlib
import io
import json
import os
import sys
import threading
from types import SimpleNamespace

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class AttrDict(dict):
    __getattr__ = dict.get


def build_pipeline():
    from transformers import CLIPTextConfig, CLIPTextModel
    from gyre_amd import config as gcfg, weights
    from gyre_amd.engine import GyreUnifiedPipeline
    from gyre_amd.modules import DiagonalGaussian
    from oracle import models_ref as M

    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))

    class UNet(torch.nn.Module):                      # oracle arithmetic behind the module surface the manager clones
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
            self.config = ucfg

        def forward(self, latents, t, encoder_hidden_states=None, **_):
            t = torch.as_tensor(t)
            if t.ndim == 0:
                t = t.expand(latents.shape[0])
            return SimpleNamespace(sample=M.unet_forward(usd, ucfg, latents, t, encoder_hidden_states))

    class VAE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
            self.config = vcfg

        def decode(self, z):
            return SimpleNamespace(sample=M.vae_decode(vsd, vcfg, z))

        def encode(self, x):
            return SimpleNamespace(latent_dist=DiagonalGaussian(M.vae_encode_moments(vsd, vcfg, x)))

        def enable_tiling(self): pass
        def disable_tiling(self): pass

    torch.manual_seed(0)
    te = CLIPTextModel(CLIPTextConfig(vocab_size=49408, hidden_size=ucfg.cross_attention_dim, intermediate_size=128,
                                      num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
                                      bos_token_id=49406, eos_token_id=49407, pad_token_id=49407)).eval()

    def tokenizer(text, add_special_tokens=False):    # no vocabulary ships offline: a deterministic word hash
        return {"input_ids": [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 40000) for w in text.split()]}

    sched = SimpleNamespace(config=AttrDict(prediction_type="epsilon", steps_offset=1, beta_start=0.00085, beta_end=0.012,
                                            beta_schedule="scaled_linear", num_train_timesteps=1000))
    from transformers import CLIPConfig, CLIPModel
    clip = CLIPModel(CLIPConfig(text_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                 vocab_size=49408, max_position_embeddings=77, bos_token_id=49406,
                                                 eos_token_id=49407),
                                vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                                   num_attention_heads=2, image_size=32, patch_size=8),
                                projection_dim=16)).eval()
    for p_ in clip.parameters():
        p_.requires_grad_(False)
    fe = SimpleNamespace(image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711], size=32)
    return GyreUnifiedPipeline(vae=VAE(), text_encoder=te, tokenizer=tokenizer, unet=UNet(), scheduler=sched,
                               clip_model=clip, feature_extractor=fe)


def main():
    import make_golden as mg
    mg.STUB_ROOTS.update({"huggingface_hub", "omegaconf", "pytorch_lightning", "ldm"} - set(sys.modules))
    mg._install()
    import generation_pb2
    from gyre import manager as ref_manager
    from gyre.pipeline.pipeline_wrapper import DiffusionPipelineWrapper
    out = {}

    pipeline = build_pipeline()
    mode = ref_manager.EngineMode(vram_optimisation_level=0, enable_cuda=False)
    wrapper = DiffusionPipelineWrapper(id="native-test", mode=mode, pipeline=pipeline)
    out["samplers"] = len(wrapper.get_samplers())
    wrapper.activate("cpu")
    progress = []
    kw = dict(prompt=["a photo of a cat", "a (red:1.3) house"], negative_prompt=["blurry", "blurry"], seed=[420420420, 420420421],
              height=128, width=128, num_inference_steps=4, guidance_scale=7.5,
              progress_callback=lambda **k: progress.append(k), suppress_output=True)
    images, nsfw = wrapper(sampler=generation_pb2.SAMPLER_K_DPMPP_2M, **kw)
    out["l1_shape"] = list(images.shape)
    out["l1_range_ok"] = bool(images.min() >= 0 and images.max() <= 1 and torch.isfinite(images).all())
    out["l1_nsfw"] = nsfw
    out["l1_sampler_seen"] = __import__("gyre_amd.engine", fromlist=["x"]).sampler_name(pipeline.scheduler)
    # same request straight through the host pipeline: identical tensors
    from gyre_amd.pipeline import GyrePipeline
    cond, unc = pipeline._embed(kw["prompt"], kw["negative_prompt"], 2, 1, True, 3)
    direct = GyrePipeline(pipeline.unet, pipeline.vae, device="cpu")(seeds=kw["seed"], text_embeddings=cond, uncond_embeddings=unc,
                                                                     height=128, width=128, num_inference_steps=4, sampler="dpmpp_2m")
    out["l1_equals_direct"] = bool(torch.equal(images, direct.float().cpu()))
    # another sampler family + img2img through the same wrapper
    img = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(1))
    # (the stubbed diffusers scheduler objects are falsy, which _inject_scheduler rejects: hand over a named stand-in the
    #  way a caller passes `scheduler=`; with real diffusers the enum lookup above yields the same class name)
    DDIMScheduler = type("DDIMScheduler", (), {"config": {}})
    images2, _ = wrapper(scheduler=DDIMScheduler(), image=img, strength=0.6, **kw)
    out["l1_ddim_img2img"] = [list(images2.shape), __import__("gyre_amd.engine", fromlist=["x"]).sampler_name(pipeline.scheduler)]
    # cancellation: the stop event aborts between UNet calls and the wrapper returns None (pipeline_wrapper.py:390-393)
    ev = threading.Event()
    ev.set()
    out["l1_cancelled"] = wrapper(sampler=generation_pb2.SAMPLER_K_EULER, stop_event=ev, **kw) is None
    # CLIP guidance through the reference's wrapper: the keyword reaches the native engine and changes the images
    plain, _ = wrapper(sampler=generation_pb2.SAMPLER_K_EULER, **kw)
    guided, _ = wrapper(sampler=generation_pb2.SAMPLER_K_EULER, clip_guidance_scale=0.3, **kw)
    out["l1_clip_guidance"] = [list(guided.shape), float((guided - plain).abs().max()) > 1e-3, bool(torch.isfinite(guided).all())]
    # tiling (circular convolution padding, unified_pipeline.py:1671-1712) is served; together with CLIP guidance it is not:
    # unsupported combination -> NotImplementedError (mapped to gRPC UNIMPLEMENTED by the reference)
    tiled, _ = wrapper(sampler=generation_pb2.SAMPLER_K_EULER, tiling=True, **kw)
    out["l1_tiling"] = [list(tiled.shape), bool(torch.isfinite(tiled).all())]     # (its effect on the images: tests/test_gpu_engine.py)
    try:
        wrapper(sampler=generation_pb2.SAMPLER_K_EULER, tiling=True, clip_guidance_scale=0.3, **kw)
        out["l1_unsupported"] = "no error"
    except NotImplementedError:
        out["l1_unsupported"] = "NotImplementedError"

    # ---- level 2: the gRPC servicer ------------------------------------------------------------------------------------
    try:
        from gyre import images as ref_images
        from gyre.services import generate as ref_generate

        def to_png(tensor):
            from PIL import Image
            arr = (tensor.clamp(0, 1)[0].permute(1, 2, 0).numpy() * 255).round().astype("uint8")
            buf = io.BytesIO()
            Image.fromarray(arr).save(buf, format="PNG")
            return buf.getvalue()
        ref_images.toPngBytes = lambda tensor: [to_png(tensor[i:i + 1]) for i in range(tensor.shape[0])]
        ref_generate.get_best_match = lambda accept, available: "image/png"      # accept_types is an absent third-party root

        class Manager:
            batchMode = SimpleNamespace(batchmax=lambda pixels: 4)
            _ram_monitor = None

            @contextlib.contextmanager
            def with_engine(self, engine_id=None, task=None):
                yield wrapper

            def _find_spec(self, **kw):
                return SimpleNamespace(id="native-test", task="generate")

        servicer = ref_generate.GenerationServiceServicer(Manager(), tensor_cache=None, resource_provider=None)
        req = generation_pb2.Request(engine_id="native-test", request_id="r1")
        req.prompt.add(text="a photo of a cat")
        req.image.height, req.image.width, req.image.samples, req.image.steps = 128, 128, 2, 3
        req.image.seed.extend([420420420, 420420421])
        req.image.transform.diffusion = generation_pb2.SAMPLER_K_EULER_ANCESTRAL

        class Ctx:
            def add_callback(self, cb): return True
            def invocation_metadata(self): return []
            def set_code(self, code): self.code = code
            def set_details(self, d): self.details = d
            def abort(self, code, details): raise RuntimeError(f"{code}: {details}")
        answers = list(servicer.Generate(req, Ctx()))
        arts = [a for ans in answers for a in ans.artifacts]
        out["l2_artifacts"] = len(arts)
        out["l2_types"] = sorted({int(a.type) for a in arts})
        out["l2_png"] = all(a.binary[:8] == b"\x89PNG\r\n\x1a\n" for a in arts if a.type == generation_pb2.ARTIFACT_IMAGE)
        out["l2_seeds"] = [int(a.seed) for a in arts if a.type == generation_pb2.ARTIFACT_IMAGE]
    except Exception as e:  # noqa: BLE001 - level 2 leans on more of the reference's stubbed surroundings
        import traceback
        out["l2_error"] = (type(e).__name__ + ": " + str(e))[:400]
        out["l2_trace"] = traceback.format_exc()[-1500:]
    print("PROBE_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
