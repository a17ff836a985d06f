"""SURVEY.md 8(f2) on the HIP path (-m gpu): the engine class an engines.yaml names (gyre_amd.engine.GyreUnifiedPipeline)
over the NATIVE GyreHipUNet / GyreHipVAE, called with exactly the keyword set the reference's
``DiffusionPipelineWrapper.__call__`` hands to its pipeline (gyre/pipeline/pipeline_wrapper.py:350-386), with the
scheduler / progress-bar objects the wrapper injects (:255-267, :26-47), and checked for

  * the return contract of ``UnifiedPipeline.__call__`` with output_type="tensor", return_dict=False
    (unified_pipeline.py:2512-2534): ``(images [B,3,H,W] float CPU tensor in 0..1, [nsfw flag per image])``
  * images equal to the fp32 CPU oracle pipeline on the same embeddings / seeds: PSNR >= 30 dB (bf16 vs fp32)
  * the safety checker being RUN when one is loaded (unified_pipeline.py:2514-2523), flags and blacked-out images returned
  * cancellation between UNet calls through the injected progress bar (ProgressBarAbort, pipeline_wrapper.py:26-47)

/root/reference does not exist on the GPU box, so the wrapper's side is restated here as data (the kwargs dict) - the same
wrapper is EXECUTED over this engine class on the CPU in tests/test_reference_service_path.py."""
import functools
from types import SimpleNamespace

import pytest
import torch

from gyre_amd import config as gcfg, weights
from gyre_amd.engine import GyreUnifiedPipeline
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gpu_util import DEV
from oracle import pipeline_ref as PR

pytestmark = pytest.mark.gpu


def sample_dpmpp_2m(*a, **k):            # the wrapper injects functools.partial(<k-diffusion sampler fn>, ...) (samplers.py:58-66)
    raise AssertionError("the engine maps the injected callable to its native sampler by NAME; it is never called")


def sample_euler_ancestral(*a, **k):
    raise AssertionError("never called")


def tokenizer(text, add_special_tokens=False):     # no CLIP vocabulary offline: a deterministic word hash
    return {"input_ids": [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 40000) for w in text.split()]}


def wrapper_kwargs(**over):
    """pipeline_args of DiffusionPipelineWrapper.__call__ (pipeline_wrapper.py:350-386), defaults of its signature (:288-341)."""
    kw = dict(prompt="", negative_prompt=None, num_images_per_prompt=1, generator=None, width=512, height=512,
              guidance_scale=7.5, cfg_execution="parallel", clip_guidance_scale=None, clip_guidance_base=None,
              prediction_type="epsilon", eta=None, churn=None, churn_tmin=None, churn_tmax=None, sigma_min=None, sigma_max=None,
              karras_rho=None, scheduler_noise_type="normal", num_inference_steps=50, image=None, mask_image=None,
              outmask_image=None, depth_map=None, hint_images=None, strength=None, lora=None, token_embeddings=None,
              hires_fix=None, hires_oos_fraction=None, tiling=False, debug_latent_tags=None, debug_latent_prefix="",
              output_type="tensor", return_dict=False)
    kw.update(over)
    return kw


def generators(seeds):                   # _build_generator (pipeline_wrapper.py:243-253): one generator per image on the mode device
    return [torch.Generator("cpu").manual_seed(s) for s in seeds]


class ProgressBar:
    """What the wrapper injects as pipeline.progress_bar: called with the iterable or total=, polls a stop event on update."""

    class Abort(BaseException):
        pass

    def __init__(self, stop_after=None):
        self.updates, self.stop_after = 0, stop_after

    def __call__(self, iterable=None, total=None):
        return self

    def update(self, n=1):
        self.updates += n
        if self.stop_after is not None and self.updates >= self.stop_after:
            raise ProgressBar.Abort()


def build_engine(ucfg, vcfg, te_layers=2, **extra):
    from transformers import CLIPTextConfig, CLIPTextModel
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    unet, vae = GyreHipUNet(ucfg), GyreHipVAE(vcfg)
    unet.load_state_dict(usd)
    vae.load_state_dict(vsd)
    torch.manual_seed(0)
    te = CLIPTextModel(CLIPTextConfig(vocab_size=49408, hidden_size=ucfg.cross_attention_dim, intermediate_size=128,
                                      num_hidden_layers=te_layers, num_attention_heads=2, max_position_embeddings=77,
                                      bos_token_id=49406, eos_token_id=49407, pad_token_id=49407)).eval()
    eng = GyreUnifiedPipeline(vae=vae.to(DEV), text_encoder=te.to(DEV), tokenizer=tokenizer, unet=unet.to(DEV), **extra)
    return usd, vsd, eng


@pytest.fixture(scope="module")
def tiny_engine():
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    usd, vsd, eng = build_engine(ucfg, vcfg)
    return ucfg, vcfg, usd, vsd, eng


def oracle_images(eng, ucfg, vcfg, usd, vsd, prompt, negative, seeds, size, steps, sampler, **kw):
    cond, unc = eng._embed(prompt, negative, len(seeds), 1, True, 3)
    ref, evals = PR.generate_ref(usd, ucfg, vsd, vcfg, cond.float().cpu(), unc.float().cpu(), seeds, size, size, steps, 7.5,
                                 sampler, unet_sample_size=ucfg.sample_size, **kw)
    return ref


def test_engine_on_native_modules_matches_oracle_and_return_contract(tiny_engine):
    ucfg, vcfg, usd, vsd, eng = tiny_engine
    prompt, negative, seeds = ["a photo of a cat", "a (red:1.3) house"], ["blurry", "blurry"], [420420420, 420420421]
    eng.scheduler = functools.partial(sample_dpmpp_2m, warmup_lms=True, ddim_cutoff=0.1)
    bar = eng.progress_bar = ProgressBar()
    out = eng(**wrapper_kwargs(prompt=prompt, negative_prompt=negative, generator=generators(seeds), width=128, height=128,
                               num_inference_steps=8))
    assert isinstance(out, tuple) and len(out) == 2
    images, nsfw = out
    assert isinstance(images, torch.Tensor) and images.device.type == "cpu" and images.dtype == torch.float32
    assert images.shape == (2, 3, 128, 128) and float(images.min()) >= 0 and float(images.max()) <= 1
    assert nsfw == [False, False]
    assert bar.updates >= 8                                        # the injected progress bar saw every step
    ref = oracle_images(eng, ucfg, vcfg, usd, vsd, prompt, negative, seeds, 128, 8, "dpmpp_2m")
    p = PR.psnr(images, ref)
    print(f"[parity] engine (native UNet/VAE) dpmpp_2m 8 steps vs fp32 oracle: PSNR {p:.1f} dB")
    assert p >= 30.0
    # return_dict form (the diffusers-style caller)
    eng.progress_bar = None
    res = eng(**wrapper_kwargs(prompt=prompt, negative_prompt=negative, generator=generators(seeds), width=128, height=128,
                               num_inference_steps=8, return_dict=True))
    assert torch.equal(res.images, images) and res.nsfw_content_detected == [False, False]


def test_engine_img2img_euler_a_and_num_images_per_prompt(tiny_engine):
    ucfg, vcfg, usd, vsd, eng = tiny_engine
    eng.scheduler, eng.progress_bar = sample_euler_ancestral, None
    seeds = [7, 8]
    image = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(2))
    images, nsfw = eng(**wrapper_kwargs(prompt=["a lighthouse"], negative_prompt=None, num_images_per_prompt=2,
                                        generator=generators(seeds), width=128, height=128, num_inference_steps=8, image=image,
                                        strength=0.5))
    assert images.shape == (2, 3, 128, 128) and nsfw == [False, False]
    cond, unc = eng._embed(["a lighthouse"], None, 2, 2, True, 3)
    ref, _ = PR.generate_ref(usd, ucfg, vsd, vcfg, cond.float().cpu(), unc.float().cpu(), seeds, 128, 128, 8, 7.5, "euler_a",
                             image=image, strength=0.5, unet_sample_size=ucfg.sample_size)
    p = PR.psnr(images, ref)
    print(f"[parity] engine img2img euler_a (2 images per prompt) vs fp32 oracle: PSNR {p:.1f} dB")
    assert p >= 30.0


def test_engine_runs_the_safety_checker_and_returns_its_verdict(tiny_engine):
    ucfg, vcfg, usd, vsd, eng = tiny_engine
    seen = {}

    class FeatureExtractor:                              # CLIPImageProcessor surface: (PIL list, return_tensors="pt") -> .pixel_values
        def __call__(self, pil_images, return_tensors="pt"):
            import numpy as np
            seen["pil"] = [im.size for im in pil_images]
            px = torch.stack([torch.from_numpy(np.asarray(im.resize((32, 32)))).permute(2, 0, 1).float() / 255 for im in pil_images])
            ns = SimpleNamespace(pixel_values=px)
            ns.to = lambda dev: SimpleNamespace(pixel_values=px.to(dev))
            return ns

    class Checker(torch.nn.Module):                      # StableDiffusionSafetyChecker surface
        def forward(self, images, clip_input):
            seen["clip_input"] = (clip_input.device.type, clip_input.dtype, tuple(clip_input.shape))
            seen["images"] = (type(images).__name__, images.shape)
            images = images.copy()
            images[1] = 0.0                              # "black out" the second image
            return images, [False, True]

    eng.scheduler, eng.progress_bar = sample_euler_ancestral, None
    eng.safety_checker, eng.feature_extractor = Checker(), FeatureExtractor()
    try:
        kw = wrapper_kwargs(prompt=["a", "b"], generator=generators([1, 2]), width=128, height=128, num_inference_steps=3)
        images, nsfw = eng(**kw)
        assert nsfw == [False, True] and float(images[1].abs().max()) == 0.0 and float(images[0].max()) > 0
        assert seen["pil"] == [(128, 128)] * 2 and seen["images"] == ("ndarray", (2, 128, 128, 3))
        assert seen["clip_input"][0] == "cuda" and seen["clip_input"][1] == torch.float32
        # the reference's escape hatch: run_safety_checker=False (unified_pipeline.py:1753,2514)
        images2, nsfw2 = eng(**dict(kw, generator=generators([1, 2]), run_safety_checker=False))
        assert nsfw2 == [False, False] and float(images2[1].max()) > 0
        assert torch.equal(images2[0], images[0])
    finally:
        eng.safety_checker = eng.feature_extractor = None


def test_engine_cancellation_between_unet_calls(tiny_engine):
    ucfg, vcfg, usd, vsd, eng = tiny_engine
    eng.scheduler = sample_euler_ancestral
    eng.progress_bar = ProgressBar(stop_after=2)
    with pytest.raises(ProgressBar.Abort):                # the wrapper catches its ProgressBarAbort and returns None (:390-393)
        eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=6))
    # the handles stay usable after an aborted request
    eng.progress_bar = None
    images, _ = eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=3))
    assert bool(torch.isfinite(images).all())
    # scheduler_noise_type = "brownian" (common_scheduler.py:596-606): served (parity unpinned, torchsde absent) - an ancestral
    # sampler then reads ONE Brownian path per image: reproducible per seed, different from the normal-noise run
    bkw = lambda: wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=3,
                                 scheduler_noise_type="brownian")
    brown, _ = eng(**bkw())
    brown2, _ = eng(**bkw())
    assert bool(torch.isfinite(brown).all()) and torch.equal(brown, brown2) and not torch.equal(brown, images)
    with pytest.raises(ValueError):
        eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, scheduler_noise_type="pink"))
    # `latents`: declared and documented by the reference's __call__ (unified_pipeline.py:1749,1807-1810) but never read - same here
    kw = wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=3)
    kw["latents"] = torch.full((1, 4, 16, 16), 7.0)
    with pytest.warns(RuntimeWarning, match="latents"):            # ... but not silently (round 6)
        ignored, _ = eng(**kw)
    assert torch.equal(ignored, images)
    # tiling (unified_pipeline.py:1671-1712, :1845): a request option the native conv gather serves; the next request is plain again
    tiled, _ = eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=3, tiling=True))
    again, _ = eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=3))
    assert bool(torch.isfinite(tiled).all()) and not torch.equal(tiled, images) and torch.equal(again, images)
    with pytest.raises(ValueError):
        eng(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, tiling="diagonal"))


def test_engine_full_size_sd15_request_on_native_modules():
    """The real SD1.5 topology behind the engine class: one 512x512 image, 3 Euler-a steps, vs the fp32 oracle."""
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    usd, vsd, eng = build_engine(ucfg, vcfg, te_layers=1)
    eng.scheduler = sample_euler_ancestral
    seeds = [420420420]
    images, nsfw = eng(**wrapper_kwargs(prompt=["a photo of an astronaut"], negative_prompt=[""], generator=generators(seeds),
                                        num_inference_steps=3))
    assert images.shape == (1, 3, 512, 512) and nsfw == [False]
    ref = oracle_images(eng, ucfg, vcfg, usd, vsd, ["a photo of an astronaut"], [""], seeds, 512, 3, "euler_a")
    p = PR.psnr(images, ref)
    print(f"[parity] engine SD1.5 512x512 3-step euler_a vs fp32 oracle: PSNR {p:.1f} dB")
    assert p >= 30.0


def test_engine_routes_an_sdxl_unet_through_both_text_towers():
    """BASELINE configs[3] above the UNet: an engine whose UNet has addition_embed_type "text_time" conditions on two text
    towers (context = penultimate states of both, pooled text_embeds of the second, time_ids from the request size) - the
    published SDXL-base scheme (the reference has no SDXL: extension, parity unpinned).  Native engine vs the same host flow on
    the fp32 oracle models fed with the engine's own conditioning."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from test_host_pipeline import OracleUNet, OracleVAE
    from gyre_amd.pipeline import GyrePipeline
    ucfg = gcfg.tiny_sdxl_unet()                                   # context 64 = 24 + 40, pooled 32
    vcfg = gcfg.VAEConfig(block_out_channels=(32, 64, 64, 64), sample_size=64, scaling_factor=0.13025)
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    unet, vae = GyreHipUNet(ucfg), GyreHipVAE(vcfg)
    unet.load_state_dict(usd); vae.load_state_dict(vsd)
    torch.manual_seed(0)
    kw = dict(vocab_size=49408, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
              bos_token_id=49406, eos_token_id=49407, pad_token_id=49407)
    te1 = CLIPTextModel(CLIPTextConfig(hidden_size=24, **kw)).eval()
    te2 = CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=40, projection_dim=32, **kw)).eval()
    eng = GyreUnifiedPipeline(vae=vae.to(DEV), text_encoder=te1.to(DEV), tokenizer=tokenizer, unet=unet.to(DEV),
                              text_encoder_2=te2.to(DEV), tokenizer_2=tokenizer)
    eng.scheduler = functools.partial(sample_dpmpp_2m, warmup_lms=True, ddim_cutoff=0.1)
    prompt, negative, seeds = ["a photo of a cat", "a (red:1.3) house"], ["", "blurry"], [11, 12]
    images, nsfw = eng(**wrapper_kwargs(prompt=prompt, negative_prompt=negative, generator=generators(seeds), width=128, height=128,
                                        num_inference_steps=6, guidance_scale=5.0))
    assert images.shape == (2, 3, 128, 128) and bool(torch.isfinite(images).all()) and nsfw == [False, False]
    cond, unc, added, uadded = eng._embed_sdxl(prompt, negative, 2, 1, True, 3, 128, 128)
    assert cond.shape == (2, 77, 64) and added["text_embeds"].shape == (2, 32) and added["time_ids"].tolist() == [[128.0, 128.0, 0.0, 0.0, 128.0, 128.0]] * 2
    # explicit negative prompts - also the empty string - are ENCODED; no negative prompt at all conditions on zeros (published rule)
    assert float(unc[0].abs().max()) > 0 and float(unc[1].abs().max()) > 0
    _, unc_none, _, uadded_none = eng._embed_sdxl(prompt, None, 2, 1, True, 3, 128, 128)
    assert float(unc_none.abs().max()) == 0 and float(uadded_none["text_embeds"].abs().max()) == 0
    cpu = lambda d: {k: v.float().cpu() for k, v in d.items()}
    ref = GyrePipeline(OracleUNet(usd, ucfg), OracleVAE(vsd, vcfg), device="cpu")(
        seeds=seeds, text_embeddings=cond.float().cpu(), uncond_embeddings=unc.float().cpu(), height=128, width=128,
        num_inference_steps=6, sampler="dpmpp_2m", guidance_scale=5.0, added_cond=cpu(added), uncond_added_cond=cpu(uadded))
    p = PR.psnr(images, ref.float().cpu())
    print(f"[parity] tiny SDXL engine request: PSNR {p:.1f} dB")
    assert p >= 30.0
    # an SDXL UNet without the second tower is a configuration error, not a silent SD1.x request
    eng2 = GyreUnifiedPipeline(vae=vae, text_encoder=te1, tokenizer=tokenizer, unet=unet)
    eng2.scheduler = eng.scheduler
    with pytest.raises(ValueError):
        eng2(**wrapper_kwargs(prompt=["a"], generator=generators([1]), width=128, height=128, num_inference_steps=2))
