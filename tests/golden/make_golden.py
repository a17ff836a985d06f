"""Generate golden vectors by importing the REFERENCE's own modules.

Runs only in the build container (needs /root/reference, which does not travel to
the GPU box).  Writes small .npz fixtures next to this file; they are data
(inputs + the reference's outputs), not source.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Third-party roots the reference imports but which are absent here are replaced
by permissive stub modules through a sys.meta_path finder - the same trick the
reference uses itself (gyre/src/__init__.py:22-37).  Only torch-only members of
the reference are executed.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

STUB_ROOTS = {
    "diffusers", "torchvision", "k_diffusion", "cv2", "kornia", "easing_functions", "pynvml", "accept_types",
    "twisted", "colorama", "gdown", "hupper", "wsgicors", "ftfy", "stanza", "nltk", "torchsde", "torchdiffeq",
    "mmcv", "mmseg", "mmpose", "mmdet", "timm", "tome", "xformers", "picklemagic", "resize_right",
    "interp_methods", "basicsr", "zoedepth", "midas", "lora_diffusion", "tqdm_loggable",
}


class _Meta(type):
    """Metaclass so that attribute access on a stub CLASS yields more stub classes."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _make_stub(name)

    def __add__(cls, other):
        return ""

    def __radd__(cls, other):
        return ""

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


def _make_stub(name):
    class _Stub(metaclass=_Meta):
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return a[0] if (len(a) == 1 and callable(a[0]) and not k) else _make_stub("obj")()

        def __getattr__(self, n):
            if n.startswith("__") and n.endswith("__"):
                raise AttributeError(n)
            return _make_stub(n)()

        def __class_getitem__(cls, item):
            return cls

        def __iter__(self):
            return iter(())

        def __len__(self):
            return 0

    _Stub.__name__ = name
    return _Stub


class _Anything(types.ModuleType):
    """Module whose every attribute is a permissive class (usable as base class,
    decorator, callable, or namespace)."""
    __path__ = []  # behave as a package

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        st = _make_stub(name)
        setattr(self, name, st)
        return st


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in STUB_ROOTS or fullname.startswith("gyre.src."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Anything(spec.name)

    def exec_module(self, module):
        pass


def _install():
    # real transformers first (it probes torchvision itself; must not see the stub)
    import transformers  # noqa
    from transformers import CLIPTextModel, CLIPTokenizer, CLIPModel  # noqa
    try:
        transformers.CLIPFeatureExtractor = transformers.CLIPImageProcessor
    except Exception:
        transformers.CLIPFeatureExtractor = type("CLIPFeatureExtractor", (), {})
    import transformers.models.clip as _clip
    if not hasattr(_clip, "CLIPFeatureExtractor"):
        _clip.CLIPFeatureExtractor = transformers.CLIPFeatureExtractor
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "gyre", "generated"))  # generation_pb2 does `import tensors_pb2`
    try:
        import transformers.tokenization_utils as tu
        import transformers.tokenization_utils_base as tub
        if not hasattr(tu, "BatchEncoding"):
            tu.BatchEncoding = tub.BatchEncoding
    except Exception:
        pass


def gens(seeds):
    return [torch.Generator().manual_seed(s) for s in seeds]


def main():
    _install()
    out = {}

    # (1) per-image RNG -----------------------------------------------------
    from gyre.pipeline import randtools
    seeds = [420420420, 420420421]
    out["randn_2x4x8x8"] = randtools.batched_randn([2, 4, 8, 8], gens(seeds), "cpu", torch.float32).numpy()
    out["rand_2x4x8x8"] = randtools.batched_rand([2, 4, 8, 8], gens(seeds), "cpu", torch.float32).numpy()
    out["randn_seed420420420_1x4x64x64_head"] = randtools.batched_randn(
        [1, 4, 64, 64], gens([420420420]), "cpu", torch.float32).flatten()[:16].numpy()
    out["rng_seeds"] = np.array(seeds, dtype=np.int64)

    # (5,6) beta / sigma tables, sigma<->t ------------------------------------
    from gyre.pipeline.kschedulers.scheduling_utils import KSchedulerMixin
    from gyre.pipeline import common_scheduler as cs

    ks = object.__new__(cs.KDiffusionScheduler)
    ks.device = "cpu"
    betas = cs.KDiffusionScheduler.get_betas(ks)
    ac = cs.KDiffusionScheduler.get_alphas_cumprod(ks, cs.KDiffusionScheduler.get_alphas(ks, betas))
    out["betas"] = betas.numpy()
    out["alphas_cumprod"] = ac.numpy()

    mix = KSchedulerMixin()
    sig = ((1 - ac) / ac) ** 0.5
    mix.log_sigmas = sig.log()
    for n in (20, 50):
        t = torch.linspace(999, 0, n)
        s = mix.t_to_sigma(t)
        out[f"sigmas_n{n}"] = torch.cat([s, torch.zeros(1)]).numpy()
        out[f"sigma_to_t_n{n}"] = mix.sigma_to_t(s).numpy()
    out["sigma_min_max"] = np.array([float(sig[0]), float(sig[-1])], dtype=np.float64)

    # (2) DPM-Solver++(2M) with an analytic denoiser ---------------------------
    from gyre.pipeline.schedulers import sample_dpmpp_2m as s2m
    import tqdm as _tq
    s2m.trange = lambda n, disable=None: range(n)  # silence the progress bar
    for n in (20, 50):
        calls = []

        def toy(x, sigma):
            calls.append(float(sigma[0]))
            return x / (1 + sigma.view(-1, 1, 1, 1) ** 2)

        x0 = randtools.batched_randn([2, 4, 8, 8], gens(seeds), "cpu", torch.float32)
        sigmas = torch.from_numpy(out[f"sigmas_n{n}"])
        x = s2m.sample_dpmpp_2m(toy, x0 * sigmas[0], sigmas, warmup_lms=True, ddim_cutoff=0.1)
        out[f"dpmpp2m_n{n}_x"] = x.numpy()
        out[f"dpmpp2m_n{n}_evals"] = np.array(len(calls))
        out[f"dpmpp2m_n{n}_eval_sigmas"] = np.array(calls, dtype=np.float64)
        calls.clear()
        x = s2m.sample_dpmpp_2m(toy, x0 * sigmas[0], sigmas)  # plain 2M, no warm-up / cutoff
        out[f"dpmpp2m_plain_n{n}_x"] = x.numpy()

    # (3) CFG wrappers with a fake UNet ----------------------------------------
    from gyre.pipeline.unet import cfg as rcfg
    seen = {}

    def fake_f(latents, t):
        seen["shape"] = tuple(latents.shape)
        seen["t"] = t.clone()
        w = torch.arange(1, latents.shape[0] + 1, dtype=latents.dtype).view(-1, 1, 1, 1)
        return latents * w + t.view(-1, 1, 1, 1).to(latents.dtype) * 0.001

    children = rcfg.CFGChildUnets(g=lambda l, t: fake_f(l, t) * 2.0, u=lambda l, t: fake_f(l, t) * 0.5, f=fake_f)
    lat = randtools.batched_randn([2, 4, 8, 8], gens(seeds), "cpu", torch.float32)
    tt = torch.tensor([981, 981])
    out["cfg_in"] = lat.numpy()
    out["cfg_t"] = tt.numpy()
    out["cfg_parallel_out"] = rcfg.CFGUNet_Parallel(children, 7.5, 2)(lat, tt).numpy()
    out["cfg_parallel_call_shape"] = np.array(seen["shape"])
    out["cfg_parallel_call_t"] = seen["t"].numpy()
    out["cfg_sequential_out"] = rcfg.CFGUNet_Sequential(children, 7.5, 2)(lat, tt).numpy()

    # (4) UnetWithExtraChannels (runway inpaint 9-ch assembly) -------------------
    from gyre.pipeline.unet import core as rcore
    got = {}

    def rec(latents, t):
        got["x"] = latents
        return latents[:, :4]

    extra = torch.linspace(0, 1, 2 * 5 * 8 * 8).view(2, 5, 8, 8)
    rcore.UnetWithExtraChannels(rec, extra)(lat, tt)
    out["extra_channels_in"] = extra.numpy()
    out["extra_channels_cat"] = got["x"].numpy()

    # (7) Txt2imgMode.generateLatents ------------------------------------------
    from gyre.pipeline import unified_pipeline as up

    class _Sched:
        def prepare_initial_latents(self, l):
            return l * 14.5

    for name, (lh, lw) in {"512x512": (64, 64), "512x768": (64, 96), "256x256": (32, 32), "256x768": (32, 96)}.items():
        mode = object.__new__(up.Txt2imgMode)
        mode.latents_shape = (2, 4, lh, lw)
        mode.unet_sample_size = 64
        mode.generators = gens(seeds)
        mode.device = "cpu"
        mode.latents_dtype = torch.float32
        mode.scheduler = _Sched()
        lt = mode.generateLatents()
        # keep fixtures small: store a strided sample + full-tensor checksum
        out[f"txt2img_{name}_shape"] = np.array(lt.shape)
        out[f"txt2img_{name}_sum"] = np.array(lt.double().sum().item())
        out[f"txt2img_{name}_abs_sum"] = np.array(lt.double().abs().sum().item())
        out[f"txt2img_{name}_sample"] = lt[:, :, ::7, ::5].numpy()

    # (8) mask helpers -----------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    mask = (torch.rand(1, 1, 64, 64, generator=g) > 0.3).float()
    mask[:, :, 16:40, 8:32] = 1.0
    out["mask_in"] = mask.numpy()
    out["mask_boxmin"] = up.downscale_boxop_2d(mask, 8, "min").numpy()
    out["mask_boxmax"] = up.downscale_boxop_2d(mask, 8, "max").numpy()
    mp = up.MaskProcessorMixin()
    soft = torch.rand(1, 1, 16, 16, generator=g)
    out["mask_soft"] = soft.numpy()
    out["mask_latent"] = mp.mask_to_latent_mask(mask).numpy()
    out["mask_round"] = mp.round_mask(soft).numpy()
    out["mask_round_high"] = mp.round_mask_high(soft).numpy()
    out["mask_round_low"] = mp.round_mask_low(soft).numpy()

    # (9) EnhancedInpaintMode._fillWithShapedNoise (strength >= 1: shuffle-fill of the repaint area) ------------
    gsn = torch.Generator().manual_seed(21)
    sn_lat = torch.randn(2, 4, 8, 8, generator=gsn)
    sn_mask = torch.ones(1, 1, 64, 64)
    sn_mask[:, :, 16:48, 24:64] = 0.0                       # 1K0D: zero = repaint
    sn_mask[:, :, 8:16, 24:64] = 0.4                         # soft edge
    for tag, sns in (("s1", 1.0), ("s07", 0.7)):
        em = object.__new__(up.EnhancedInpaintMode)
        em.generators = gens(seeds)
        em.latents_dtype = torch.float32
        em.shaped_noise_strength = sns
        em.latent_mask = torch.cat([em.mask_to_latent_mask(sn_mask)] * 2)
        em.latent_high_mask = em.round_mask_high(em.latent_mask)
        em.latent_low_mask = em.round_mask_low(em.latent_mask)
        out[f"shaped_noise_{tag}_out"] = em._fillWithShapedNoise(sn_lat.clone()).numpy()
    out["shaped_noise_latents"] = sn_lat.numpy()
    out["shaped_noise_mask"] = sn_mask.numpy()

    # (9b) histogram matching of the outmask composite (gyre/match_histograms.py via images.match_histograms) -------
    from gyre.match_histograms import match_histograms as ref_match
    ghm = torch.Generator().manual_seed(33)
    hm_img = (torch.rand(2, 12, 10, 3, generator=ghm) ** 2 * 255).round().to(torch.uint8).numpy()
    hm_ref = (torch.rand(2, 12, 10, 3, generator=ghm) * 200 + 30).round().to(torch.uint8).numpy()
    out["histmatch_image_u8"] = hm_img
    out["histmatch_reference_u8"] = hm_ref
    out["histmatch_out_u8"] = ref_match(hm_img, hm_ref, channel_axis=3)

    # (10) VaeApproximator ---------------------------------------------------------
    from gyre.pipeline.vae_approximator import VaeApproximator
    va = VaeApproximator(device="cpu", dtype=torch.float32)
    l4 = randtools.batched_randn([1, 4, 8, 8], gens([11]), "cpu", torch.float32)
    out["vae_approx_in"] = l4.numpy()
    out["vae_approx_out"] = va(l4).numpy()

    # (9) prompt attention parser + token/weight padding (long-prompt weighting) ------------------------
    import json
    from gyre.pipeline.text_embedding import lpw_text_embedding as lpw
    prompts = ["normal text", "an (important) word", "(unbalanced", "\\(literal\\]", "(unnecessary)(parens)",
               "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).", "", "[[deep]] (a:2) b) c] \\\\ d",
               "(x:1.5)(y:0.25) [z", "no (nested [mix (of:3) all] kinds) here"]
    out["lpw_prompts"] = np.array(json.dumps(prompts))
    out["lpw_parsed"] = np.array(json.dumps([lpw.parse_prompt_attention(t) for t in prompts]))
    toks = [[5, 6, 7], [], list(range(100, 180))]
    wts = [[1.1, 1.0, 0.5], [], [1.0 + 0.01 * i for i in range(80)]]
    import copy
    for nbm in (True, False):
        t2, w2 = lpw.pad_tokens_and_weights(copy.deepcopy(toks), copy.deepcopy(wts), 152, 49406, 49407,
                                            no_boseos_middle=nbm, chunk_length=77)
        out[f"lpw_pad_tokens_nbm{int(nbm)}"] = np.array(t2, dtype=np.int64)
        out[f"lpw_pad_weights_nbm{int(nbm)}"] = np.array(w2, dtype=np.float64)
    # mean-preserving weighting as applied at lpw_text_embedding.py:366-375
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(2, 77, 16, generator=g) + 0.3
    w = torch.rand(2, 77, generator=g) * 1.5
    prev = emb.mean(axis=[-2, -1])
    e2 = emb * w.unsqueeze(-1)
    e2 = e2 * (prev / e2.mean(axis=[-2, -1])).unsqueeze(-1).unsqueeze(-1)
    out["lpw_weighting_in"] = emb.numpy(); out["lpw_weighting_w"] = w.numpy(); out["lpw_weighting_out"] = e2.numpy()

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", len(out), "arrays:", sorted(out))


if __name__ == "__main__":
    main()
