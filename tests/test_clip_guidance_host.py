"""CLIP guidance through the host pipeline on CPU (oracle UNet / VAE under torch autograd, a tiny random CLIP):
orchestration of gyre_amd/clipguided.py inside GyrePipeline - reference unified_pipeline.py:1876-1909, 2373-2406."""
from types import SimpleNamespace

import pytest
import torch

from gyre_amd import clipguided as CG
from gyre_amd.pipeline import GyrePipeline
from gyre_amd.resize import resize_right
from test_host_pipeline import OracleUNet, OracleVAE, tiny  # noqa: F401  (fixture)


def patch_embed_as_matmul(clip):
    return CG.patch_embedding_as_matmul(clip)


def tiny_clip():
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(1)
    clip = CLIPModel(CLIPConfig(text_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                 vocab_size=1000, max_position_embeddings=16, bos_token_id=1, eos_token_id=2),
                                vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                                   num_attention_heads=2, image_size=32, patch_size=8),
                                projection_dim=16)).eval()
    for p in clip.parameters():
        p.requires_grad_(False)
    patch_embed_as_matmul(clip)
    fe = SimpleNamespace(image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711],
                         size={"shortest_edge": 32})
    return clip, fe


def make(tiny, clip=True):
    ucfg, vcfg, usd, vsd, text, unc = tiny
    cm, fe = tiny_clip() if clip else (None, None)
    pipe = GyrePipeline(OracleUNet(usd, ucfg), OracleVAE(vsd, vcfg), device="cpu", clip_model=cm, feature_extractor=fe)
    kw = dict(seeds=[420420420, 420420421], text_embeddings=text, uncond_embeddings=unc, height=128, width=128,
              num_inference_steps=4, guidance_scale=7.5)
    ids = torch.randint(3, 900, (2, 16), generator=torch.Generator().manual_seed(3))
    return pipe, kw, ids


@pytest.mark.parametrize("sampler,extra", [("euler", {}), ("dpmpp_2m", dict(clip_guidance_base="mixed")),
                                           ("euler_a", dict(vae_cutouts=2, approx_cutouts=0)), ("ddim", {})])
def test_guidance_runs_and_changes_the_image(tiny, sampler, extra):
    pipe, kw, ids = make(tiny)
    plain = pipe(sampler=sampler, **kw)
    plain_evals = pipe.last_unet_evals
    guided = pipe(sampler=sampler, clip_guidance_scale=0.5, clip_input_ids=ids, **extra, **kw)
    mode = pipe.last_clip_modes[0]
    assert guided.shape == plain.shape and bool(torch.isfinite(guided).all())
    assert float((guided - plain).abs().max()) > 1e-3
    assert len(mode.lossavg) == mode.grad_evals >= 4 and all(0 < v < 10 for v in mode.lossavg)
    if sampler != "ddim":
        # guided base: one extra (conditional-stem) evaluation per guided step on top of the plain loop; mixed: none
        base = extra.get("clip_guidance_base", "guided")
        assert pipe.last_unet_evals == plain_evals + (mode.grad_evals if base == "guided" else 0)
    again = pipe(sampler=sampler, clip_guidance_scale=0.5, clip_input_ids=ids, **extra, **kw)
    assert torch.equal(guided, again)                                    # per-image generators: reproducible


def test_one_call_over_both_cfg_halves_equals_the_two_calls_of_the_reference(tiny):
    """Guided base, k-diffusion sampler: the reference evaluates the conditional stem under autograd and then the unconditional stem on
    the same latents (clipguided.py:218-241).  With the parallel CFG binding the host mode makes ONE call on cat[x, x] /
    cat[uncond, cond] and differentiates the conditional half (wrap_guidance_unet(unet_both=...)); with the sequential binding it
    makes the reference's two calls.  Same images (fp32 oracle UNet: batch 2B vs B only reorders nothing inside a sample)."""
    pipe, kw, ids = make(tiny)
    one = pipe(sampler="dpmpp_2m", clip_guidance_scale=0.5, clip_input_ids=ids, **kw)
    loss_one = list(pipe.last_clip_modes[0].lossavg)
    two = pipe(sampler="dpmpp_2m", clip_guidance_scale=0.5, clip_input_ids=ids, cfg_execution="sequential", **kw)
    loss_two = list(pipe.last_clip_modes[0].lossavg)
    assert len(loss_one) == len(loss_two) >= 4
    assert max(abs(a - b) for a, b in zip(loss_one, loss_two)) < 1e-4
    assert float((one - two).abs().max()) < 2e-3, float((one - two).abs().max())


def test_guidance_follows_the_dtype_the_clip_model_was_loaded_in(tiny):
    """The reference feeds its CLIP model tensors of the pipeline's own (fp16) dtype (clipguided.py:400-404); here the decoded image is
    fp32 and is cast to the model's dtype, the loss stays fp32: a bf16 CLIP model guides like the fp32 one, up to bf16 noise."""
    pipe, kw, ids = make(tiny)
    ref = pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, **kw)
    plain = pipe(sampler="euler", **kw)
    pipe.clip_model = pipe.clip_model.to(torch.bfloat16)
    low = pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, **kw)
    assert bool(torch.isfinite(low).all())
    d_ref, d_low = (ref - plain).flatten().double(), (low - plain).flatten().double()
    cos = float(d_ref @ d_low / (d_ref.norm() * d_low.norm()))
    assert cos > 0.9, cos                                             # the same push, not the same bits
    assert float((low - ref).abs().max()) < 0.5 * float((ref - plain).abs().max())


def test_guidance_without_a_clip_model_is_ignored_with_a_warning(tiny, capsys):
    pipe, kw, ids = make(tiny, clip=False)
    plain = pipe(sampler="euler", **kw)
    got = pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, **kw)
    assert "CLIP guidance passed to a pipeline without a CLIP model" in capsys.readouterr().out
    assert torch.equal(plain, got)


def test_guidance_argument_errors(tiny):
    pipe, kw, ids = make(tiny)
    with pytest.raises(ValueError, match="v-prediction"):
        pipe(sampler="ddim", clip_guidance_scale=0.5, clip_input_ids=ids, prediction_type="v_prediction", **kw)
    with pytest.raises(ValueError, match="clip_input_ids"):
        pipe(sampler="euler", clip_guidance_scale=0.5, **kw)
    with pytest.raises(ValueError, match="guided"):
        pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, clip_guidance_base="other", **kw)
    with pytest.raises(ValueError, match="must be equal"):
        pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, vae_cutouts=1, approx_cutouts=2, **kw)


def test_flat_loss_switches_guidance_off(tiny):
    pipe, kw, ids = make(tiny)
    kw = dict(kw, num_inference_steps=8)
    pipe(sampler="euler", clip_guidance_scale=0.5, clip_input_ids=ids, clip_gradient_length=2, clip_gradient_threshold=100.0,
         clip_gradient_maxloss=100.0, **kw)
    mode = pipe.last_clip_modes[0]
    assert mode.flatloss and mode.grad_evals == 3                         # history must exceed gradient_length, then stop


def test_cutouts_are_drawn_per_image_and_grouped_by_image():
    gens = lambda: [torch.Generator().manual_seed(7), torch.Generator().manual_seed(8)]
    x = torch.randn(2, 3, 40, 48, generator=torch.Generator().manual_seed(0))
    both = CG.MakeCutouts(16, gens())(x, 3)
    assert both.shape == (6, 3, 16, 16)
    solo = CG.MakeCutouts(16, gens()[1:])(x[1:], 3)                       # image 1 alone, with its own generator
    assert torch.equal(both[3:], solo)
    # minimum crop = cut_size, maximum = the short edge; identity when the image already has cut_size
    same = CG.MakeCutouts(16, gens()[:1])(x[:1, :, :16, :16], 1)
    assert torch.allclose(same, x[:1, :, :16, :16], atol=1e-5)


def test_resize_right_restatement_properties():
    x = torch.randn(1, 2, 24, 30, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(resize_right(x, out_shape=(24, 30), pad_mode="reflect"), x, atol=1e-6)     # identity
    c = torch.full((1, 1, 20, 20), 0.75)
    for shape in ((7, 9), (20, 33), (64, 64)):                           # rows sum to one, also with reflect padding
        assert torch.allclose(resize_right(c, out_shape=shape, pad_mode="reflect"), torch.full((1, 1) + shape, 0.75), atol=1e-6)
    ramp = torch.arange(32.0).view(1, 1, 1, 32).expand(1, 1, 8, 32).contiguous()
    up = resize_right(ramp, out_shape=(8, 64), pad_mode="replicate")
    assert torch.allclose(up[..., 8:56], torch.linspace(-0.25, 31.25, 64)[8:56].expand(1, 1, 8, 48), atol=1e-4)  # cubic keeps lines
    down = resize_right(ramp, out_shape=(8, 8), pad_mode="reflect")      # antialiased: the box of 4 inputs -> their mean
    assert torch.allclose(down[0, 0, 0, 2:6], torch.tensor([9.5, 13.5, 17.5, 21.5]), atol=0.05)


def test_spherical_loss_and_approximator():
    a = torch.nn.functional.normalize(torch.randn(4, 16, generator=torch.Generator().manual_seed(2)), dim=-1)
    assert torch.allclose(CG.spherical_dist_loss(a, a), torch.zeros(4), atol=1e-6)
    e = torch.eye(16)
    assert torch.allclose(CG.spherical_dist_loss(e[:4], e[4:8]), torch.full((4,), 2 * (torch.pi / 4) ** 2), atol=1e-5)   # orthogonal
    z = torch.zeros(1, 4, 2, 2)
    z[0, 3] = 1.0
    assert torch.allclose(CG.VaeApproximator()(z)[0, :, 0, 0], torch.tensor([-0.184, -0.271, -0.473]))
