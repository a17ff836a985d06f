"""CPU tests of the hires-fix / graft host code (gyre_amd/resize.py, hires.py, pipeline hires path) against the
independent loop restatement in oracle/hires_ref.py and through its defining properties.  The third-party pieces
(ResizeRight, easing_functions) are absent from the reference tree, so there are no golden vectors: parity unpinned."""
import math
from types import SimpleNamespace

import pytest
import torch

from gyre_amd import hires as H
from gyre_amd import schedulers as PS
from gyre_amd.pipeline import GyrePipeline
from gyre_amd.resize import lanczos2, resize_lanczos2
from oracle import hires_ref as R
from test_host_pipeline import OracleUNet, OracleVAE, tiny  # noqa: F401  (fixture)


def gens(seeds):
    return [torch.Generator().manual_seed(int(s)) for s in seeds]


@pytest.mark.parametrize("h,w,scale", [(8, 8, 1.5), (12, 8, 2 / 3), (7, 9, 1.37), (16, 16, 0.5), (5, 11, 2.0), (1, 6, 1.5)])
def test_lanczos_resize_matches_loop_oracle(h, w, scale):
    x = torch.randn(2, 3, h, w, generator=torch.Generator().manual_seed(h * 100 + w))
    got, ref = resize_lanczos2(x, scale), R.resize_lanczos2_ref(x, scale)
    assert got.shape == ref.shape == (2, 3, math.ceil(h * scale), math.ceil(w * scale))
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_lanczos_properties():
    assert float(lanczos2(torch.tensor(0.0))) == pytest.approx(1.0)
    assert float(lanczos2(torch.tensor(1.0)).abs()) < 1e-6 and float(lanczos2(torch.tensor(2.5))) == 0.0
    c = torch.full((1, 2, 9, 13), 0.75)
    assert torch.allclose(resize_lanczos2(c, 1.7), torch.full((1, 2, 16, 23), 0.75), atol=1e-6)   # partition of unity
    x = torch.randn(1, 1, 10, 10, generator=torch.Generator().manual_seed(0))
    assert torch.equal(resize_lanczos2(x, 1), x)
    assert torch.allclose(resize_lanczos2(x.flip(-1), 1.5), resize_lanczos2(x, 1.5).flip(-1), atol=1e-6)  # symmetric grid
    a, b = x, torch.randn(1, 1, 10, 10, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(resize_lanczos2(2 * a - b, 0.8), 2 * resize_lanczos2(a, 0.8) - resize_lanczos2(b, 0.8), atol=1e-5)
    with pytest.raises(ValueError):
        resize_lanczos2(x, 0)


def test_easing_curves():
    e = H.Easing(0, 0, 0.667, "cubic")
    for u in (0, 0.05, 0.2, 0.3335, 0.5, 0.66, 0.667, 0.9):
        assert e.interp(u) == pytest.approx(R.ease_ref("cubic", 0, 0, 0.667, u), abs=1e-12)
    g = H.Easing(0, 0.1, 0.3, "sine")
    assert g.interp(0.05) == 0 and g.interp(0.2) == pytest.approx(0.5) and g.interp(0.31) == 1
    for name, fn in H.EASINGS.items():
        assert fn(0) == pytest.approx(0, abs=1e-3) and fn(1) == pytest.approx(1, abs=1e-3) and fn(0.5) == pytest.approx(0.5, abs=1e-9), name
        vals = [fn(i / 50) for i in range(51)]
        assert all(b >= a - 1e-12 for a, b in zip(vals, vals[1:])), name
    assert H.Easing(0.25, 0, 1, "linear").interp(0.5) == pytest.approx(0.625)
    with pytest.raises(ValueError):
        H.Easing(0, 0, 1, "bounce")


def test_scale_factors_and_scale_into():
    # 96x64 -> 64x64: oos 1 fits the long side inside (scale 2/3), oos 0 fills (scale 1)
    assert H.down_scale_factor((96, 64), (64, 64), 1.0) == pytest.approx(2 / 3)
    assert H.down_scale_factor((96, 64), (64, 64), 0.0) == pytest.approx(1.0)
    assert H.up_scale_factor((64, 64), (96, 64), 1.0) == pytest.approx(1.5)
    x = torch.randn(1, 4, 12, 8, generator=torch.Generator().manual_seed(3))
    out = H.scale_into(x, 2 / 3, target_shape=(1, 4, 8, 8))           # 8x6 placed in 8x8, border replicated
    small = resize_lanczos2(x, 2 / 3)
    assert out.shape == (1, 4, 8, 8) and torch.equal(out[..., 1:7], small) and torch.equal(out[..., 0], small[..., 0])
    canvas = torch.zeros(1, 4, 12, 12)
    got = H.scale_into(x, 1.0, target=canvas)
    assert got is canvas and torch.equal(canvas[..., 2:10], x) and float(canvas[..., :2].abs().max()) == 0
    crop = H.scale_into(x, 1.0, target_shape=(1, 4, 8, 8))
    assert torch.equal(crop, x[:, :, 2:10])
    with pytest.raises(ValueError):
        H.scale_into(x, 1.0)
    nat = H.image_to_natural(8, torch.rand(1, 3, 12, 16), 1.0)
    assert nat.shape == (1, 3, 8, 8)


@pytest.mark.parametrize("u", [0.0, 0.2, 0.45, 0.7])
def test_hires_wrapper_matches_loop_oracle(u):
    B, th, h, w = 2, 4, 6, 8
    nat = lambda x, s, uu: x * 0.5 + 0.1
    hi = lambda x, s, uu: x * 0.25 - 0.2
    lat = torch.randn(2 * B, 4, h, w, generator=torch.Generator().manual_seed(11))
    wrap = H.HiresUnetWrapper(nat, hi, gens([5, 6]), [th, th], 0.6)
    got = wrap(lat, torch.tensor(3.0), u)
    g = gens([5, 6])
    rand_lo = PS.batched_rand([B, 4, th, th], g, "cpu", torch.float32)
    rand_hi = PS.batched_rand([B, 4, h, w], g, "cpu", torch.float32)
    ref = R.hires_step_ref(nat, hi, lat, torch.tensor(3.0), u, rand_lo, rand_hi, (th, th), 0.6)
    assert got.shape == ref.shape == lat.shape
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)
    if u >= 0.667:   # exchange over: natural half is passed through untouched, only the hires denoiser ran
        assert torch.equal(got[:B], lat[:B]) and torch.equal(got[B:], hi(lat[B:], None, u))


def test_hires_merge_and_split():
    left, right = torch.ones(2, 4, 4, 4), torch.full((2, 4, 8, 6), 2.0)
    m = H.HiresUnetWrapper.merge_initial_latents(left, right)
    assert m.shape == (4, 4, 8, 6) and torch.equal(m[2:], right)
    assert float(m[:2].sum()) == 2 * 4 * 16 and torch.equal(m[:2, :, 2:6, 1:5], left)
    assert torch.equal(H.HiresUnetWrapper.split_result(None, m), right)


def test_graft_blend():
    root = lambda x, s, u: torch.zeros_like(x)
    top = lambda x, s, u: torch.ones_like(x)
    x = torch.zeros(2, 4, 16, 16)
    gr = H.GraftUnets(root, top, gens([1, 2]))
    assert float(gr(x, None, 0.05).sum()) == 0 and float(gr(x, None, 0.5).mean()) == 1
    mid = gr(x, None, 0.2)                      # p = 0.5: about half the elements come from each
    assert 0.4 < float(mid.mean()) < 0.6 and set(mid.unique().tolist()) == {0.0, 1.0}
    custom = H.GraftUnets(root, top, gens([1, 2]), blend={"start": 0.0, "end": 1.0, "easing": "linear"})
    assert 0.2 < float(custom(x, None, 0.3).mean()) < 0.4
    assert H.GraftUnets.merge_initial_latents("l", "r") == "l" and H.GraftUnets.split_result("l", "r") == "r"


def test_pipeline_hires_fix_cpu(tiny):
    """192x256 with the tiny UNet (natural size 128 px): the hires tree runs both leaves while p < 0.999."""
    ucfg, vcfg, usd, vsd, text, unc = tiny
    calls = []

    class CountingUNet(OracleUNet):
        def __call__(self, latents, t, encoder_hidden_states=None):
            calls.append(tuple(latents.shape))
            return super().__call__(latents, t, encoder_hidden_states=encoder_hidden_states)

    pipe = GyrePipeline(CountingUNet(usd, ucfg), OracleVAE(vsd, vcfg), device="cpu")
    kw = dict(text_embeddings=text, uncond_embeddings=unc, height=192, width=256, num_inference_steps=6,
              sampler="euler_a", output_type="latent")
    out = pipe(seeds=[3, 4], **kw)
    assert out.shape == (2, 4, 24, 32) and bool(torch.isfinite(out).all())
    nat_calls = [c for c in calls if c[-2:] == (16, 16)]
    hi_calls = [c for c in calls if c[-2:] == (24, 32)]
    # u = i/5 for 6 steps: p(u) < 0.999 for u in {0, .2, .4, .6} -> 4 natural evals, 6 hires evals, CFG batch 4
    assert len(hi_calls) == 6 and len(nat_calls) == 4 and all(c[0] == 4 for c in calls)
    assert pipe.last_unet_evals == 10
    assert torch.equal(out, pipe(seeds=[3, 4], **kw))                               # deterministic
    one = pipe(seeds=[4], **{**kw, "text_embeddings": text[1:], "uncond_embeddings": unc[1:]})
    assert torch.allclose(out[1:], one, rtol=1e-4, atol=1e-3)                       # batch independent
    calls.clear()
    plain = pipe(seeds=[3, 4], hires_fix=False, **kw)
    assert len(calls) == 6 and not torch.allclose(plain, out)
    # at or just above the natural size the fix is skipped (threshold 3.33 %)
    calls.clear()
    pipe(seeds=[3], **{**kw, "height": 128, "width": 128, "text_embeddings": text[:1], "uncond_embeddings": unc[:1]})
    assert all(c[-2:] == (16, 16) for c in calls) and len(calls) == 6
    with pytest.raises(ValueError, match="Hires fix"):
        pipe(seeds=[3, 4], **{**kw, "sampler": "ddim"})


def test_pipeline_hires_img2img_and_inpaint_cpu(tiny):
    ucfg, vcfg, usd, vsd, text, unc = tiny
    pipe = GyrePipeline(OracleUNet(usd, ucfg), OracleVAE(vsd, vcfg), device="cpu")
    image = torch.rand(1, 3, 192, 192, generator=torch.Generator().manual_seed(2))
    mask = torch.zeros(1, 1, 192, 192)
    mask[:, :, 48:144, 48:144] = 1.0
    kw = dict(seeds=[1, 2], text_embeddings=text, uncond_embeddings=unc, height=192, width=192, num_inference_steps=6,
              sampler="euler", image=image, strength=0.75, output_type="latent")
    a = pipe(**kw)
    b = pipe(mask_image=mask, **kw)
    assert a.shape == b.shape == (2, 4, 24, 24) and bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    assert not torch.allclose(a, b)
