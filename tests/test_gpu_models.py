"""Model-level parity (-m gpu): the native UNet / VAE behind the nn.Module shells against the
fp32 CPU oracle on the same seeded weights and inputs.

Tolerances (bf16 storage / fp32 accumulate vs fp32 oracle; SURVEY.md section 8d parity gates):
single UNet call rel-L2 <= 3e-2, VAE <= 3e-2.
"""
import pytest
import torch

from gyre_amd import _lib, config as gcfg
from gyre_amd import weights
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gpu_util import HDT, DEV, randn, rel_l2, report
from oracle import models_ref as M

pytestmark = pytest.mark.gpu


def make_unet(cfg, seed=0):
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), seed)
    net = GyreHipUNet(cfg)
    net.load_state_dict(sd)
    return net.to(DEV), sd


def make_vae(cfg, seed=0):
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg), seed)
    net = GyreHipVAE(cfg)
    net.load_state_dict(sd)
    return net.to(DEV), sd


@pytest.mark.parametrize("inch,H,W,S", [(4, 16, 16, 77), (9, 16, 24, 77), (4, 8, 8, 154),
                                        # latent sizes that are not multiples of 8: odd levels, upsample to the skip size
                                        (4, 18, 14, 77), (4, 9, 15, 77), (4, 3, 5, 77)])
def test_tiny_unet_parity(inch, H, W, S):
    cfg = gcfg.tiny_unet(inch)
    net, sd = make_unet(cfg)
    x = randn(2, inch, H, W, seed=1)
    t = torch.tensor([981, 17])
    ctx = randn(2, S, cfg.cross_attention_dim, seed=2)
    ref = M.unet_forward(sd, cfg, x, t, ctx)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    assert got.dtype == torch.float32 and got.shape == ref.shape
    report(f"tiny unet in{inch} {H}x{W} S{S}", got.cpu(), ref, 3e-2)


def test_tiny_unet_dtypes_and_scalar_t():
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 16, 16, seed=1)
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    ref = M.unet_forward(sd, cfg, x, torch.tensor([500, 500]), ctx)
    for dt in (torch.bfloat16, torch.float16):
        got = net(x.to(DEV, dt), 500, encoder_hidden_states=ctx.to(DEV, dt)).sample
        assert got.dtype == dt
        report(f"tiny unet io {dt}", got.float().cpu(), ref, 4e-2)


def test_unet_batch_independence_bit_exact():
    """reference property tests/batch_independance.py:15-27, as a bit-exact check."""
    cfg = gcfg.tiny_unet()
    net, _ = make_unet(cfg)
    x = randn(3, 4, 16, 16, seed=3).to(DEV)
    t = torch.tensor([900, 500, 20], device=DEV)
    ctx = randn(3, 77, cfg.cross_attention_dim, seed=4).to(DEV)
    full = net(x, t, encoder_hidden_states=ctx).sample
    for i in range(3):
        one = net(x[i:i + 1].contiguous(), t[i:i + 1], encoder_hidden_states=ctx[i:i + 1].contiguous()).sample
        assert torch.equal(full[i:i + 1], one), f"sample {i} differs between batch compositions"
    again = net(x, t, encoder_hidden_states=ctx).sample
    assert torch.equal(full, again)


def test_unet_errors():
    cfg = gcfg.tiny_unet()
    net, _ = make_unet(cfg)
    x = randn(1, 4, 16, 16).to(DEV)
    ctx = randn(1, 77, cfg.cross_attention_dim).to(DEV)
    with pytest.raises(ValueError):
        net(x[:, :3].contiguous(), 1, encoder_hidden_states=ctx)
    out = net(randn(1, 4, 12, 12).to(DEV), 1, encoder_hidden_states=ctx).sample  # not a multiple of 8: fine (diffusers too)
    assert out.shape == (1, 4, 12, 12) and bool(torch.isfinite(out).all())
    with pytest.raises(NotImplementedError):
        net(x, 1, encoder_hidden_states=ctx, adapter_states=[[x]])
    with pytest.raises(RuntimeError):
        GyreHipUNet(cfg)(x.cpu(), 1, encoder_hidden_states=ctx.cpu())  # no CPU fallback


def test_tiny_vae_parity():
    cfg = gcfg.tiny_vae()
    net, sd = make_vae(cfg)
    z = randn(2, 4, 8, 12, seed=5)
    ref = M.vae_decode(sd, cfg, z)
    got = net.decode(z.to(DEV)).sample
    report("tiny vae decode", got.cpu(), ref, 3e-2)
    img = randn(2, 3, 64, 96, seed=6).clamp(-1, 1)
    ref = M.vae_encode_moments(sd, cfg, img)
    dist = net.encode(img.to(DEV)).latent_dist
    report("tiny vae encode moments", dist.parameters.cpu(), ref, 3e-2)
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    s = dist.sample(generator=g1)
    ref_s = M.vae_posterior_sample(ref, g2)
    report("tiny vae posterior sample", s.cpu(), ref_s, 3e-2)


def test_sd15_unet_parity_full_size():
    """Full SD1.5 UNet (859.5 M params), CFG pair at 64x64 latents, vs the fp32 CPU oracle."""
    cfg = gcfg.sd15_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 64, 64, seed=7)
    t = torch.tensor([981, 981])
    ctx = randn(2, 77, 768, seed=8)
    ref = M.unet_forward(sd, cfg, x, t, ctx)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    report("SD1.5 unet 2x4x64x64", got.cpu(), ref, 3e-2)
    # the same call with the transformer blocks' LayerNorms as separate passes instead of folded into their GEMMs
    # (planner debug bit 11): both forms sit at the same distance from the oracle
    L = _lib.lib()
    old = L.gyre_debug_gemm_ablation(0x800)
    try:
        plain = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    finally:
        L.gyre_debug_gemm_ablation(old)
    report("SD1.5 unet 2x4x64x64, separate LayerNorm passes", plain.cpu(), ref, 3e-2)
    e_f, e_p = rel_l2(got.cpu(), ref), rel_l2(plain.cpu(), ref)
    assert e_f < 1.3 * e_p + 1e-3, (e_f, e_p)


def test_sd15_vae_decode_full_size():
    cfg = gcfg.sd15_vae()
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg), 0)
    net = GyreHipVAE(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    z = randn(1, 4, 64, 64, seed=9)
    ref = M.vae_decode(sd, cfg, z)
    got = net.decode(z.to(DEV)).sample
    assert got.shape == (1, 3, 512, 512)
    report("SD1.5 vae decode 1x4x64x64", got.cpu(), ref, 3e-2)


def test_sd15_vae_encode_full_size():
    """VAE encode at the real 512x512 size against the oracle's moments (unified_pipeline.py:309-313 `vae.encode(x)
    .latent_dist`): mean and clamped log-variance, plus the posterior sample drawn from the same generator."""
    cfg = gcfg.sd15_vae()
    sd = weights.synthetic_state_dict(weights.vae_param_shapes(cfg), 0)
    net = GyreHipVAE(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    img = (torch.stack([xx, yy, xx * yy])[None] + 0.25 * randn(1, 3, 512, 512, seed=4)).clamp(-1, 1)
    ref = M.vae_encode_moments(sd, cfg, img)
    dist = net.encode(img.to(DEV)).latent_dist
    assert dist.parameters.shape == (1, 8, 64, 64)
    report("SD1.5 vae encode 1x3x512x512 (moments)", dist.parameters.float().cpu(), ref, 3e-2)
    report("SD1.5 vae encode mean", dist.mean.float().cpu(), ref[:, :4], 3e-2)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    report("SD1.5 vae posterior sample", dist.sample(generator=g1).float().cpu(), M.vae_posterior_sample(ref, g2), 3e-2)


def test_tiny_sdxl_topology_parity():
    """SDXL-style UNet (3 levels, depth 0/2/3, linear projections, text_time added conditioning) - an extension
    beyond the reference (BASELINE config 4); parity against the build's own oracle."""
    cfg = gcfg.tiny_sdxl_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 16, 16, seed=11)
    t = torch.tensor([981, 17])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=12)
    ac = {"text_embeds": randn(2, 32, seed=13), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2)}
    ref = M.unet_forward(sd, cfg, x, t, ctx, added_cond=ac)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=ac).sample
    report("tiny sdxl-topology unet", got.cpu(), ref, 3e-2)
    with pytest.raises(ValueError):
        net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV))


def test_context_cache_matches_uncached_and_tracks_changes():
    """The shell projects the text context once (gyre_unet_set_context) and reuses it while the SAME tensor is passed;
    a new tensor, or an in-place edit of the old one, must be picked up."""
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 16, 16, seed=21).to(DEV)
    t = torch.tensor([700, 30], device=DEV)
    c1 = randn(2, 77, cfg.cross_attention_dim, seed=22).to(DEV)
    c2 = randn(2, 77, cfg.cross_attention_dim, seed=23).to(DEV)
    a1 = net(x, t, encoder_hidden_states=c1).sample
    a1b = net(x, t, encoder_hidden_states=c1).sample          # cached path
    assert torch.equal(a1, a1b)
    a2 = net(x, t, encoder_hidden_states=c2).sample            # different tensor -> re-projected
    assert not torch.equal(a1, a2)
    ref2 = M.unet_forward(sd, cfg, x.cpu(), t.cpu(), c2.cpu())
    report("ctx cache second context", a2.cpu(), ref2, 3e-2)
    c2.copy_(c1)                                               # in-place edit bumps the version counter
    a3 = net(x, t, encoder_hidden_states=c2).sample
    assert torch.equal(a3, a1)
    # raw C ABI: ctx == NULL without a matching set_context is an error, not a crash
    import ctypes as C
    from gyre_amd import _lib
    L = _lib.lib()
    h = net._handle
    ws = torch.empty(L.gyre_unet_workspace_bytes(C.c_void_p(h), 3, 16, 16, 77) + 256, dtype=torch.uint8, device=DEV)
    x3 = randn(3, 4, 16, 16).to(DEV); t3 = torch.zeros(3, dtype=torch.int64, device=DEV); o3 = torch.empty_like(x3)
    rc = L.gyre_unet_forward(C.c_void_p(h), None, C.c_void_p(x3.data_ptr()), 0, C.c_void_p(t3.data_ptr()), None, 0, 3, 16, 16, 77,
                             C.c_void_p((ws.data_ptr() + 255) & ~255), ws.numel() - 256, C.c_void_p(o3.data_ptr()), 0)
    assert rc == -1 and b"set_context" in L.gyre_last_error()


def test_uniform_timestep_fast_path_is_bit_identical():
    """A scalar timestep (what the samplers pass) lets the time-embedding MLP run for one row that every resnet reads
    (gyre_unet_hint_uniform_timestep): same bits as the per-sample tensor form, at full size and at a ragged batch.
    Round 5: up to four rows take the one-row-specialised kernels of the chain (k_rowvec_small: embedding and SiLU applied on
    load, three launches instead of six); the B = 6 cases put the general kernels (B rows, separate embedding / SiLU launches)
    on the tensor side and the specialised ones on the scalar side - same bits."""
    for cfg, B, hw in ((gcfg.tiny_unet(), 3, 16), (gcfg.sd15_unet(), 4, 32), (gcfg.tiny_unet(), 6, 16), (gcfg.sd15_unet(), 6, 16)):
        net, _ = make_unet(cfg)
        x = randn(B, 4, hw, hw, seed=31).to(DEV)
        ctx = randn(B, 77, cfg.cross_attention_dim, seed=32).to(DEV)
        a = net(x, 637, encoder_hidden_states=ctx).sample
        b = net(x, torch.full((B,), 637, device=DEV), encoder_hidden_states=ctx).sample
        c = net(x, torch.tensor(637), encoder_hidden_states=ctx).sample
        assert torch.equal(a, b) and torch.equal(a, c)
        t2 = torch.tensor([637] * (B - 1) + [12], device=DEV)          # per-sample timesteps still honoured
        d = net(x, t2, encoder_hidden_states=ctx).sample
        assert torch.equal(d[:B - 1], a[:B - 1]) and not torch.equal(d[B - 1], a[B - 1])


@pytest.mark.parametrize("which,B,hw", [("tiny", 4, 16), ("sd15", 4, 64), ("sd15", 2, 32), ("inpaint9", 2, 32)])
def test_cfg_pair_batches_share_their_prefix(which, B, hw):
    """CFG-parallel batches (reference unet/cfg.py:49-57: cat[x, x] against cat[uncond, cond]): with the caller's hint
    (modules.cfg_pairs -> gyre_unet_hint_cfg_pairs) conv_in, the first resnet and the first self-attention run once per
    pair.  Same numbers as the plain call (bit-equal where the planner picks the same kernels for B/2 samples, else two
    summation orders of the same GroupNorm sums), fewer FLOPs, and the fp32 oracle still agrees."""
    from gyre_amd import _lib
    from gyre_amd.modules import cfg_pairs
    cfg = {"tiny": gcfg.tiny_unet(), "sd15": gcfg.sd15_unet(), "inpaint9": gcfg.sd15_unet(in_channels=9)}[which]
    net, sd = make_unet(cfg)
    half = randn(B // 2, cfg.in_channels, hw, hw, seed=51)
    x = torch.cat([half, half]).to(DEV)
    ctx = randn(B, 77, cfg.cross_attention_dim, seed=52).to(DEV)
    plain = net(x, 500, encoder_hidden_states=ctx).sample
    n_plain = _lib.lib().gyre_last_launch_count()
    with cfg_pairs():
        shared = net(x, 500, encoder_hidden_states=ctx).sample
        n_shared = _lib.lib().gyre_last_launch_count()
        again = net(x, 500, encoder_hidden_states=ctx).sample
    assert torch.equal(shared, again)
    d = float((shared.float() - plain.float()).norm() / plain.float().norm())
    print(f"[property] {which} B={B} {hw}x{hw}: CFG halves sharing their prefix vs the plain call: rel-L2 {d:.2e} "
          f"(bit-equal: {bool(torch.equal(shared, plain))}), launches {n_shared} vs {n_plain}")
    assert d < 1.2e-2
    if which == "tiny":
        ref = M.unet_forward(sd, cfg, x.cpu(), torch.full((B,), 500), ctx.cpu())
        report("tiny unet, CFG halves sharing their prefix", shared.cpu(), ref, 3e-2)
    # per-sample timesteps in tensor form, equal between the halves, work as well
    t = torch.tensor([500, 20] * (B // 4) if B >= 4 else [500], device=DEV)
    tt = torch.cat([t, t])[:B] if B >= 4 else torch.full((B,), 500, device=DEV)
    p2 = net(x, tt, encoder_hidden_states=ctx).sample
    with cfg_pairs():
        s2 = net(x, tt, encoder_hidden_states=ctx).sample
    assert float((s2.float() - p2.float()).norm() / p2.float().norm()) < 1.2e-2
    # the hint is per call: a later plain call on other data is not affected
    y = randn(B, cfg.in_channels, hw, hw, seed=53).to(DEV)
    q1 = net(y, 500, encoder_hidden_states=ctx).sample
    assert not torch.equal(q1[: B // 2], q1[B // 2:])


def test_context_cache_holds_alternating_contexts():
    """GYRE_CTX_SLOTS entries: hires-fix leaves / CFGUNet_Sequential (reference unet/cfg.py:27-38,
    unet/hires_fix.py:123-235) alternate between contexts on every call - no re-projection after the first round, results
    bit-equal to a handle that only ever saw that context; a fifth context evicts the least recently used one; a weight
    update drops them all."""
    import ctypes as C
    from gyre_amd import _lib
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 16, 16, seed=21).to(DEV)
    t = torch.tensor([700, 30], device=DEV)
    ctxs = [randn(2, 77 if i != 2 else 154, cfg.cross_attention_dim, seed=40 + i).to(DEV) for i in range(5)]
    fresh = []
    for c in ctxs:
        other, _ = make_unet(cfg)
        fresh.append(other(x, t, encoder_hidden_states=c).sample)
    L = _lib.lib()
    calls = {"n": 0}
    orig = L.gyre_unet_set_context_slot

    class Counting:                                            # count the projections the shell asks for
        def __call__(self, *a):
            calls["n"] += 1
            return orig(*a)
    L.gyre_unet_set_context_slot = Counting()
    try:
        for rnd in range(3):
            for i in range(4):
                assert torch.equal(net(x, t, encoder_hidden_states=ctxs[i]).sample, fresh[i]), (rnd, i)
        assert calls["n"] == 4 and len(net._ctx_slots) == 4
        assert torch.equal(net(x, t, encoder_hidden_states=ctxs[4]).sample, fresh[4])       # evicts context 0 (LRU)
        assert calls["n"] == 5
        assert torch.equal(net(x, t, encoder_hidden_states=ctxs[3]).sample, fresh[3]) and calls["n"] == 5
        assert torch.equal(net(x, t, encoder_hidden_states=ctxs[0]).sample, fresh[0]) and calls["n"] == 6
        # selecting an empty / invalidated entry through the raw C ABI is an error, not stale data
        net.load_state_dict(sd)
        net(x, t, encoder_hidden_states=ctxs[1])
        assert calls["n"] == 7 and len(net._ctx_slots) == 1
        assert L.gyre_unet_select_context(C.c_void_p(net._handle), 3) == -1 and b"slot" in L.gyre_last_error()
        assert L.gyre_unet_select_context(C.c_void_p(net._handle), 7) == -1
    finally:
        L.gyre_unet_set_context_slot = orig


def test_lora_merge_on_native_unet():
    """LoRA folded into the weights (gyre_amd/lora.py): the native forward follows the merged weights (vs the fp32
    oracle run on the same merged state dict), and removing the LoRA restores the original output bit-exactly."""
    from gyre_amd import lora as LR
    from test_lora_host import kohya_lora
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x, t = randn(2, 4, 16, 16, seed=1), torch.tensor([700, 30])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    run = lambda: net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
    base = run()
    lora = {k: v * 3 for k, v in kohya_lora(net).items()}
    assert LR.apply_lora(net, lora, "a", scale=1.0) == 4
    merged_sd = {k: v.detach().cpu().float() for k, v in net.state_dict().items()}
    with_lora = run()
    ref = M.unet_forward(merged_sd, cfg, x, t, ctx)
    report("tiny unet + LoRA", with_lora, ref, 3e-2)
    assert float((with_lora - base).norm() / base.norm()) > 1e-2        # the LoRA really changed the function
    LR.remove_lora_from_model(net)
    assert torch.equal(run(), base)


@pytest.mark.parametrize("full", [False, True])
def test_unet_per_level_parity(full):
    """SURVEY 8(d) per-block gate (bf16 native vs fp32 oracle, rel-L2 <= 2e-2 per block): the activations at the end of
    every down level, the mid block and every up level, taken from one forward through gyre_unet_debug_tap, against
    the oracle's taps.  full=True is the SD1.5 architecture at 32x32 latents (CPU oracle stays in seconds)."""
    import ctypes as C
    from gyre_amd import _lib
    cfg = gcfg.sd15_unet() if full else gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    H = W = 32 if full else 16
    x, t = randn(2, 4, H, W, seed=1), torch.tensor([901, 77])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    taps = {}
    ref = M.unet_forward(sd, cfg, x, t, ctx, taps=taps)
    net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV))          # uploads weights, creates the handle
    n = len(cfg.block_out_channels)
    names = [f"down{i}" for i in range(n)] + ["mid"] + [f"up{i}" for i in range(n)]
    bufs = {k: torch.empty(taps[k].shape, dtype=torch.float32, device=DEV) for k in names}
    L = _lib.lib()
    for k, b in bufs.items():
        _lib.check(L.gyre_unet_debug_tap(C.c_void_p(net._handle), k.encode(), C.c_void_p(b.data_ptr()), b.numel() * 4))
    out = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    torch.cuda.synchronize()
    for k in names:
        # per-level gate: 2e-2 with bf16 storage; 5e-3 with fp16 storage (the fp16 flavour's run of this file)
        report(f"{'sd15' if full else 'tiny'} unet level {k} {tuple(taps[k].shape)}", bufs[k].cpu(), taps[k], 5e-3 if HDT == torch.float16 else 2e-2)
    report("unet eps", out.cpu(), ref, 5e-3 if HDT == torch.float16 else 3e-2)
    # a too-small tap buffer is an error, and taps do not persist into the following forward
    small = torch.empty(4, device=DEV)
    _lib.check(L.gyre_unet_debug_tap(C.c_void_p(net._handle), b"mid", C.c_void_p(small.data_ptr()), 16))
    with pytest.raises(ValueError):
        net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV))


@pytest.mark.parametrize("r", [8, 40, 1000])
def test_tiny_unet_tome_parity(r):
    """ToMe in the native UNet (reference option "tome: r", nonfree/tome_unet.py:138-182) vs the oracle UNet with the
    restated algorithm (oracle/tome_ref.py, same tie rules and roundings).  r = 1000 exceeds every level's N / 2 and is
    clipped per layer; the result must differ from the unmerged UNet (the feature is really on)."""
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x, t, ctx = randn(2, 4, 16, 16, seed=61), torch.tensor([500, 40]), randn(2, 77, cfg.cross_attention_dim, seed=62)
    base = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
    net.set_tome(r)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
    ref = M.unet_forward(sd, cfg, x, t, ctx, tome_r=r)
    report(f"tiny unet + ToMe r={r}", got, ref, 3e-2)
    assert float((got - base).norm() / base.norm()) > 1e-3
    net.set_tome(0)
    assert torch.equal(net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu(), base)


def test_sd15_unet_tome_full_size_properties():
    """BASELINE config 5 shape (SD1.5, batch 16 = 8 images x CFG, 64x64 latents) with ToMe r = 1024: finite, deterministic,
    close to the unmerged prediction (merging redundant keys is an approximation of the same attention), and faster
    attention is reported by tools/, not asserted here."""
    import gpu_util
    cfg = gcfg.sd15_unet()
    net = GyreHipUNet(cfg).to(HDT).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, device=DEV, generator=g, dtype=torch.float32) / p[0].numel() ** 0.5)
            elif "norm" in k and k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    net._invalidate()
    x = torch.randn(16, 4, 64, 64, device=DEV, generator=g)
    ctx = torch.randn(16, 77, 768, device=DEV, generator=g)
    t = torch.full((16,), 500, device=DEV)
    base = net(x, t, encoder_hidden_states=ctx).sample
    net.set_tome(1024)
    a = net(x, t, encoder_hidden_states=ctx).sample
    b = net(x, t, encoder_hidden_states=ctx).sample
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    rel = float((a - base).norm() / base.norm())
    print(f"[property] SD1.5 UNet with ToMe r=1024 vs without: rel-L2 {rel:.3e}")
    assert 1e-4 < rel < 0.5


def test_tiny_unet_controlnet_residual_injection():
    """`down_block_additional_residuals` / `mid_block_additional_residual` (reference unet/core.py:40-64; semantics of the
    in-tree patcher controlnet/unet_patcher.py:30-95) on the native path vs the oracle"""
    cfg = gcfg.tiny_unet()
    net, sd = make_unet(cfg)
    x = randn(2, 4, 16, 24, seed=1)
    t = torch.tensor([981, 17])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=2)
    taps = {}
    plain = M.unet_forward(sd, cfg, x, t, ctx, taps=taps)
    # skip shapes in production order: conv_in, then per level layers_per_block resnets (+ the downsampler except at the last)
    shapes, hh, ww = [(cfg.block_out_channels[0], 16, 24)], 16, 24
    for i, c in enumerate(cfg.block_out_channels):
        shapes += [(c, hh, ww)] * cfg.layers_per_block
        if i < len(cfg.block_out_channels) - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            shapes.append((c, hh, ww))
    down = [randn(2, *s, seed=10 + k) * 0.5 for k, s in enumerate(shapes)]
    mid = randn(2, cfg.block_out_channels[-1], hh, ww, seed=99) * 0.5
    ref = M.unet_forward(sd, cfg, x, t, ctx, down_res=down, mid_res=mid)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), down_block_additional_residuals=[d.to(DEV) for d in down],
              mid_block_additional_residual=mid.to(DEV)).sample
    report("tiny unet + ControlNet residuals", got.cpu(), ref, 3e-2)
    assert float((ref - plain).norm() / plain.norm()) > 0.05              # the residuals matter
    only_mid = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), mid_block_additional_residual=mid.to(DEV, torch.bfloat16)).sample
    report("tiny unet + mid residual only (bf16 residual)", only_mid.cpu(), M.unet_forward(sd, cfg, x, t, ctx, mid_res=mid), 3e-2)
    with pytest.raises(ValueError):
        net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), down_block_additional_residuals=[d.to(DEV) for d in down[:-1]])
    # T2I-adapter states: one per down level, added in place inside the down path (t2i_adapter/unet_patcher.py:21-86)
    ashapes, ah, aw = [], 16, 24
    for i, c in enumerate(cfg.block_out_channels):
        ashapes.append((c, ah, aw))
        if i < len(cfg.block_out_channels) - 1:
            ah, aw = (ah + 1) // 2, (aw + 1) // 2
    adapters = [randn(2, *s, seed=50 + k) * 0.5 for k, s in enumerate(ashapes)]
    ref = M.unet_forward(sd, cfg, x, t, ctx, adapter_states=adapters)
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), adapter_states=[a.to(DEV) for a in adapters]).sample
    report("tiny unet + T2I adapter states", got.cpu(), ref, 3e-2)
    assert float((ref - plain).norm() / plain.norm()) > 0.05
    both = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), adapter_states=[a.to(DEV) for a in adapters],
               down_block_additional_residuals=[d.to(DEV) for d in down], mid_block_additional_residual=mid.to(DEV)).sample
    report("tiny unet + adapters + ControlNet residuals", both.cpu(),
           M.unet_forward(sd, cfg, x, t, ctx, adapter_states=adapters, down_res=down, mid_res=mid), 3e-2)
    with pytest.raises(ValueError):
        net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), adapter_states=[a.to(DEV) for a in adapters[:-1]])
    with pytest.raises(NotImplementedError):
        net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV), adapter_states=[[adapters[0]]])


@pytest.mark.parametrize("mode", ["xy", "x", "y"])
def test_tiling_circular_convolutions_match_the_reference_patch(mode):
    """Request option `tiling` (reference unified_pipeline.py:1671-1712: every Conv2d's own padding becomes F.pad(mode="circular")
    along x and / or y): tiny UNet and VAE decode / encode against the oracle under the same patch, and the property the option
    exists for - rolling the input along a circular axis rolls the output (no seam), which zero padding does not give."""
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
    unet, vae = GyreHipUNet(ucfg), GyreHipVAE(vcfg)
    unet.load_state_dict(usd); vae.load_state_dict(vsd)
    unet, vae = unet.to(DEV), vae.to(DEV)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 4, 16, 24, generator=g)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g)
    t = torch.tensor([500, 500])
    plain = unet(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
    unet.set_tiling(mode); vae.set_tiling(mode)
    try:
        got = unet(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
        with M.tiling(mode):
            ref = M.unet_forward(usd, ucfg, x, t, ctx)
            zref = M.vae_decode(vsd, vcfg, x[:1, :, :8, :12])
        report(f"tiny UNet tiling={mode}", got, ref, 3e-2)
        assert rel_l2(got, plain) > 1e-3                          # the option does something
        img = vae.decode(x[:1, :, :8, :12].to(DEV)).sample.cpu()
        report(f"tiny VAE decode tiling={mode}", img, zref, 3e-2)
        # seamless: a circular shift of the latents along a wrapped axis (by a multiple of the network's total stride, 8) shifts
        # the result, up to bf16 rounding (the GroupNorm partial sums are taken in another order)
        sx, sy = (8 if mode != "y" else 0), (8 if mode != "x" else 0)
        rolled = unet(torch.roll(x, (sy, sx), (2, 3)).to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
        assert rel_l2(rolled, torch.roll(got, (sy, sx), (2, 3))) < 2.5e-2
    finally:
        unet.set_tiling(False); vae.set_tiling(False)
    back = unet(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.cpu()
    assert torch.equal(back, plain)


def test_sd_like_attention_logits_through_the_unet_redo_path_gives_the_checked_bits():
    """Round 6 (verdict item 3): the default self-attention pass is optimistic - rows centred on the first key tile, no per-tile
    check, a workgroup whose row sums leave (0, 2^60) repeats its tile with the checked pass - and every measurement so far ran on
    N(0, 1/fan_in) weights whose logits never come near that.  Here the logits are SD-like THROUGH THE UNET PATH: a D = 40 UNet
    (SD1.5's head dim at the 64x64 level) whose first-level self-attentions get to_k = to_q = 3 x the synthetic weights: off-diagonal
    scores ~ N(0, 9^2) natural units (row maxima 30 - 45 over 1024 keys), the diagonal |q_i|^2 / sqrt(D) ~ 57 +- 13 with a tail beyond
    88 - a continuum of excesses over the first tile's maximum, below, inside and above the band the old bound (1e25) accepted and
    the checked pass re-centred.  Asserts: workgroups did repeat their pass (the redo path ran inside gyre_unet_forward), the output
    is finite and BIT-equal to the always-checked pass (variant 7), and both sit at the usual distance from the fp32 oracle."""
    from gyre_amd.config import UNetConfig
    L = _lib.lib()
    cfg = UNetConfig(block_out_channels=(160, 160, 320, 320), num_heads=(4, 4, 8, 8), cross_attention_dim=64, sample_size=32)
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 0)
    for name in list(sd):
        if name.endswith("attn1.to_q.weight") and (name.startswith("down_blocks.0.") or name.startswith("up_blocks.3.")):
            sd[name] = sd[name] * 3.0
            sd[name.replace("to_q", "to_k")] = sd[name].clone()
    net = GyreHipUNet(cfg)
    net.load_state_dict(sd)
    net = net.to(DEV)
    x = randn(2, 4, 32, 32, seed=31)
    t = torch.tensor([700, 700])
    ctx = randn(2, 77, cfg.cross_attention_dim, seed=32)
    ref = M.unet_forward(sd, cfg, x, t, ctx)
    # what the logits of the first such attention look like (oracle tap-free estimate from the weights' construction is not needed:
    # measure them on the oracle's own hidden states through the public block functions would couple the test to oracle internals;
    # the redo counter below is the evidence that rows crossed the bound)
    c0 = L.gyre_debug_attn_redo_count()
    assert c0 >= 0
    got = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
    torch.cuda.synchronize()
    c1 = L.gyre_debug_attn_redo_count()
    old = L.gyre_debug_force_attn_variant(7)
    try:
        chk = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample
        torch.cuda.synchronize()
    finally:
        L.gyre_debug_force_attn_variant(old)
    c2 = L.gyre_debug_attn_redo_count()
    print(f"[redo] workgroups that repeated their pass: default {c1 - c0}, always-checked {c2 - c1}")
    assert c1 - c0 > 0, "the SD-like logits were meant to push rows over the acceptance bound"
    assert c2 == c1, "the always-checked pass never repeats"
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got, chk), "optimistic default differs from the always-checked pass"
    report("D=40 UNet with SD-like self-attention logits", got.cpu(), ref, 3e-2)


def test_fused_cross_attention_block_matches_the_three_launch_chain_and_the_oracle():
    """Round 6: at SD1.x's 64x64 level the cross-attention block (to_q with the folded LayerNorm, attention over the cached text keys,
    to_out + bias + residual, row statistics for the next LayerNorm) is ONE launch (kernels_xattn.hip) once the grid covers the chip
    (batch >= 8).  Same UNet call with the fused kernel and with the three-launch chain (tuning bit 13): two launches fewer per site,
    results equal up to the rounding of the softmax (one exact pass here, two 64-key tiles there) - the distance between any two
    tile plans - and both at the usual distance from the fp32 oracle (checked on two of the eight samples: the oracle runs on the
    CPU).  Smaller batches and longer texts keep the chain."""
    L = _lib.lib()
    cfg = gcfg.sd15_unet()
    net, sd = make_unet(cfg)
    B = 8
    x = randn(B, 4, 64, 64, seed=41)
    t = torch.tensor([321] * B)
    ctx = randn(B, 77, 768, seed=42)
    ref = M.unet_forward(sd, cfg, x[:2], t[:2], ctx[:2])

    def run(xx, tt, cc):
        outs, launches = {}, {}
        for bits in (0x2000, 0):                     # first calls: the per-handle weight copies (LayerNorm folds, blocked copies) are made here
            old = L.gyre_debug_gemm_ablation(bits)
            try:
                net(xx.to(DEV), tt.to(DEV), encoder_hidden_states=cc.to(DEV))
                outs[bits] = net(xx.to(DEV), tt.to(DEV), encoder_hidden_states=cc.to(DEV)).sample.float().cpu()
                launches[bits] = L.gyre_last_launch_count()
            finally:
                L.gyre_debug_gemm_ablation(old)
        return outs, launches
    outs, launches = run(x, t, ctx)
    sites = 5                                        # transformer blocks at the 64x64 level: down 2, up 3
    assert launches[0x2000] - launches[0] == 2 * sites, launches
    report("SD1.5 UNet batch 8, fused cross-attention (samples 0, 1)", outs[0][:2], ref, 3e-2)
    report("SD1.5 UNet batch 8, three-launch chain (samples 0, 1)", outs[0x2000][:2], ref, 3e-2)
    assert rel_l2(outs[0], outs[0x2000]) < 2.5e-2
    e_f, e_c = rel_l2(outs[0][:2], ref), rel_l2(outs[0x2000][:2], ref)
    assert e_f < 1.3 * e_c + 1e-3, (e_f, e_c)
    # outside the fused kernel's domain - a small batch (the grid would leave most CUs idle), a two-chunk text (154 keys) - the chain
    # runs either way: same launches, same bits
    for xx, tt, cc in ((x[:2], t[:2], ctx[:2]), (x, t, randn(B, 154, 768, seed=43))):
        o2, n2 = run(xx, tt, cc)
        assert n2[0] == n2[0x2000] and torch.equal(o2[0], o2[0x2000])


def test_resnet_shortcut_folded_into_conv2_matches_the_two_launches():
    """Round 6: where a resnet changes its channel count (or reads a skip concatenation) its 1x1 shortcut rides inside conv2 as extra K
    steps of the pipelined convolution tile (GemmParams::sc_*) instead of being a launch of its own whose result conv2 re-reads as
    residual.  Same UNet call both ways (tuning bit 7 = two launches): fewer launches, results equal up to one rounding per output
    (the folded form rounds once, the two launches twice), both at the usual distance from the oracle; batch-invariant planning
    keeps the two launches (bit-exact batch splits: tests/test_gpu_configs.py)."""
    L = _lib.lib()
    cfg = gcfg.sd15_unet()
    net, sd = make_unet(cfg)
    B = 8
    x = randn(B, 4, 64, 64, seed=51)
    t = torch.tensor([77] * B)
    ctx = randn(B, 77, 768, seed=52)
    ref = M.unet_forward(sd, cfg, x[:2], t[:2], ctx[:2])
    outs, launches = {}, {}
    for bits in (0x80, 0):
        old = L.gyre_debug_gemm_ablation(bits)
        try:
            net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV))          # (first call: per-handle weight copies)
            outs[bits] = net(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.float().cpu()
            launches[bits] = L.gyre_last_launch_count()
        finally:
            L.gyre_debug_gemm_ablation(old)
    print(f"[fold] launches {launches[0x80]} -> {launches[0]}")
    assert launches[0] < launches[0x80], launches              # at least the 64x64 / 32x32 shortcuts fold at this batch
    report("SD1.5 UNet batch 8, shortcuts folded (samples 0, 1)", outs[0][:2], ref, 3e-2)
    report("SD1.5 UNet batch 8, shortcuts as launches (samples 0, 1)", outs[0x80][:2], ref, 3e-2)
    assert rel_l2(outs[0], outs[0x80]) < 2.5e-2
    assert rel_l2(outs[0][:2], ref) < 1.2 * rel_l2(outs[0x80][:2], ref) + 1e-3
