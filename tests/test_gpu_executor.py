"""One request over several device slots inside ONE process (gyre_amd/executor.py; -m gpu).  The box has one GPU, so the
"slots" are two (three) replicas on cuda:0 - two native handles, two host threads, two HIP streams, exactly what the
in-server sharding runs per device, minus the peer copy.  Reference split rule: batched_seeds, services/generate.py:977-990."""
import copy

import pytest
import torch

from gyre_amd import config as gcfg, weights
from gyre_amd.executor import DeviceSlotExecutor
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gyre_amd.pipeline import GyrePipeline
from gpu_util import HDT, DEV

pytestmark = pytest.mark.gpu


def _pipe(ucfg, vcfg):
    unet, vae = GyreHipUNet(ucfg), GyreHipVAE(vcfg)
    unet.load_state_dict(weights.synthetic_state_dict(weights.unet_param_shapes(ucfg)))
    vae.load_state_dict(weights.synthetic_state_dict(weights.vae_param_shapes(vcfg)))
    return GyrePipeline(unet.to(HDT).to(DEV), vae.to(HDT).to(DEV), device=DEV)


def test_module_copies_get_their_own_native_handle():
    pipe = _pipe(gcfg.tiny_unet(), gcfg.tiny_vae())
    x = torch.randn(1, 4, 16, 16, device=DEV)
    ctx = torch.randn(1, 77, pipe.unet.config.cross_attention_dim, device=DEV)
    a = pipe.unet(x, 10, encoder_hidden_states=ctx).sample
    twin = copy.deepcopy(pipe.unet)
    assert twin._handle is None
    b = twin(x, 10, encoder_hidden_states=ctx).sample
    assert twin._handle is not None and twin._handle != pipe.unet._handle and torch.equal(a, b)
    del twin                                         # destroying the copy leaves the original usable
    assert torch.equal(pipe.unet(x, 10, encoder_hidden_states=ctx).sample, a)


@pytest.mark.parametrize("slots,B", [(2, 4), (3, 5), (2, 1)])
def test_request_split_over_device_slots_equals_the_single_slot_run(slots, B):
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    pipe = _pipe(ucfg, vcfg)
    g = torch.Generator().manual_seed(3)
    text = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g)
    unc = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g).expand(B, -1, -1).contiguous()
    seeds = [420420420 + i for i in range(B)]
    kw = dict(text_embeddings=text, uncond_embeddings=unc, height=128, width=128, num_inference_steps=6, sampler="euler_a")
    whole = pipe(seeds=seeds, output_type="latent", **kw)
    ex = DeviceSlotExecutor.replicate(pipe, [DEV] * slots)
    assert ex.world == slots and len({id(p.unet) for p in ex.pipelines}) == slots
    split = ex(seeds=seeds, **kw)
    assert split.shape == whole.shape and ex.last_unet_evals == pipe.last_unet_evals
    # tiny model: no split-K anywhere, so any split is bit-identical already
    assert torch.equal(split, whole)
    img_a, img_b = pipe.vae_decode(whole), ex.decode(split)
    assert torch.equal(img_a, img_b)


def test_full_size_request_over_two_slots_bit_exact_mode():
    """SD1.5 topology, 4 images split 2 + 2 over two handles vs the 4-image call on one: bit-identical with bit_exact=True
    (batch-invariant split-K planning, set per worker thread), equal to bf16 rounding without."""
    from gyre_amd.modules import set_batch_invariant
    ucfg, vcfg = gcfg.sd15_unet(), gcfg.sd15_vae()
    pipe = _pipe(ucfg, vcfg)
    g = torch.Generator().manual_seed(4)
    text = torch.randn(4, 77, 768, generator=g)
    unc = torch.randn(1, 77, 768, generator=g).expand(4, -1, -1).contiguous()
    seeds = [11, 12, 13, 14]
    kw = dict(text_embeddings=text, uncond_embeddings=unc, height=512, width=512, num_inference_steps=3, sampler="dpmpp_2m")
    prev = set_batch_invariant(16)
    try:
        whole = pipe(seeds=seeds, output_type="latent", **kw)
    finally:
        set_batch_invariant(prev)
    ex = DeviceSlotExecutor.replicate(pipe, [DEV, DEV])
    split = ex(seeds=seeds, bit_exact=True, **kw)
    assert torch.equal(split, whole)
    loose = ex(seeds=seeds, **kw)
    d = float((loose - whole).norm() / whole.norm())
    print(f"[property] 2 + 2 split vs one call of 4, default planning: latent rel-L2 {d:.2e}")
    assert d < 5e-2


def test_engine_option_shard_devices_fans_one_request():
    import functools
    from test_gpu_engine import build_engine, generators, sample_dpmpp_2m, wrapper_kwargs
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    _, _, eng = build_engine(ucfg, vcfg)
    eng.scheduler = functools.partial(sample_dpmpp_2m, warmup_lms=True, ddim_cutoff=0.1)
    prompt, seeds = ["a", "b", "c"], [5, 6, 7]
    kw = dict(prompt=prompt, width=128, height=128, num_inference_steps=5)
    one, nsfw = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    eng.set_options({"shard_devices": [DEV, DEV]})
    two, nsfw2 = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    assert eng._executor is not None and eng._executor.world == 2
    assert torch.equal(one, two) and nsfw == nsfw2 == [False] * 3
    eng.set_options({"shard_devices": []})
    again, _ = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    assert torch.equal(one, again)


def test_replicas_follow_weight_changes_of_the_engine_modules():
    """The replica cache is keyed on the source modules' identity AND weight version: after load_state_dict (or .to / .half /
    a LoRA merge) on the engine's UNet the other slots must not keep generating with the old weights."""
    import functools
    from test_gpu_engine import build_engine, generators, sample_dpmpp_2m, wrapper_kwargs
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    _, _, eng = build_engine(ucfg, vcfg)
    eng.scheduler = functools.partial(sample_dpmpp_2m, warmup_lms=True, ddim_cutoff=0.1)
    prompt, seeds = ["a", "b", "c"], [5, 6, 7]
    kw = dict(prompt=prompt, width=128, height=128, num_inference_steps=5)
    eng.set_options({"shard_devices": ["cuda", DEV]})           # 'cuda' and 'cuda:0' are one slot: no clone of the source
    first, _ = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    ex1 = eng._executor
    assert ex1.pipelines[0].unet is eng.unet and bool(torch.isfinite(first).all())
    again, _ = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    assert eng._executor is ex1 and torch.equal(first, again)    # unchanged weights: the replicas are reused
    sd = {k: v * 1.05 for k, v in eng.unet.state_dict().items()}
    eng.unet.load_state_dict(sd)
    sharded, _ = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    assert eng._executor is not ex1                              # rebuilt from the new weights
    eng.set_options({"shard_devices": []})
    single, _ = eng(**wrapper_kwargs(generator=generators(seeds), **kw))
    assert torch.equal(sharded, single) and not torch.equal(sharded, first)
