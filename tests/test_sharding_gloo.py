"""N>1 path on CPU: 2 processes, gloo backend, oracle-backed tiny models.  Checks that the sharded
generation + all_gather reproduces the single-process batch (the collective is the same code RCCL runs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gyre_amd.sharding import shard_bounds


def test_shard_bounds():
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]
    with pytest.raises(ValueError):
        shard_bounds(4, 0)
    for total in range(0, 20):
        for world in (1, 2, 3, 8):
            b = shard_bounds(total, world)
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_host_pipeline import OracleUNet, OracleVAE
        from gyre_amd import config as gcfg, weights
        from gyre_amd.pipeline import GyrePipeline
        from gyre_amd.sharding import generate_sharded
        ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
        usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
        vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
        g = torch.Generator().manual_seed(5)
        text = torch.randn(total, 77, ucfg.cross_attention_dim, generator=g)
        unc = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
        pipe = GyrePipeline(OracleUNet(usd, ucfg), OracleVAE(vsd, vcfg), device="cpu")
        seeds = [100 + i for i in range(total)]
        kw = dict(height=128, width=128, num_inference_steps=3, sampler="euler_a", guidance_scale=5.0)
        full, (s, e) = generate_sharded(pipe, seeds=seeds, text_embeddings=text, uncond_embeddings=unc, **kw)
        if rank == 0:
            ref = pipe(seeds=seeds, text_embeddings=text, uncond_embeddings=unc, output_type="latent", **kw)
            q.put((tuple(full.shape), bool(torch.allclose(full, ref, rtol=1e-4, atol=1e-3)), (s, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [3, 4, 1])
def test_sharded_generation_matches_single_process(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    shape, ok, bounds = q.get(timeout=10)
    assert shape == (total, 4, 16, 16) and ok
    assert bounds == shard_bounds(total, 2)[0]
