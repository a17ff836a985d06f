"""Worker of tests/test_gpu_sharding_nccl.py: two ranks, BOTH on cuda:0, one request sharded with generate_sharded over
the "nccl" backend (= RCCL on ROCm).  Prints RESULT_OK / RCCL_REFUSED:<why> on rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        probe = torch.ones(4, device=dev) * (rank + 1)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert float(probe[0]) == 3.0
    except Exception as e:  # noqa: BLE001 - RCCL refuses two ranks on one device on some builds
        if rank == 0:
            print("RCCL_REFUSED:" + str(e).replace("\n", " ")[:300], flush=True)
        return 0
    from gyre_amd import config as gcfg, weights
    from gyre_amd.modules import GyreHipUNet, GyreHipVAE
    from gyre_amd.pipeline import GyrePipeline
    from gyre_amd.sharding import generate_sharded, shard_bounds
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    unet, vae = GyreHipUNet(ucfg), GyreHipVAE(vcfg)
    unet.load_state_dict(weights.synthetic_state_dict(weights.unet_param_shapes(ucfg)))
    vae.load_state_dict(weights.synthetic_state_dict(weights.vae_param_shapes(vcfg)))
    pipe = GyrePipeline(unet.to(dev), vae.to(dev), device=dev)
    g = torch.Generator().manual_seed(5)
    text = torch.randn(3, 77, ucfg.cross_attention_dim, generator=g)
    unc = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g).expand(3, -1, -1).contiguous()
    kw = dict(height=128, width=128, num_inference_steps=5, sampler="euler_a")
    seeds = [11, 12, 13]
    full, (s, e) = generate_sharded(pipe, seeds=seeds, text_embeddings=text, uncond_embeddings=unc, bit_exact=True, **kw)
    assert (s, e) == shard_bounds(3, world)[rank] and full.shape == (3, 4, 16, 16)
    if rank == 0:
        from gyre_amd.modules import set_batch_invariant
        prev = set_batch_invariant(16)
        single = pipe(seeds=seeds, text_embeddings=text, uncond_embeddings=unc, output_type="latent", **kw)
        set_batch_invariant(prev)
        ok = bool(torch.equal(single, full))
        print("RESULT_OK" if ok else f"RESULT_MISMATCH max abs {float((single - full).abs().max()):.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
