"""Thread-concurrency of the C ABI (-m gpu).  include/gyre_hip.h promises "different handles may be used concurrently
from different threads"; the reference depends on it: its gRPC ThreadPoolExecutor(max_workers=4) (server.py:154) hands
every request thread one device queue slot for the whole request (manager.py:2107-2139), so two pipeline clones run from
two threads of the same process at the same time.

Here: two (three) handles on cuda:0, one host thread and one HIP stream each, started together behind a barrier and
looping over calls so that their launches interleave on the device; every result must be BIT-equal to the same calls made
serially.  Covers the thread-local pieces of the library (error string, launch counter, profiler state, debug knobs), the
shared lazily-created ones (attention zero page, kernel attribute caches) and the per-handle ones (workspace planner,
context K/V cache, folded-LayerNorm weight copies)."""
import threading

import pytest
import torch

from gyre_amd import config as gcfg, weights
from gyre_amd.modules import GyreHipUNet, GyreHipVAE
from gpu_util import HDT, DEV

pytestmark = pytest.mark.gpu


def _unet(cfg, seed):
    net = GyreHipUNet(cfg)
    net.load_state_dict(weights.synthetic_state_dict(weights.unet_param_shapes(cfg), seed))
    return net.to(HDT).to(DEV)


def _inputs(cfg, B, hw, seed, S=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg.in_channels, hw, hw, generator=g).to(DEV)
    ctx = torch.randn(B, S, cfg.cross_attention_dim, generator=g).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    return x, t, ctx


def _run_threads(jobs, iters):
    """jobs: list of callables f(i) -> tensor.  Serial reference first, then all jobs concurrently (one thread + stream
    each).  Returns (serial, threaded) result lists."""
    serial = [[f(i).clone() for i in range(iters)] for f in jobs]
    torch.cuda.synchronize()
    out = [[None] * iters for _ in jobs]
    errs = []
    gate = threading.Barrier(len(jobs))

    def worker(k):
        try:
            torch.cuda.set_device(torch.device(DEV))
            stream = torch.cuda.Stream(device=DEV)
            gate.wait()
            with torch.cuda.stream(stream):
                for i in range(iters):
                    out[k][i] = jobs[k](i).clone()
                stream.synchronize()
        except BaseException as e:  # noqa: BLE001 - reported in the main thread
            errs.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    return serial, out


def test_two_unet_handles_from_two_threads_are_bit_equal_to_serial():
    cfg = gcfg.sd15_unet()
    nets = [_unet(cfg, 0), _unet(cfg, 1)]                  # two clones with different weights: a mix-up would show
    ins = [[_inputs(cfg, 2, 64, 10 * k + i) for i in range(2)] for k in range(2)]
    iters = 6

    def job(k):
        def f(i):
            x, t, ctx = ins[k][i % 2]                       # alternating contexts: the K/V cache is refreshed under load
            return nets[k](x, t, encoder_hidden_states=ctx).sample
        return f
    serial, threaded = _run_threads([job(0), job(1)], iters)
    for k in range(2):
        for i in range(iters):
            assert torch.equal(serial[k][i], threaded[k][i]), f"handle {k}, call {i}: differs from the serial run"
    assert not torch.equal(serial[0][0], serial[1][0])


def test_unet_vae_and_error_paths_concurrently():
    """Three threads: a UNet handle, a VAE handle (decode + encode), and a thread that keeps provoking library errors -
    the thread-local error string / status of one thread must never leak into the others' calls."""
    ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
    unet = _unet(ucfg, 0)
    vae = GyreHipVAE(vcfg)
    vae.load_state_dict(weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 0))
    vae = vae.to(HDT).to(DEV)
    bad = _unet(ucfg, 2)
    x, t, ctx = _inputs(ucfg, 2, 16, 3)
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4)).to(DEV)
    x3 = x[:, :3].contiguous()                              # wrong channel count -> ValueError (tests/test_gpu_models.py)
    seen = []

    def f_unet(i):
        return unet(x, t, encoder_hidden_states=ctx).sample

    def f_vae(i):
        img = vae.decode(z).sample
        return vae.encode(img.clamp(-1, 1)).latent_dist.parameters

    def f_err(i):
        try:
            bad(x3, t, encoder_hidden_states=ctx)
            seen.append("no error")
        except ValueError as e:
            seen.append(str(e))
        return bad(x, t, encoder_hidden_states=ctx).sample       # and the handle keeps working afterwards
    serial, threaded = _run_threads([f_unet, f_vae, f_err], 12)
    for k in range(3):
        for i in range(12):
            assert torch.equal(serial[k][i], threaded[k][i]), f"job {k}, call {i}"
    assert "no error" not in seen and len(seen) == 24 and len(set(seen)) == 1, set(seen)   # always ITS OWN error text
