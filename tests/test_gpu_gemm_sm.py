"""The small-problem GEMM kernel (tile config 32, kernels_gemm_sm.hip; tuning bit 5 = off) against the register-staged 4-wave kernel
it replaces, on the shapes it takes over: deep-level C x C projections at small batch, text-context K / V projections, ragged M.
Same K order on one accumulator, so the outputs have to agree bit for bit.  Replaces the cuBLAS GEMMs behind torch.nn.Linear in
the UNet the reference calls at gyre/pipeline/unet/core.py:274."""
import math

import pytest
import torch

from gpu_util import DEV, randn, repack_bias, repack_linear, st, vp
from gyre_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,res,bias", [(2048, 640, 640, True, True), (512, 1280, 1280, False, True), (154, 768, 320, False, False),
                                            (77, 768, 1280, False, False), (4096 + 8, 320, 320, True, True), (1, 64, 64, False, True),
                                            (63, 128, 192, True, False), (1024, 2560, 640, True, True), (512, 5120, 1280, True, True)])
def test_small_problem_kernel_matches_the_4_wave_kernel(M, K, N, res, bias):
    L = _lib.lib()
    x = (randn(M, K, seed=1) * 1.3).to(torch.bfloat16).to(DEV)
    w0, b0 = randn(N, K, seed=2) / math.sqrt(K), randn(N, seed=3) * 0.3
    w, b = repack_linear(w0), (repack_bias(b0) if bias else None)
    r = randn(M, N, seed=4).to(torch.bfloat16).to(DEV) if res else None
    outs, names = [], []
    for bits in (0x20, 0):
        L.gyre_debug_gemm_ablation(bits)
        try:
            _lib.prof_enable(None)
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            _lib.check(L.gyre_op_linear(st(), vp(x), M, K, vp(w), N, vp(b), vp(r), 0, vp(y)))
            torch.cuda.synchronize()
            names.append(set(_lib.prof_collect()))
            outs.append(y)
        finally:
            _lib.prof_enable([])
            L.gyre_debug_gemm_ablation(0)
    if "k_gemm_sm" not in names[1]:
        pytest.skip(f"the planner keeps a larger tile for this shape: {names[1]}")
    assert "k_gemm_sm" not in names[0]
    assert torch.isfinite(outs[1].float()).all()
    if "k_splitk_reduce" in names[0]:        # the old path cut K into slices (fp32 slabs): same sums in another order
        assert (outs[0].float() - outs[1].float()).abs().max().item() < 0.0625
    else:
        assert torch.equal(outs[0], outs[1])
    ref = x.float() @ w0.to(torch.bfloat16).float().to(DEV).T + (b0.to(DEV) if bias else 0) + (r.float() if res else 0)
    assert (outs[1].float() - ref).abs().max().item() < 0.08
