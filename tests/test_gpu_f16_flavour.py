"""The fp16-storage flavour of the library (libgyre_hip_f16.so: the same sources built with -DGYRE_STORE_F16, csrc/common.h) - the
reference's own GPU arithmetic (gyre/manager.py:146-151,1199-1200 loads its pipelines in torch.float16).

Every raw-operator, model-parity, input-gradient and full-length test of the suite is written against `gpu_util.HDT`, the 16-bit
storage dtype of the flavour under test; this file re-runs those files in a child process with GYRE_STORAGE=f16, where HDT is
torch.float16, `_lib.lib()` is the fp16 library and float32-parameter modules route to it too.  Same oracle, same tolerances (fp16
storage has three more mantissa bits than bf16, so every bf16 tolerance holds with room); the full-length c1 run is held to the
>= 50 dB SURVEY 8(d) names as the fp32-grade figure (tests/test_gpu_full_runs.py).  In-process here: the two flavours side by side
(one process, two libraries, two handles) and the dtype -> library routing of the module shells."""
import os
import subprocess
import sys

import pytest
import torch

from gyre_amd import _lib, config as gcfg, weights
from gyre_amd.modules import GyreHipUNet
from gpu_util import DEV, randn, rel_l2
from oracle import models_ref as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_kernels.py", "tests/test_gpu_gemm_ar.py", "tests/test_gpu_gemm_sm.py", "tests/test_gpu_models.py",
         "tests/test_gpu_properties.py", "tests/test_gpu_vjp.py", "tests/test_gpu_configs.py", "tests/test_gpu_full_runs.py"]


@pytest.mark.skipif(_lib.default_storage() == _lib.F16, reason="already inside the fp16 run")
def test_operator_model_and_full_run_suites_pass_on_the_fp16_flavour():
    env = dict(os.environ, GYRE_STORAGE="f16")
    r = subprocess.run([sys.executable, "-m", "pytest", *FILES, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=3000)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    print(tail)
    assert r.returncode == 0, f"fp16 flavour failed:\n{tail}\n{r.stderr[-2000:]}"


def test_both_flavours_in_one_process_and_dtype_routing():
    """float16 parameters -> libgyre_hip_f16.so, bfloat16 -> libgyre_hip.so; both handles live side by side (the libraries export the
    same symbol names: -Bsymbolic + local dlopen scope), each destroyed by the library that made it; .to(other dtype) moves a
    module to the other library.  The fp16 result must sit CLOSER to the fp32 oracle than the bf16 one."""
    assert _lib.lib(_lib.BF16).gyre_storage_dtype() == _lib.BF16 and _lib.lib(_lib.F16).gyre_storage_dtype() == _lib.F16
    cfg = gcfg.tiny_unet()
    sd = weights.synthetic_state_dict(weights.unet_param_shapes(cfg), 0)
    nets = {}
    for dt in (torch.bfloat16, torch.float16):
        n = GyreHipUNet(cfg)
        n.load_state_dict(sd)
        nets[dt] = n.to(dt).to(DEV)
    x, t, ctx = randn(2, 4, 16, 16, seed=1), torch.tensor([981, 17]), randn(2, 77, cfg.cross_attention_dim, seed=2)
    ref = M.unet_forward(sd, cfg, x, t, ctx)
    out = {dt: n(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.float().cpu() for dt, n in nets.items()}
    assert nets[torch.float16]._handle_storage == _lib.F16 and nets[torch.bfloat16]._handle_storage == _lib.BF16
    e_b, e_h = rel_l2(out[torch.bfloat16], ref), rel_l2(out[torch.float16], ref)
    print(f"[parity] tiny UNet vs fp32 oracle: bf16 storage {e_b:.3e}, fp16 storage {e_h:.3e}")
    assert e_b < 3e-2 and e_h < 5e-3 and e_h < 0.5 * e_b
    # interleaved calls keep their own results (no cross-library state)
    again = nets[torch.bfloat16](x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.float().cpu()
    assert torch.equal(again, out[torch.bfloat16])
    # .to(dtype) re-homes the module
    moved = nets[torch.bfloat16].to(torch.float16)
    got = moved(x.to(DEV), t.to(DEV), encoder_hidden_states=ctx.to(DEV)).sample.float().cpu()
    assert moved._handle_storage == _lib.F16
    assert rel_l2(got, ref) < 3e-2            # (its parameters went through bf16 on the way: bf16-sized weight rounding stays)
