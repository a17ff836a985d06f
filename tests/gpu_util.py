"""Helpers for the -m gpu parity tests: call libgyre_hip through its C ABI on torch-allocated
device buffers and compare with fp32 ATen references."""
import ctypes as C

import torch

from gyre_amd import _lib

DEV = "cuda:0"
# the 16-bit storage dtype of the library flavour under test: bfloat16, or float16 when the suite runs with GYRE_STORAGE=f16
# (tests/test_gpu_f16_flavour.py re-runs the operator / model parity files that way)
HDT = _lib.STORAGE_TORCH_DTYPE[_lib.default_storage()]


_KEEP = []  # device tensors whose raw pointers were handed to the library; cleared (after a sync) per test


def vp(t):
    if t is None:
        return None
    _KEEP.append(t)  # the C ABI sees raw pointers only: keep the tensor alive until the kernel has run
    return C.c_void_p(t.data_ptr())


def release_kept():
    torch.cuda.synchronize()
    _KEEP.clear()


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, got, ref, tol):
    e = rel_l2(got, ref)
    d = (got.double().cpu() - ref.double().cpu()).abs()
    print(f"[parity] {name}: rel_l2={e:.3e} max_abs={float(d.max()):.3e} ref_absmax={float(ref.abs().max()):.3e} tol={tol}")
    if not e <= tol:
        idx = int(d.flatten().argmax())
        print(f"   worst at flat index {idx}: got {float(got.flatten()[idx]):.5f} ref {float(ref.flatten()[idx]):.5f}")
        # error distribution along the last axis helps to spot layout bugs
        if d.ndim >= 2:
            print("   per-last-axis mean abs err (first 16):", d.reshape(-1, d.shape[-1]).mean(0)[:16].tolist())
    assert e <= tol, f"{name}: rel_l2 {e:.3e} > {tol}"


def bf16_round(t):
    """Round to the library's 16-bit storage dtype (HDT: bf16, or fp16 for the fp16 flavour) and back."""
    return t.to(HDT).to(torch.float32)


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def repack_conv(w_oihw: torch.Tensor, cin_pad=None) -> torch.Tensor:
    L = _lib.lib()
    O, I, KH, KW = w_oihw.shape
    cin_pad = cin_pad or (I + 7) // 8 * 8
    src = w_oihw.float().contiguous().to(DEV)
    out = torch.zeros(O * KH * KW * cin_pad, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_repack_conv_weight(st(), vp(src), O, I, KH, KW, cin_pad, vp(out)))
    return out


def repack_linear(w: torch.Tensor, geglu=False) -> torch.Tensor:
    L = _lib.lib()
    O, I = w.shape
    src = w.float().contiguous().to(DEV)
    out = torch.zeros(O * I, dtype=HDT, device=DEV)
    _lib.check(L.gyre_op_repack_linear_weight(st(), vp(src), O, I, int(geglu), vp(out)))
    return out


def repack_bias(b: torch.Tensor, geglu=False) -> torch.Tensor:
    L = _lib.lib()
    src = b.float().contiguous().to(DEV)
    out = torch.zeros(b.numel(), dtype=torch.float32, device=DEV)
    _lib.check(L.gyre_op_repack_bias(st(), vp(src), b.numel(), int(geglu), vp(out)))
    return out
