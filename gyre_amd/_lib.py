"""ctypes binding of libgyre_hip.so (C ABI: include/gyre_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, every entry
point raises.  A silent PyTorch/CPU path would void the parity claims.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must be imported first: maps torch's libamdhip64.so.7, which our .so then shares)

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 8

F32, BF16, F16 = 0, 1, 2
_TORCH_DTYPE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
# One library per 16-bit STORAGE type, same sources and same ABI (csrc/common.h, gyre_storage_dtype()): bf16 (default) and fp16 - the
# reference's own GPU arithmetic (manager.py:146-151 loads fp16).  A module picks its library by the dtype of its parameters
# (modules._NativeModule._storage): float16 -> F16, bfloat16 -> BF16, float32 -> the process default, which is BF16 unless
# GYRE_STORAGE=f16 is set (how the test suite runs every raw-operator test on the fp16 flavour) or set_default_storage() is called.
# (GYRE_HIP_LIB / GYRE_HIP_LIB_F16: tuning / reproducer builds of the same ABI.)
LIB_PATHS = {BF16: os.environ.get("GYRE_HIP_LIB") or os.path.join(_HERE, "libgyre_hip.so"),
             F16: os.environ.get("GYRE_HIP_LIB_F16") or os.path.join(_HERE, "libgyre_hip_f16.so")}
LIB_PATH = LIB_PATHS[BF16]
_default_storage = F16 if os.environ.get("GYRE_STORAGE", "").lower() in ("f16", "fp16", "float16", "half") else BF16
STORAGE_TORCH_DTYPE = {BF16: torch.bfloat16, F16: torch.float16}


def default_storage() -> int:
    return _default_storage


def set_default_storage(storage: int) -> int:
    """Process-wide default storage flavour (BF16 / F16) for float32-parameter modules, lib() without an argument and the
    profiler helpers.  Returns the previous value."""
    global _default_storage
    if storage not in LIB_PATHS:
        raise ValueError("storage must be _lib.BF16 or _lib.F16")
    old, _default_storage = _default_storage, storage
    return old


def storage_for(dtype: torch.dtype) -> int:
    return F16 if dtype == torch.float16 else BF16 if dtype == torch.bfloat16 else _default_storage


class GyreError(RuntimeError):
    pass


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("n_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * MAX_LEVELS), ("layers_per_block", C.c_int32),
                ("attn_levels", C.c_int32 * MAX_LEVELS), ("num_heads", C.c_int32 * MAX_LEVELS),
                ("transformer_depth", C.c_int32 * MAX_LEVELS), ("cross_attention_dim", C.c_int32),
                ("norm_num_groups", C.c_int32), ("use_linear_projection", C.c_int32),
                ("flip_sin_to_cos", C.c_int32), ("freq_shift", C.c_float)]


class VAECfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("latent_channels", C.c_int32),
                ("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * MAX_LEVELS),
                ("layers_per_block", C.c_int32), ("norm_num_groups", C.c_int32)]


_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
_SIGS = {
    "gyre_abi_version": (C.c_int, []),
    "gyre_storage_dtype": (C.c_int, []),
    "gyre_last_error": (C.c_char_p, []),
    "gyre_last_launch_count": (C.c_int64, []),
    "gyre_unet_create": (_i, [C.POINTER(UNetCfg), _i, C.POINTER(_vp)]),
    "gyre_unet_destroy": (None, [_vp]),
    "gyre_unet_num_params": (_i, [_vp]),
    "gyre_unet_param_key": (C.c_char_p, [_vp, _i]),
    "gyre_unet_set_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i, _vp]),
    "gyre_unet_finalize": (_i, [_vp, _vp]),
    "gyre_unet_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "gyre_unet_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i]),
    "gyre_unet_set_context": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "gyre_unet_set_context_slot": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "gyre_unet_select_context": (_i, [_vp, _i]),
    "gyre_unet_hint_uniform_timestep": (_i, [_vp, _i]),
    "gyre_unet_hint_cfg_pairs": (_i, [_vp, _i]),
    "gyre_unet_set_tome": (_i, [_vp, _i]),
    "gyre_unet_set_tiling": (_i, [_vp, _i]),
    "gyre_vae_set_tiling": (_i, [_vp, _i]),
    "gyre_unet_debug_tap": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "gyre_unet_forward_ex": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp]),
    "gyre_unet_forward_ctrl": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp,
                                    C.POINTER(_vp), _i, _i, _vp, C.POINTER(_vp), _i]),
    "gyre_unet_vjp_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "gyre_unet_vjp": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _i, _vp, _i, _vp]),
    "gyre_unet_vjp_begin": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp]),
    "gyre_unet_vjp_finish": (_i, [_vp, _vp, _vp, _i, _vp, _i]),
    "gyre_unet_vjp_finish_range": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i]),
    "gyre_unet_vjp_pending": (_i, [_vp]),
    "gyre_vae_create": (_i, [C.POINTER(VAECfg), _i, C.POINTER(_vp)]),
    "gyre_vae_destroy": (None, [_vp]),
    "gyre_vae_num_params": (_i, [_vp]),
    "gyre_vae_param_key": (C.c_char_p, [_vp, _i]),
    "gyre_vae_set_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i, _vp]),
    "gyre_vae_finalize": (_i, [_vp, _vp]),
    "gyre_vae_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "gyre_vae_encode": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _i]),
    "gyre_vae_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _i]),
    "gyre_vae_decode_vjp_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "gyre_vae_decode_vjp": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _i, _vp, _i]),
    "gyre_prof_set_mask": (_i, [C.c_uint64]),
    "gyre_prof_num_classes": (_i, []),
    "gyre_prof_class_name": (C.c_char_p, [_i]),
    "gyre_prof_collect": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gyre_debug_force_gemm_cfg": (_i, [_i]),
    "gyre_debug_set_splitk_workspace": (_i, [_vp, _sz]),
    "gyre_debug_set_ar_workspace": (_i, [_vp, _sz]),
    "gyre_debug_set_wblk_workspace": (_i, [_vp, _sz]),
    "gyre_debug_attn_redo_count": (C.c_long, []),
    "gyre_debug_xattn_stamps": (_i, [_vp]),
    "gyre_debug_force_attn_variant": (_i, [_i]),
    "gyre_debug_gemm_ablation": (_i, [_i]),
    "gyre_set_batch_invariant": (_i, [_i]),
    "gyre_get_batch_invariant": (_i, []),
    "gyre_op_groupnorm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _sz, _vp]),
    "gyre_op_groupnorm_workspace": (_sz, [_i, _i, _i, _i]),
    "gyre_op_gemm_splitk_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gyre_op_conv3x3_colstats": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gyre_op_linear_colstats": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gyre_op_groupnorm_colstats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "gyre_op_layernorm": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _vp]),
    "gyre_op_linear": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "gyre_op_linear_t": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "gyre_op_ln_linear_workspace": (_sz, [_i, _i, _i]),
    "gyre_op_ln_linear": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "gyre_op_linear_rowstats_parts": (_i, [_i, _i, _i, _i]),
    "gyre_op_linear_rowstats": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "gyre_op_conv3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "gyre_op_conv3x3_shortcut": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "gyre_op_conv3x3_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i]),
    "gyre_op_repack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gyre_op_repack_linear_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "gyre_op_repack_bias": (_i, [_vp, _vp, _i, _i, _vp]),
    "gyre_op_attention": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "gyre_op_qkv": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i]),
    "gyre_op_attention_ex": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i]),
    "gyre_op_groupnorm_bwd_workspace": (_sz, [_i, _i, _i, _i]),
    "gyre_op_groupnorm_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "gyre_op_layernorm_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _f, _vp, _vp]),
    "gyre_op_geglu_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "gyre_op_attention_bwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "gyre_op_attention_bwd": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz,
                                   _vp, _i, _vp, _i, _vp, _i]),
    "gyre_op_tome_workspace": (_sz, [_i, _i, _i]),
    "gyre_op_tome_merge": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _i, _vp, _vp]),
    "gyre_op_cross_attention_block": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "gyre_op_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gyre_op_copy_probe": (_i, [_vp, _vp, _vp, _sz]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_libs: dict = {}
import threading as _threading
_tls = _threading.local()          # .last = the library this thread asked for last (check() reads ITS thread-local error string)


def lib(storage: Optional[int] = None) -> C.CDLL:
    """Load (once) and return the native library of the given storage flavour (None = the process default).  Raises GyreError if
    it is not built."""
    storage = _default_storage if storage is None else storage
    l = _libs.get(storage)
    if l is None:
        path = LIB_PATHS[storage]
        if not os.path.exists(path):
            raise GyreError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"(hipcc --offload-arch=gfx950).  There is no non-HIP fallback.")
        try:
            # local scope: both flavours export the same symbol names (each is linked -Bsymbolic and binds its own)
            l = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        except OSError as e:  # pragma: no cover
            raise GyreError(f"cannot load {path}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.gyre_abi_version() != 1:
            raise GyreError("libgyre_hip ABI version mismatch")
        if l.gyre_storage_dtype() != storage:
            raise GyreError(f"{path} stores dtype code {l.gyre_storage_dtype()}, expected {storage}")
        _libs[storage] = l
    _tls.last = l
    return l


def all_libs():
    """Both flavours (loading them): for process-/thread-wide planner switches that must hold whichever library a module uses."""
    return [lib(BF16), lib(F16)]


_EXC = {-1: ValueError, -2: KeyError, -3: GyreError, -4: GyreError, -5: GyreError, -6: NotImplementedError}


def check(rc: int, L: Optional[C.CDLL] = None) -> None:
    """Map a gyre_status to the Python exception the reference's error plumbing expects
    (services/exception_to_grpc: NotImplementedError -> UNIMPLEMENTED, ValueError -> generic).  L = the library the failing
    call went to (its thread-local error string; default: the library this thread asked lib() for last)."""
    if rc != 0:
        msg = (L or getattr(_tls, "last", None) or lib()).gyre_last_error().decode(errors="replace")
        raise _EXC.get(rc, GyreError)(f"libgyre_hip: {msg} (status {rc})")


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _TORCH_DTYPE[t.dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {t.dtype}; use float32, bfloat16 or float16") from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise GyreError(f"{name} must live on the GPU: the native path has no CPU fallback (got {t.device})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def prof_enable(class_prefixes=None) -> None:
    """Enable HIP-event timing for the kernel classes whose name starts with one of the prefixes
    (None = all, [] = off)."""
    L = lib()
    n = L.gyre_prof_num_classes()
    mask = 0
    for k in range(n):
        name = L.gyre_prof_class_name(k).decode()
        if class_prefixes is None or any(name.startswith(p) for p in class_prefixes):
            mask |= 1 << k
    L.gyre_prof_set_mask(mask)


def prof_collect() -> dict:
    """{class name: dict(launches, ms, flops, bytes)}; call after synchronising the stream."""
    L = lib()
    n = L.gyre_prof_num_classes()
    la, ms, fl, by = (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    L.gyre_prof_collect(la, ms, fl, by)
    return {L.gyre_prof_class_name(k).decode(): dict(launches=int(la[k]), ms=float(ms[k]), flops=float(fl[k]),
                                                       bytes=float(by[k])) for k in range(n) if la[k]}
