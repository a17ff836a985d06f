"""LoRA for the native UNet: merged into the weights before they are repacked for the kernels.

The reference attaches a forward hook to every targeted nn.Linear / nn.Conv2d that adds
``up(down(input)) * (alpha / r) * scale`` to the layer output (gyre/pipeline/lora.py:96-160), applied per request
(unified_pipeline.py:2207-2233) and removed at the start of the next one (:2190-2200).  The native UNet is one C call
with no per-layer Python hooks, so the same linear map is folded into the weight instead:

    Linear:  W' = W + s * up @ down                      (y = x W'^T  ==  x W^T + s * (x down^T) up^T)
    Conv2d:  W' = W + s * einsum(up[o,r,1,1], down[r,i,kh,kw])   (1x1 "up" after a kxk "down", same stride/padding)

with s = scale * alpha / r (alpha absent -> 1).  Key formats as detected by reference lora.py:58-93:
  kohya-ss   ``lora_unet_<module path with _>.lora_down.weight / .lora_up.weight / .alpha``   (lora.py:271-330)
  diffusers  ``<path>.processor.<to_q|to_k|to_v|to_out>_lora.down.weight / .up.weight``        (lora.py:232-268)
  cloneofsimo needs the un-vendored lora_diffusion submodule's module-search order -> NotImplementedError.
Text-encoder entries (``lora_te_``) are ignored here (CLIP stays host PyTorch and keeps the reference's own path).

The touched base weights are kept (CPU copies) so that removing / re-scaling restores them bit-exactly.

Precision: the merge happens in fp32 and the merged fp32 tensor is what the native module uploads
(``unet._weight_overrides``, consumed by modules._NativeModule._upload_source), so base + delta is rounded to bf16 once,
inside the repack kernel.  Writing the sum back into a bf16 / fp16 master parameter first would round it against the
master's grid: a delta element below half an ulp of W (|delta| < 0.2-0.4 % of |W|, typical for LoRA) would vanish,
which the reference's activation-space hook never does.  The master parameter also receives the (rounded) merged value,
for ``state_dict()`` consumers only.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, Optional

import torch

Tensor = torch.Tensor

_DETECT = ((":0:up", "cloneofsimo"), (".lora_up.weight", "kohya-ss"), (".to_k_lora.up.weight", "diffusers"))
_KOHYA_FIELDS = (".lora_up.weight", ".lora_down.weight", ".alpha")


def detect_lora_type(lora: Mapping[str, Tensor]) -> str:
    kind = None
    for key in lora.keys():
        for pat, name in _DETECT:
            if key.endswith(pat):
                kind = name
                break
        if kind:
            break
    if kind is None:
        raise ValueError("Unknown LoRA format (or not a LoRA)")
    if kind == "kohya-ss":
        for key in lora.keys():
            if not key.endswith(_KOHYA_FIELDS):
                raise ValueError("LoRA contains unknown fields, probably a Lycoris")
    return kind


def lora_delta(up: Tensor, down: Tensor, alpha: Optional[Tensor] = None) -> Tensor:
    """The weight-space image of one LoRA pair, without the user scale (fp32)."""
    r = down.shape[0]
    iscale = float(alpha) / r if alpha is not None else 1.0
    up, down = up.to(torch.float32), down.to(torch.float32)
    if down.ndim == 2:
        return (up @ down) * iscale
    if down.ndim == 4:
        if up.shape[2:] != (1, 1):
            raise ValueError("conv LoRA: the up projection must be 1x1")
        return torch.einsum("or,rikl->oikl", up[:, :, 0, 0], down) * iscale
    raise ValueError(f"Can't apply LoRA of rank-{down.ndim} tensors")


def _targets(unet: torch.nn.Module, lora: Mapping[str, Tensor]) -> Dict[str, Tensor]:
    """weight-parameter name -> unscaled delta"""
    params = dict(unet.named_parameters())
    kind = detect_lora_type(lora)
    out: Dict[str, Tensor] = {}
    if kind == "cloneofsimo":
        raise NotImplementedError("cloneofsimo LoRA files need lora_diffusion's module search order (not vendored)")
    if kind == "kohya-ss":
        flat = {name[:-len(".weight")].replace(".", "_"): name for name in params if name.endswith(".weight")}
        for key in lora.keys():
            if not key.endswith(".lora_down.weight"):
                continue
            if key.startswith("lora_te_"):
                continue
            if not key.startswith("lora_unet_"):
                raise ValueError(f"Unknown key in Kohya LoRA, don't know how to apply - {key}")
            mod = key[len("lora_unet_"):].split(".")[0]
            if mod not in flat:
                raise RuntimeError(f"Couldn't find model for {key} when applying LoRA")
            out[flat[mod]] = lora_delta(lora[key.replace(".lora_down.", ".lora_up.")], lora[key],
                                        lora.get(key.replace(".lora_down.weight", ".alpha")))
        return out
    for key in lora.keys():                                                   # diffusers attention-processor format
        if not key.endswith(".down.weight"):
            continue
        fixed = re.sub(r"processor.(.+)_lora.down.weight$",
                       lambda m: m[1] + ".0" if m[1] == "to_out" else m[1], key)
        name = fixed + ".weight"
        if name not in params:
            raise RuntimeError(f"Couldn't find model for {key} when applying LoRA")
        out[name] = lora_delta(lora[key.replace(".down.weight", ".up.weight")], lora[key])
    return out


def _state(unet):
    st = getattr(unet, "_lora_state", None)
    if st is None:
        st = {"base": {}, "loras": {}}          # base: name -> original weight (CPU); loras: id -> (deltas, scale)
        unet._lora_state = st
    return st


@torch.no_grad()
def _rebuild(unet) -> None:
    st = _state(unet)
    params = dict(unet.named_parameters())
    for name, base in st["base"].items():
        w = base.to(torch.float32)
        for deltas, scale in st["loras"].values():
            if name in deltas and scale != 0:
                d = deltas[name]
                if d.shape != w.shape:
                    raise ValueError(f"LoRA delta for {name} has shape {tuple(d.shape)}, weight {tuple(w.shape)}")
                w = w + d * scale
        p = params[name]
        p.copy_(w.to(p.device, p.dtype))
        if st["loras"] and p.dtype != torch.float32:
            ov = getattr(unet, "_weight_overrides", None)
            if ov is None:
                ov = unet._weight_overrides = {}
            ov[name] = w                      # fp32, uploaded instead of the rounded master
        elif getattr(unet, "_weight_overrides", None):
            unet._weight_overrides.pop(name, None)
    if not st["loras"]:
        st["base"].clear()
        if getattr(unet, "_weight_overrides", None):
            unet._weight_overrides.clear()
    if hasattr(unet, "_invalidate"):
        unet._invalidate()                       # the native copy is repacked from the merged weights at next use


@torch.no_grad()
def apply_lora(unet, lora: Mapping[str, Tensor], lora_id, scale: float = 1.0) -> int:
    """Merge one LoRA (a dict of tensors, e.g. safetensors.torch.load_file) under ``lora_id``.  Returns the number of
    weights touched."""
    deltas = _targets(unet, lora)
    st = _state(unet)
    params = dict(unet.named_parameters())
    for name in deltas:
        if name not in st["base"]:
            st["base"][name] = params[name].detach().to("cpu", copy=True)
    st["loras"][lora_id] = (deltas, float(scale))
    _rebuild(unet)
    return len(deltas)


def set_lora_scale(unet, lora_id, scale: float = 1.0) -> None:
    st = _state(unet)
    if lora_id not in st["loras"]:
        raise KeyError(lora_id)
    st["loras"][lora_id] = (st["loras"][lora_id][0], float(scale))
    _rebuild(unet)


def remove_lora_from_model(unet) -> None:
    st = _state(unet)
    if st["loras"]:
        st["loras"].clear()
        _rebuild(unet)
