"""nn.Module shells over the native UNet / VAE (the drop-in boundary).

The reference swaps model classes through engines.yaml ``class: "pkg.mod.Class"``
(gyre/manager.py:1024-1066,1114-1143) and then only touches this surface:

  UNet   ``unet(latents, t, encoder_hidden_states=...).sample``   unet/core.py:262-274
         ``unet.config.in_channels / sample_size / _diffusers_version``, ``unet.dtype``,
         ``.modules()`` sweeps (LoRA removal, unified_pipeline.py:2193-2200), ``clone_model``
         (model_utils.py:172-259: ordinary nn.Module parameters)
  VAE    ``vae.encode(x).latent_dist.sample(generator=g)``         unified_pipeline.py:309-313
         ``vae.decode(z).sample``                                   unified_pipeline.py:1531-1533
         ``vae.config.block_out_channels``, ``vae.dtype``, enable/disable_slicing/tiling

so these classes are plain ``nn.Module`` trees whose ``state_dict()`` keys are exactly
the diffusers checkpoint names (manager.py:1068-1112 ``load_state_dict`` fallback).
The parameters are the host/torch master copy; the first forward on a GPU uploads
+ repacks them into libgyre_hip's own bf16 NHWC/KRSC buffers.  All compute happens
in the HIP library - there is no PyTorch fallback path.
"""
from __future__ import annotations

import ctypes as C
import json
import threading
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import math
import torch
from torch import nn

from . import _lib
from .config import UNetConfig, VAEConfig, sd15_unet, sd15_vae
from .weights import synthetic_state_dict, unet_param_shapes, vae_param_shapes


class _Node(nn.Module):
    """Anonymous container; the tree only exists to give parameters diffusers names."""


def _build_tree(root: nn.Module, shapes) -> None:
    for key, shape in shapes.items():
        parts = key.split(".")
        node = root
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape), requires_grad=False))


class _NativeModule(nn.Module):
    """Common weight-sync / workspace machinery."""

    _kind = ""

    def __init__(self):
        super().__init__()
        self._handle: Optional[int] = None
        self._handle_device: Optional[torch.device] = None
        self._dirty = True
        self._ws: Optional[torch.Tensor] = None
        self._ctx_slots: list = []        # [(tensor, version, handle, B, native slot)], most recently used last
        self.register_load_state_dict_post_hook(lambda m, ik: m._weights_replaced())

    # -- lifecycle ---------------------------------------------------------------------------
    def _weights_replaced(self):
        """load_state_dict ran: the parameters are the new truth.  fp32 LoRA overrides (gyre_amd/lora.py) and the LoRA base
        cache describe the OLD weights and would shadow the new ones at the next upload - drop them."""
        ov = getattr(self, "_weight_overrides", None)
        if ov:
            ov.clear()
        st = getattr(self, "_lora_state", None)
        if st is not None:
            st["base"].clear()
            st["loras"].clear()
        self._invalidate()

    def _invalidate(self):
        self._dirty = True
        self._ctx_slots = []
        # monotonically increasing: replicas of this module (executor.DeviceSlotExecutor) compare it to know they are stale
        self._weights_version = getattr(self, "_weights_version", 0) + 1

    def _apply(self, fn, *a, **k):  # .to() / .half() / .cuda(): the native copy is stale afterwards
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def set_tiling(self, tiling) -> None:
        """Circular convolutions (reference request option ``tiling``: True / "xy" / "x" / "y" / False,
        unified_pipeline.py:1671-1712 ``set_tiling_mode`` patches every Conv2d's own padding to ``F.pad(mode="circular")``)."""
        mode = {False: 0, None: 0, True: 3, "xy": 3, "x": 1, "y": 2}.get(tiling, -1)
        if mode < 0:
            raise ValueError(f"tiling must be True, False, 'x', 'y' or 'xy', got {tiling!r}")
        self._tiling = mode
        if self._handle is not None:
            _lib.check(getattr(self._L(), f"gyre_{self._kind}_set_tiling")(C.c_void_p(self._handle), mode))
            self._tiling_applied = (self._handle, mode)

    def _apply_tiling(self, h) -> None:
        mode = getattr(self, "_tiling", 0)
        if (getattr(self, "_tiling_applied", None) or (None, 0)) != (h, mode):
            _lib.check(getattr(self._L(), f"gyre_{self._kind}_set_tiling")(C.c_void_p(h), mode))
            self._tiling_applied = (h, mode)

    def _destroy(self):
        if getattr(self, "_handle", None):
            getattr(self._L(), f"gyre_{self._kind}_destroy")(C.c_void_p(self._handle))
            self._handle = None
        # per-handle options are re-applied to whatever handle comes next: the allocator may hand a new struct the old address
        self._tiling_applied = None
        self._tome_applied = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    _NATIVE_STATE = ("_handle", "_handle_device", "_handle_storage", "_ws", "_ws_vjp", "_ctx_slots", "_vjp_pending")

    def __deepcopy__(self, memo):
        """A copy gets its OWN native handle (created at first use): the reference clones modules with ``deepcopy`` - per
        device slot in ``clone_model`` (gyre/pipeline/model_utils.py:172-217) - and two Python objects sharing one
        ``gyre_unet*`` would free it twice and race on its workspace planner."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._NATIVE_STATE:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._handle, new._handle_device, new._ws, new._ctx_slots, new._dirty = None, None, None, [], True
        return new

    def _c_cfg(self):
        raise NotImplementedError

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    def _storage(self) -> int:
        """Storage flavour of the native copy: float16 parameters -> the fp16 library (the reference's GPU dtype, manager.py:146-151),
        bfloat16 -> the bf16 one, float32 -> the process default (_lib.default_storage)."""
        return _lib.storage_for(self.dtype)

    def _L(self):
        """The library this module's handle lives in (a handle is destroyed by the library that created it); without a handle, the
        one its parameters' dtype selects."""
        st = getattr(self, "_handle_storage", None) if getattr(self, "_handle", None) is not None else None
        return _lib.lib(self._storage() if st is None else st)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _sync(self, device: torch.device) -> int:
        """Create the native handle on `device` if needed and (re)upload dirty weights."""
        L = self._L()
        if device.type != "cuda":
            raise _lib.GyreError(f"{type(self).__name__} runs on the MI355X HIP path only; move it to a GPU "
                                 f"(no CPU fallback)")
        if self._handle is not None and (self._handle_device != device or getattr(self, "_handle_storage", None) != self._storage()):
            self._destroy()                      # other device, or .to(dtype) moved the module to the other storage flavour
            L = self._L()
        with torch.cuda.device(device):
            if self._handle is None:
                h = C.c_void_p()
                cfg = self._c_cfg()
                _lib.check(getattr(L, f"gyre_{self._kind}_create")(C.byref(cfg), device.index or 0, C.byref(h)))
                self._handle, self._handle_device, self._dirty = h.value, device, True
                self._handle_storage = self._storage()
            if self._dirty:
                st = _lib.stream_ptr(device)
                setw = getattr(L, f"gyre_{self._kind}_set_weight")
                nkeys = getattr(L, f"gyre_{self._kind}_num_params")(C.c_void_p(self._handle))
                native = {getattr(L, f"gyre_{self._kind}_param_key")(C.c_void_p(self._handle), i).decode()
                          for i in range(nkeys)}
                for key, p in self.state_dict().items():
                    if key not in native:
                        continue  # host-side parameters (e.g. SDXL add_embedding) stay in PyTorch
                    t = self._upload_source(key, p)
                    if t.device != device:
                        t = t.to(device)
                    t = t.contiguous()
                    shape = (C.c_int64 * t.ndim)(*t.shape)
                    _lib.check(setw(C.c_void_p(self._handle), key.encode(), C.c_void_p(t.data_ptr()),
                                    _lib.dtype_code(t), shape, t.ndim, C.c_void_p(st)))
                _lib.check(getattr(L, f"gyre_{self._kind}_finalize")(C.c_void_p(self._handle), C.c_void_p(st)))
                torch.cuda.current_stream(device).synchronize()  # load time only: temporaries may now be freed
                self._dirty = False
        return self._handle

    def _upload_source(self, key: str, p: torch.Tensor) -> torch.Tensor:
        """The tensor handed to gyre_*_set_weight for `key`: the parameter, or an fp32 override of it (LoRA merges are
        kept in fp32 so that base + delta is rounded to bf16 ONCE, by the repack kernel - a bf16 master would swallow
        deltas below half an ulp of the weight, gyre_amd/lora.py)."""
        ov = getattr(self, "_weight_overrides", None)
        if ov and key in ov:
            return ov[key].detach()
        return p.detach()

    def _workspace(self, nbytes: int, device: torch.device) -> torch.Tensor:
        if nbytes == 0:
            raise _lib.GyreError("libgyre_hip: " + self._L().gyre_last_error().decode())
        if self._ws is None or self._ws.device != device or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        return self._ws

    # -- loading -----------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config=None, **kw):
        return cls(config, **kw) if config is not None else cls(**kw)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: Optional[torch.dtype] = None, variant: Optional[str] = None,
                        subfolder: Optional[str] = None, **_ignored):
        """Reads the HF diffusers folder layout (config.json + *.safetensors) without diffusers;
        signature subset of what gyre/manager.py:1176-1242 passes (torch_dtype, variant="fp16", and kwargs it only
        forwards when present in the signature)."""
        from safetensors.torch import load_file
        if subfolder:
            path = os.path.join(path, subfolder)
        cfg = None
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                cfg = cls._config_from_json(json.load(f))
        model = cls(cfg) if cfg is not None else cls()
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no *.safetensors in {path}")
        # diffusers naming: diffusion_pytorch_model[.<variant>].safetensors; prefer the requested variant, else the
        # un-suffixed file, else whatever single file is there (manager.py:1068-1112 fallback behaviour)
        want = [f for f in files if variant and f.endswith(f".{variant}.safetensors")]
        plain = [f for f in files if f.count(".") == 1]
        pick = (want or plain or files)[0]
        model.load_state_dict(load_file(os.path.join(path, pick)))
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        model._source = path
        return model.eval()

    def load_synthetic(self, seed: int = 0):
        """Seeded synthetic weights of the exact architecture (no SD checkpoints exist offline)."""
        self.load_state_dict(synthetic_state_dict(self._shapes(), seed))
        return self


class GyreHipUNet(_NativeModule):
    """Drop-in for diffusers.UNet2DConditionModel on the reference's hot path."""

    _kind = "unet"
    CTX_SLOTS = 4                         # include/gyre_hip.h GYRE_CTX_SLOTS

    def __init__(self, config: Optional[UNetConfig] = None):
        super().__init__()
        self.config = config or sd15_unet()
        self._ctx_slots = []
        _build_tree(self, unet_param_shapes(self.config))

    def _shapes(self):
        return unet_param_shapes(self.config)

    @staticmethod
    def _config_from_json(j: dict) -> UNetConfig:
        boc = tuple(j.get("block_out_channels", (320, 640, 1280, 1280)))
        n = len(boc)
        down = j.get("down_block_types", ["CrossAttnDownBlock2D"] * (n - 1) + ["DownBlock2D"])
        # diffusers quirk kept by every SD config: "attention_head_dim" holds the NUMBER of heads per level
        ahd = j.get("num_attention_heads") or j.get("attention_head_dim", 8)
        heads = tuple(ahd) if isinstance(ahd, (list, tuple)) else (ahd,) * n
        tl = j.get("transformer_layers_per_block", 1)
        depth = tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * n
        kw = {}
        if j.get("addition_embed_type") == "text_time":          # SDXL
            kw = dict(addition_embed_type="text_time", addition_time_embed_dim=j.get("addition_time_embed_dim", 256),
                      projection_class_embeddings_input_dim=j.get("projection_class_embeddings_input_dim", 2816))
        elif j.get("addition_embed_type") is not None:
            raise NotImplementedError(f"addition_embed_type {j['addition_embed_type']!r}")
        if j.get("class_embed_type") is not None or j.get("num_class_embeds") is not None:
            raise NotImplementedError("class-conditioned UNets are outside the native hot path")
        return UNetConfig(in_channels=j.get("in_channels", 4), out_channels=j.get("out_channels", 4),
                          block_out_channels=boc, layers_per_block=j.get("layers_per_block", 2),
                          attn_levels=tuple("CrossAttn" in d for d in down), num_heads=heads,
                          cross_attention_dim=j.get("cross_attention_dim", 768),
                          norm_num_groups=j.get("norm_num_groups", 32), transformer_depth=depth,
                          use_linear_projection=bool(j.get("use_linear_projection", False)),
                          sample_size=j.get("sample_size", 64), flip_sin_to_cos=j.get("flip_sin_to_cos", True),
                          freq_shift=float(j.get("freq_shift", 0)), **kw)

    def _c_cfg(self):
        c, cfg = self.config, _lib.UNetCfg()
        n = len(c.block_out_channels)
        cfg.in_channels, cfg.out_channels, cfg.n_levels = c.in_channels, c.out_channels, n
        for i in range(n):
            cfg.block_out_channels[i] = c.block_out_channels[i]
            cfg.attn_levels[i] = int(c.attn_levels[i])
            cfg.num_heads[i] = c.num_heads[i]
            cfg.transformer_depth[i] = c.transformer_depth[i]
        cfg.layers_per_block = c.layers_per_block
        cfg.cross_attention_dim = c.cross_attention_dim
        cfg.norm_num_groups = c.norm_num_groups
        cfg.use_linear_projection = int(c.use_linear_projection)
        cfg.flip_sin_to_cos = int(c.flip_sin_to_cos)
        cfg.freq_shift = c.freq_shift
        return cfg

    def set_tome(self, r: int) -> None:
        """Token merging for the self-attentions (reference pipeline option ``tome: <r>``, unified_pipeline.py:1580-1588,
        nonfree/tome_patcher.py:14-52): merge the r most redundant keys / values of every self-attention (clipped to half
        the tokens of the layer).  0 switches it off."""
        r = int(r)
        if r < 0:
            raise ValueError("tome r must be >= 0")
        self._tome_r = r
        if self._handle is not None:
            _lib.check(self._L().gyre_unet_set_tome(C.c_void_p(self._handle), r))

    def _aug_embedding(self, added_cond_kwargs, B: int, dev) -> Optional[torch.Tensor]:
        """SDXL text_time conditioning: tiny MLP on the host (PyTorch-ROCm), result handed to the native call."""
        if self.config.addition_embed_type != "text_time":
            return None
        if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
            raise ValueError("this UNet needs added_cond_kwargs={'text_embeds': [B,D], 'time_ids': [B,6]}")
        c = self.config
        te_in, ids_in = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        # the conditioning of a request does not change between its steps: keep the last result (keyed by the very tensors and their
        # in-place version counters), so the 31 evaluations of a request run the little MLP once
        key = (te_in.data_ptr(), te_in._version, tuple(te_in.shape), ids_in.data_ptr(), ids_in._version, tuple(ids_in.shape), B, str(dev),
               getattr(self, "_weights_version", 0))
        memo = getattr(self, "_aug_memo", None)
        if memo is not None and memo[0] == key:
            return memo[1]
        te, ids = te_in.to(dev, torch.float32), ids_in.to(dev)
        half = c.addition_time_embed_dim // 2
        # (math.log on the host: torch.tensor(10000.0, device=dev) is a pageable host-to-device copy - a stream sync per UNet call)
        freq = torch.exp(torch.arange(half, device=dev, dtype=torch.float32) * (-math.log(10000.0) / (half - c.freq_shift)))
        ang = ids.flatten()[:, None].float() * freq[None]
        emb = torch.cat([torch.cos(ang), torch.sin(ang)] if c.flip_sin_to_cos else [torch.sin(ang), torch.cos(ang)], dim=-1)
        a = torch.cat([te, emb.reshape(B, -1)], dim=-1)
        ae = self.add_embedding
        a = torch.nn.functional.linear(a, ae.linear_1.weight.float(), ae.linear_1.bias.float())
        a = torch.nn.functional.linear(torch.nn.functional.silu(a), ae.linear_2.weight.float(), ae.linear_2.bias.float())
        a = a.contiguous()
        self._aug_memo = (key, a, te_in, ids_in)          # (the inputs are kept alive so that their addresses cannot be reused)
        return a

    def _prepare(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals,
                 mid_block_additional_residual, adapter_states):
        if encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states is required")
        if adapter_states is not None and any(not isinstance(a, torch.Tensor) for a in adapter_states):
            raise NotImplementedError("nested / style T2I-adapter states")
        if sample.ndim != 4 or sample.shape[1] != self.config.in_channels:
            raise ValueError(f"expected latents [B,{self.config.in_channels},H,W], got {tuple(sample.shape)}")
        B = sample.shape[0]
        if encoder_hidden_states.ndim != 3 or encoder_hidden_states.shape[0] != B \
                or encoder_hidden_states.shape[2] != self.config.cross_attention_dim:
            raise ValueError(f"expected encoder_hidden_states [{B},S,{self.config.cross_attention_dim}], "
                             f"got {tuple(encoder_hidden_states.shape)}")
        dev = sample.device
        h = self._sync(dev)
        self._apply_tiling(h)
        if (getattr(self, "_tome_applied", None) or (None, None)) != (h, getattr(self, "_tome_r", 0)):
            _lib.check(self._L().gyre_unet_set_tome(C.c_void_p(h), getattr(self, "_tome_r", 0)))
            self._tome_applied = (h, getattr(self, "_tome_r", 0))
        self._t_uniform = not isinstance(timestep, torch.Tensor) or timestep.numel() == 1
        if isinstance(timestep, torch.Tensor):
            t = timestep.to(dev).to(torch.int64).reshape(-1)
            if t.numel() == 1:
                t = t.expand(B)
        else:
            # a Python scalar (what the k-diffusion wrapper passes): fill on the device.  torch.as_tensor(int, device=...)
            # is a pageable host-to-device copy, which on ROCm blocks the host until the stream has drained - one
            # pipeline bubble per UNet call.
            t = torch.full((B,), int(timestep), dtype=torch.int64, device=dev)
        if t.numel() != B:
            raise ValueError(f"timestep must be a scalar or have {B} elements")
        return h, t.contiguous()

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor = None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, adapter_states=None,
                added_cond_kwargs=None, return_dict: bool = True, **_ignored):
        """Noise prediction.  When autograd is recording and ``sample`` requires grad (the reference's CLIP-guided mode,
        unet/clipguided.py:301-338), the result carries a backward that calls the native input-gradient sweep
        (gyre_unet_vjp); weights and the text context never receive gradients."""
        h, t = self._prepare(sample, timestep, encoder_hidden_states, down_block_additional_residuals,
                             mid_block_additional_residual, adapter_states)
        residuals = None
        if down_block_additional_residuals is not None or mid_block_additional_residual is not None or adapter_states:
            # ControlNet outputs (reference unet/core.py:40-64): host tensors in NCHW, added natively to the skip connections
            # the up path consumes / to the mid block's output (controlnet/unet_patcher.py:30-95); T2I-adapter states
            # (unet/core.py:212-216): one per down level, added in place inside the down path (t2i_adapter/unet_patcher.py)
            residuals = (list(down_block_additional_residuals or []), mid_block_additional_residual, list(adapter_states or []))
        if torch.is_grad_enabled() and sample.requires_grad:
            if residuals is not None:
                raise NotImplementedError("input gradients through ControlNet residuals (CLIP guidance + ControlNet)")
            out = _UNetInputGrad.apply(sample, self, h, t, encoder_hidden_states, added_cond_kwargs,
                                       getattr(_HINTS, "grad_samples", None))
        else:
            with torch.no_grad():
                out = self._forward_native(h, sample, t, encoder_hidden_states, added_cond_kwargs, residuals)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def _forward_native(self, h, sample, t, encoder_hidden_states, added_cond_kwargs, residuals=None):
        B, _, H, W = sample.shape
        dev = sample.device
        x = sample.contiguous()
        ctx = encoder_hidden_states.to(dev).contiguous()
        _lib.require_gpu_tensor(x, "latents")
        S = ctx.shape[1]
        L = self._L()
        with torch.cuda.device(dev):
            # context cache: the denoising loop passes the SAME embeddings tensor on every call (the reference binds
            # it once per request, unet/core.py:242-259); project it through the cross-attention K/V weights once.
            # Identity + version of a tensor we keep referenced => the storage cannot have been recycled.  CTX_SLOTS
            # entries, least recently used evicted: the leaves of a hires-fix / graft tree and CFGUNet_Sequential
            # (unet/cfg.py:27-38, unet/hires_fix.py:123-235) alternate between contexts on every step.
            src = encoder_hidden_states
            try:
                ver = src._version
            except RuntimeError:          # inference tensors carry no version counter: never trust the cache for them
                ver = None
            slots = self._ctx_slots
            hit = None
            if ver is not None:
                for i, (s_src, s_ver, s_h, s_B, _) in enumerate(slots):
                    if s_src is src and s_ver == ver and s_h == h and s_B == B:
                        hit = i
                        break
            if hit is not None:
                entry = slots.pop(hit)
                _lib.check(L.gyre_unet_select_context(C.c_void_p(h), entry[4]))
                slots.append(entry)
            else:
                used = {e[4] for e in slots}
                free = [k for k in range(self.CTX_SLOTS) if k not in used]
                slot = free[0] if free else slots.pop(0)[4]
                _lib.check(L.gyre_unet_set_context_slot(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)),
                                                        C.c_void_p(ctx.data_ptr()), _lib.dtype_code(ctx), B, S, slot))
                if ver is not None:
                    slots.append((src, ver, h, B, slot))
                else:
                    self._ctx_slots = [e for e in slots if e[4] != slot]
            need = L.gyre_unet_workspace_bytes(C.c_void_p(h), B, H, W, S)
            if need == 0:
                _lib.check(-1 if "unet:" in L.gyre_last_error().decode() else -4)
            ws = self._workspace(need, dev)
            wp = (ws.data_ptr() + 255) & ~255
            out = torch.empty((B, self.config.out_channels, H, W), dtype=sample.dtype, device=dev)
            aug = self._aug_embedding(added_cond_kwargs, B, dev)
            augp = C.c_void_p(aug.data_ptr()) if aug is not None else None
            # one timestep for the whole batch (a scalar was passed): always stated explicitly, a stale hint never survives
            _lib.check(L.gyre_unet_hint_uniform_timestep(C.c_void_p(h), 1 if getattr(self, "_t_uniform", False) else 0))
            _lib.check(L.gyre_unet_hint_cfg_pairs(C.c_void_p(h), 1 if (getattr(_HINTS, "cfg_pairs", False) and B % 2 == 0) else 0))
            if residuals is None:
                _lib.check(L.gyre_unet_forward_ex(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(x.data_ptr()),
                                                  _lib.dtype_code(x), C.c_void_p(t.data_ptr()), None,
                                                  _lib.dtype_code(ctx), B, H, W, S, C.c_void_p(wp), need,
                                                  C.c_void_p(out.data_ptr()), _lib.dtype_code(out), augp))
            else:
                down, mid, adapt = residuals
                rdt = (down[0] if down else (mid if mid is not None else adapt[0])).dtype
                keep = [r.to(dev, rdt).contiguous() for r in down]             # alive until the call has been enqueued
                akeep = [r.to(dev, rdt).contiguous() for r in adapt]
                for r in keep + akeep:
                    if r.ndim != 4 or r.shape[0] != B:
                        raise ValueError(f"residual / adapter tensors must be [B,C,h,w] with B={B}, got {tuple(r.shape)}")
                midk = mid.to(dev, rdt).contiguous() if mid is not None else None
                arr = (C.c_void_p * max(len(keep), 1))(*[r.data_ptr() for r in keep])
                aarr = (C.c_void_p * max(len(akeep), 1))(*[r.data_ptr() for r in akeep])
                _lib.check(L.gyre_unet_forward_ctrl(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(x.data_ptr()),
                                                    _lib.dtype_code(x), C.c_void_p(t.data_ptr()), None, _lib.dtype_code(ctx),
                                                    B, H, W, S, C.c_void_p(wp), need, C.c_void_p(out.data_ptr()),
                                                    _lib.dtype_code(out), augp, arr, len(keep),
                                                    _lib.dtype_code(keep[0] if keep else (midk if midk is not None else akeep[0])),
                                                    C.c_void_p(midk.data_ptr()) if midk is not None else None, aarr, len(akeep)))
                self._residual_keep = (keep, midk, akeep)
        return out

    def _vjp_begin(self, h, sample, t, encoder_hidden_states, added_cond_kwargs):
        """Forward pass that leaves the reverse sweep's activations in self._ws_vjp; returns the pending token (id, eps)."""
        B, _, H, W = sample.shape
        dev = sample.device
        x = sample.contiguous()
        ctx = encoder_hidden_states.to(dev).contiguous()
        _lib.require_gpu_tensor(x, "latents")
        S = ctx.shape[1]
        L = self._L()
        with torch.cuda.device(dev):
            need = L.gyre_unet_vjp_workspace_bytes(C.c_void_p(h), B, H, W, S)
            if need == 0:
                _lib.check(-4)
            ws = getattr(self, "_ws_vjp", None)
            if ws is None or ws.device != dev or ws.numel() < need + 256:
                self._ws_vjp = None
                self._ws_vjp = ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            wp = (ws.data_ptr() + 255) & ~255
            eps = torch.empty((B, self.config.out_channels, H, W), dtype=sample.dtype, device=dev)
            aug = self._aug_embedding(added_cond_kwargs, B, dev)
            _lib.check(L.gyre_unet_vjp_begin(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(x.data_ptr()),
                                             _lib.dtype_code(x), C.c_void_p(t.data_ptr()), C.c_void_p(ctx.data_ptr()),
                                             _lib.dtype_code(ctx), B, H, W, S, C.c_void_p(wp), need, C.c_void_p(eps.data_ptr()),
                                             _lib.dtype_code(eps), C.c_void_p(aug.data_ptr()) if aug is not None else None))
        self._vjp_pending = token = (object(), eps)
        return token

    def _vjp_finish(self, h, sample, d_out, grad_samples=None):
        """Reverse sweep on the pending forward state; grad_samples = (b0, nb): only those samples' cotangent is non-zero by the
        caller's statement (modules.grad_samples) - the sweep runs on them alone, the other samples' gradient is zero."""
        dev = sample.device
        self._vjp_pending = None
        if grad_samples is not None and tuple(grad_samples) != (0, sample.shape[0]):
            b0, nb = grad_samples
            g = d_out[b0:b0 + nb].contiguous()
            _lib.require_gpu_tensor(g, "d_eps")
            dx = torch.zeros_like(sample)
            part = torch.empty_like(sample[b0:b0 + nb])
            with torch.cuda.device(dev):
                _lib.check(self._L().gyre_unet_vjp_finish_range(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(g.data_ptr()),
                                                                 _lib.dtype_code(g), C.c_void_p(part.data_ptr()), _lib.dtype_code(part),
                                                                 int(b0), int(nb)))
            dx[b0:b0 + nb] = part
            return dx
        g = d_out.contiguous()
        _lib.require_gpu_tensor(g, "d_eps")
        dx = torch.empty_like(sample)
        with torch.cuda.device(dev):
            _lib.check(self._L().gyre_unet_vjp_finish(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(g.data_ptr()),
                                                       _lib.dtype_code(g), C.c_void_p(dx.data_ptr()), _lib.dtype_code(dx)))
        return dx

    def _vjp_native(self, h, sample, t, encoder_hidden_states, added_cond_kwargs, d_out):
        """(eps, d_sample) from one native call: forward keeping the adjoints' inputs + reverse sweep."""
        B, _, H, W = sample.shape
        dev = sample.device
        x, g = sample.contiguous(), d_out.contiguous()
        ctx = encoder_hidden_states.to(dev).contiguous()
        _lib.require_gpu_tensor(x, "latents")
        _lib.require_gpu_tensor(g, "d_eps")
        S = ctx.shape[1]
        L = self._L()
        with torch.cuda.device(dev):
            need = L.gyre_unet_vjp_workspace_bytes(C.c_void_p(h), B, H, W, S)
            if need == 0:
                _lib.check(-4)
            ws = self._workspace(need, dev)
            wp = (ws.data_ptr() + 255) & ~255
            eps = torch.empty((B, self.config.out_channels, H, W), dtype=sample.dtype, device=dev)
            dx = torch.empty_like(x)
            aug = self._aug_embedding(added_cond_kwargs, B, dev)
            _lib.check(L.gyre_unet_vjp(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(x.data_ptr()),
                                       _lib.dtype_code(x), C.c_void_p(t.data_ptr()), C.c_void_p(ctx.data_ptr()),
                                       _lib.dtype_code(ctx), B, H, W, S, C.c_void_p(g.data_ptr()), _lib.dtype_code(g),
                                       C.c_void_p(wp), need, C.c_void_p(eps.data_ptr()), _lib.dtype_code(eps),
                                       C.c_void_p(dx.data_ptr()), _lib.dtype_code(dx),
                                       C.c_void_p(aug.data_ptr()) if aug is not None else None))
        return eps, dx


class _UNetInputGrad(torch.autograd.Function):
    """autograd node of GyreHipUNet.forward: the forward pass keeps the activations the adjoints need in a workspace of its
    own (gyre_unet_vjp_begin), backward runs the reverse sweep on them (gyre_unet_vjp_finish).  If anything else ran on the
    handle in between, backward recomputes through the one-shot gyre_unet_vjp instead."""

    @staticmethod
    def forward(fctx, sample, module, h, t, enc, added, grad_samples=None):
        fctx.module, fctx.h, fctx.added, fctx.grad_samples = module, h, added, grad_samples
        fctx.save_for_backward(sample.detach(), t, enc.detach())
        fctx.token = module._vjp_begin(h, sample.detach(), t, enc.detach(), added)
        return fctx.token[1]

    @staticmethod
    def backward(fctx, d_out):
        sample, t, enc = fctx.saved_tensors
        m = fctx.module
        if getattr(m, "_vjp_pending", None) is fctx.token and m._L().gyre_unet_vjp_pending(C.c_void_p(fctx.h)):
            return m._vjp_finish(fctx.h, sample, d_out, fctx.grad_samples), None, None, None, None, None, None
        # state dropped by another call on the handle (gyre_unet_vjp_pending == 0): recompute in one shot (whole batch: the
        # cotangent of the other samples is zero)
        _, dx = m._vjp_native(fctx.h, sample, t, enc, fctx.added, d_out)
        return dx, None, None, None, None, None, None


_HINTS = threading.local()


class grad_samples:
    """``with grad_samples(b0, nb): eps = unet(x.requires_grad_(), ...)`` - the caller states that only samples [b0, b0 + nb) of the
    result will receive a cotangent (the others are used detached): the native reverse sweep then runs on those samples alone
    (include/gyre_hip.h gyre_unet_vjp_finish_range).  The CLIP-guided mode evaluates cat[uncond, cond] in one activation-keeping
    pass and differentiates the conditional half (gyre_amd/clipguided.py).  Ignored by modules without a native sweep."""

    def __init__(self, b0: int, nb: int):
        self.rng = (int(b0), int(nb))

    def __enter__(self):
        self.prev = getattr(_HINTS, "grad_samples", None)
        _HINTS.grad_samples = self.rng
        return self

    def __exit__(self, *exc):
        _HINTS.grad_samples = self.prev
        return False


class cfg_pairs:
    """``with cfg_pairs(): unet(cat[x, x], t, cat[uncond, cond])`` - the caller states that sample b and sample b + B/2 of the
    UNet calls made inside (on this thread) have identical latents and timesteps, as the reference's CFGUNet_Parallel builds
    them (unet/cfg.py:49-57).  GyreHipUNet then evaluates everything in front of the first cross-attention once per pair
    (include/gyre_hip.h gyre_unet_hint_cfg_pairs).  ``GYRE_CFG_SHARED_PREFIX=0`` turns the hint off process-wide."""

    def __enter__(self):
        self._prev = getattr(_HINTS, "cfg_pairs", False)
        _HINTS.cfg_pairs = os.environ.get("GYRE_CFG_SHARED_PREFIX", "1") != "0"
        return self

    def __exit__(self, *exc):
        _HINTS.cfg_pairs = self._prev
        return False


def set_batch_invariant(canonical_samples: int = 16) -> int:
    """Plan every split-K factor for ``canonical_samples`` batch entries whatever the real batch is (0 = off, the
    default): any split of a request over GPUs / sub-batches is then bit-identical (include/gyre_hip.h
    gyre_set_batch_invariant; reference property tests/batch_independance.py:15-27).  Returns the previous value."""
    return int([L.gyre_set_batch_invariant(int(canonical_samples)) for L in _lib.all_libs()][0])      # both storage flavours


class DiagonalGaussian:
    """diffusers DiagonalGaussianDistribution [3P] as used at unified_pipeline.py:309-313."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self) -> torch.Tensor:
        return self.mean


class GyreHipVAE(_NativeModule):
    """Drop-in for diffusers.AutoencoderKL on the reference's hot path."""

    _kind = "vae"

    def __init__(self, config: Optional[VAEConfig] = None):
        super().__init__()
        self.config = config or sd15_vae()
        _build_tree(self, vae_param_shapes(self.config))

    def _shapes(self):
        return vae_param_shapes(self.config)

    @staticmethod
    def _config_from_json(j: dict) -> VAEConfig:
        return VAEConfig(in_channels=j.get("in_channels", 3), out_channels=j.get("out_channels", 3),
                         latent_channels=j.get("latent_channels", 4),
                         block_out_channels=tuple(j.get("block_out_channels", (128, 256, 512, 512))),
                         layers_per_block=j.get("layers_per_block", 2), norm_num_groups=j.get("norm_num_groups", 32),
                         scaling_factor=j.get("scaling_factor", 0.18215), sample_size=j.get("sample_size", 512))

    def _c_cfg(self):
        c, cfg = self.config, _lib.VAECfg()
        n = len(c.block_out_channels)
        cfg.in_channels, cfg.out_channels, cfg.latent_channels, cfg.n_levels = c.in_channels, c.out_channels, \
            c.latent_channels, n
        for i in range(n):
            cfg.block_out_channels[i] = c.block_out_channels[i]
        cfg.layers_per_block = c.layers_per_block
        cfg.norm_num_groups = c.norm_num_groups
        return cfg

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # accept the post-0.16 attention key names (to_q/to_k/to_v/to_out.0) as aliases
        ren = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
        fixed = {}
        for k, v in state_dict.items():
            if ".attentions." in k:
                for a, b in ren.items():
                    k = k.replace(a, b)
            fixed[k] = v
        return super().load_state_dict(fixed, strict=strict, **kw)

    # the reference toggles these for small-VRAM GPUs (pipeline_wrapper.py:171-186); with 288 GB
    # of HBM the whole batch is decoded in one pass, so they are accepted and ignored.
    def enable_slicing(self): pass
    def disable_slicing(self): pass
    def enable_tiling(self): pass
    def disable_tiling(self): pass

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if x.ndim != 4 or x.shape[1] != self.config.in_channels:
            raise ValueError(f"expected image [B,{self.config.in_channels},H,W], got {tuple(x.shape)}")
        dev = x.device
        h = self._sync(dev)
        self._apply_tiling(h)
        x = x.contiguous()
        _lib.require_gpu_tensor(x, "image")
        B, _, H, W = x.shape
        L = self._L()
        f = 2 ** (len(self.config.block_out_channels) - 1)
        with torch.cuda.device(dev):
            need = L.gyre_vae_workspace_bytes(C.c_void_p(h), B, H, W, 0)
            if need == 0:
                _lib.check(-1)
            ws = self._workspace(need, dev)
            wp = (ws.data_ptr() + 255) & ~255
            out = torch.empty((B, 2 * self.config.latent_channels, H // f, W // f), dtype=x.dtype, device=dev)
            _lib.check(L.gyre_vae_encode(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(x.data_ptr()),
                                         _lib.dtype_code(x), B, H, W, C.c_void_p(wp), need,
                                         C.c_void_p(out.data_ptr()), _lib.dtype_code(out)))
        dist = DiagonalGaussian(out)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """Latents -> image.  Differentiable with respect to ``z`` when autograd is recording (the reference decodes the
        CLIP cut-outs under autograd, unet/clipguided.py:366-386): backward = gyre_vae_decode_vjp."""
        if z.ndim != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"expected latents [B,{self.config.latent_channels},h,w], got {tuple(z.shape)}")
        h = self._sync(z.device)
        self._apply_tiling(h)
        if torch.is_grad_enabled() and z.requires_grad:
            out = _VAEDecodeInputGrad.apply(z, self, h)
        else:
            with torch.no_grad():
                out = self._decode_native(h, z, None)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def _decode_native(self, h, z, d_img):
        """d_img None: image; else (image, d_z) through gyre_vae_decode_vjp."""
        dev = z.device
        z = z.contiguous()
        _lib.require_gpu_tensor(z, "latents")
        B, _, hl, wl = z.shape
        L = self._L()
        f = 2 ** (len(self.config.block_out_channels) - 1)
        with torch.cuda.device(dev):
            need = (L.gyre_vae_workspace_bytes(C.c_void_p(h), B, hl, wl, 1) if d_img is None else
                    L.gyre_vae_decode_vjp_workspace_bytes(C.c_void_p(h), B, hl, wl))
            if need == 0:
                _lib.check(-1)
            ws = self._workspace(need, dev)
            wp = (ws.data_ptr() + 255) & ~255
            out = torch.empty((B, self.config.out_channels, hl * f, wl * f), dtype=z.dtype, device=dev)
            if d_img is None:
                _lib.check(L.gyre_vae_decode(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(z.data_ptr()),
                                             _lib.dtype_code(z), B, hl, wl, C.c_void_p(wp), need,
                                             C.c_void_p(out.data_ptr()), _lib.dtype_code(out)))
                return out
            g = d_img.contiguous()
            _lib.require_gpu_tensor(g, "d_image")
            dz = torch.empty_like(z)
            _lib.check(L.gyre_vae_decode_vjp(C.c_void_p(h), C.c_void_p(_lib.stream_ptr(dev)), C.c_void_p(z.data_ptr()),
                                             _lib.dtype_code(z), B, hl, wl, C.c_void_p(g.data_ptr()), _lib.dtype_code(g),
                                             C.c_void_p(wp), need, C.c_void_p(out.data_ptr()), _lib.dtype_code(out),
                                             C.c_void_p(dz.data_ptr()), _lib.dtype_code(dz)))
            return out, dz


class _VAEDecodeInputGrad(torch.autograd.Function):
    """autograd node of GyreHipVAE.decode: backward = gyre_vae_decode_vjp (gradient of the latents only)."""

    @staticmethod
    def forward(fctx, z, module, h):
        fctx.module, fctx.h = module, h
        fctx.save_for_backward(z.detach())
        return module._decode_native(h, z.detach(), None)

    @staticmethod
    def backward(fctx, d_img):
        (z,) = fctx.saved_tensors
        _, dz = fctx.module._decode_native(fctx.h, z, d_img)
        return dz, None, None
