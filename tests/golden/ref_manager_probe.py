"""Build-container probe: drive the REFERENCE's own engine-loading code (gyre/manager.py:1024-1252) and weight-cloning
code (gyre/pipeline/model_utils.py:172-259) with the native module classes named the way an engines.yaml `class:` entry
names them.  Absent third-party roots are stubbed as in make_golden.py.  Prints one JSON object.
Run as a subprocess by tests/test_reference_manager_loading.py; nothing here ships."""
import json
import os
import sys
import tempfile

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import make_golden as mg
    mg.STUB_ROOTS.update({"huggingface_hub", "omegaconf", "pytorch_lightning", "ldm"} - set(sys.modules))
    mg._install()
    import huggingface_hub                      # real one when installed (manager uses filter_repo_objects)
    from gyre import manager as ref_manager
    # the real diffusers.pipelines has no attribute named like an external library; the permissive stub would claim one
    import types
    ref_manager.pipelines = types.SimpleNamespace()
    from gyre.pipeline import model_utils
    from safetensors.torch import save_file
    from gyre_amd import config as gcfg, weights
    from gyre_amd.modules import GyreHipUNet, GyreHipVAE

    out = {}
    mgr = object.__new__(ref_manager.EngineManager)
    mgr._mode = type("Mode", (), {"fp16": False})()
    with tempfile.TemporaryDirectory() as tmp:
        ucfg, vcfg = gcfg.tiny_unet(), gcfg.tiny_vae()
        usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg))
        vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg))
        for sub, cfg, sd, extra in (("unet", ucfg, usd, {"attention_head_dim": list(ucfg.num_heads),
                                                         "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in ucfg.attn_levels]}),
                                    ("vae", vcfg, vsd, {})):
            os.makedirs(os.path.join(tmp, sub))
            j = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.to_dict().items()}
            j.update(extra)
            json.dump(j, open(os.path.join(tmp, sub, "config.json"), "w"))
            save_file(sd, os.path.join(tmp, sub, "diffusion_pytorch_model.safetensors"))
        # 1. `class: gyre_amd.modules.GyreHipUNet` -> _import_class + from_pretrained (manager.py:1176-1252)
        cls = mgr._import_class("gyre_amd.modules.GyreHipUNet")
        out["import_class"] = cls is GyreHipUNet
        unet = mgr._load_model_from_weights(tmp, "unet", "gyre_amd.modules.GyreHipUNet")
        out["unet_type"] = type(unet).__name__
        out["unet_source"] = unet._source == os.path.join(tmp, "unet")
        out["unet_keys_equal"] = set(unet.state_dict()) == set(usd)
        out["unet_values_equal"] = all(torch.equal(unet.state_dict()[k], usd[k]) for k in usd)
        out["unet_eval"] = not unet.training
        out["unet_config"] = [unet.config.in_channels, unet.config.sample_size, list(unet.config.block_out_channels)]
        # fp16 engines: torch_dtype / variant kwargs are forwarded because they are in the signature
        mgr._mode.fp16 = True
        h = mgr._load_model_from_weights(tmp, "vae", ("gyre_amd.modules", "GyreHipVAE"))
        out["vae_fp16_dtype"] = str(h.dtype)
        out["vae_block_out_channels"] = list(h.config.block_out_channels)
        mgr._mode.fp16 = False
        # 2. the fallback loader (manager.py:1068-1112): Class(**config) + load_state_dict + eval
        fb = mgr._load_module_fallback(os.path.join(tmp, "unet"), GyreHipUNet, allow_patterns=None, ignore_patterns=None, config=ucfg)
        out["fallback_values_equal"] = all(torch.equal(fb.state_dict()[k], usd[k]) for k in usd) and not fb.training
        # 3. `class: pkg.Class/factory(arg=v)` syntax (manager.py:1114-1143)
        name, factory, args = mgr._parse_class_details("gyre_amd.modules.GyreHipUNet/from_pretrained(variant=fp16)")
        out["class_details"] = [name, factory, args]
        # 4. per-slot cloning (pipeline_wrapper.py:114-131 -> model_utils.clone_model): parameters are shared / copied
        clone = model_utils.clone_model(unet, clone_tensors="share")
        out["clone_type"] = type(clone).__name__
        out["clone_keys_equal"] = set(clone.state_dict()) == set(usd)
        out["clone_shares_storage"] = all(clone.state_dict()[k].data_ptr() == unet.state_dict()[k].data_ptr() for k in list(usd)[:8])
        out["clone_has_config"] = clone.config.in_channels == ucfg.in_channels
        # 5. LoRA sweeps walk .modules() on every generation (unified_pipeline.py:2193-2200)
        out["modules_walk"] = sum(1 for _ in unet.modules()) > 100
    print("PROBE_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
