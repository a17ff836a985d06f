"""Build libgyre_hip.so (bf16 storage) and libgyre_hip_f16.so (fp16 storage: the same sources with -DGYRE_STORE_F16, csrc/common.h)
in-tree with hipcc for gfx950 (MI355X).

The library is plain HIP + a C ABI (include/gyre_hip.h); it does not link against
torch.  It is loaded with ctypes after `import torch`, so its DT_NEEDED
libamdhip64.so.7 resolves to the HIP runtime torch already mapped (same soname) and
device pointers / streams are shared with the torch allocator.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgyre_hip.so")
LIB_F16 = os.path.join(HERE, "libgyre_hip_f16.so")
SOURCES = ["kernels_elem.hip", "kernels_gemm.hip", "kernels_gemm4s.hip", "kernels_gemm_ar.hip", "kernels_gemm_sm.hip", "kernels_conv_out.hip", "kernels_attn.hip", "kernels_xattn.hip", "kernels_tome.hip", "kernels_bwd.hip", "model.hip", "model_vjp.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_shared.h", "model_impl.h", os.path.join("..", "..", "include", "gyre_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
         "-Wno-unused-result", "-fno-gpu-rdc",
         # the fully unrolled MFMA epilogues (up to 40 fragments per wave) exceed clang's default pragma-unroll
         # budget; without this the accumulator array stays indexable and is spilled to scratch every K step
         "-mllvm", "-pragma-unroll-threshold=1000000"]


# attention is VALU-issue bound (PMC: ~90 % issue-slot utilisation, 25 % MFMA): let the MFMAs write VGPRs directly
# (gfx950 unified register file) instead of AGPRs, which removes ~90 v_accvgpr_read/write per KV tile
PER_FILE_FLAGS = {"kernels_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}

# Kernels that synchronise LDS-DMA tiles with a *counted* s_waitcnt vmcnt(N) must not spill: scratch stores also
# count in vmcnt and may be acknowledged before older loads, which would let the wait pass while a tile is still in
# flight.  Every compile therefore records the register / scratch use of each kernel (clang remarks), and
# tests/test_host_cpu.py::test_counted_vmcnt_kernels_do_not_spill checks it.  Kernels listed here drain with vmcnt(0).
RESOURCE_FILES = ("kernels_attn.hip", "kernels_gemm.hip", "kernels_gemm4s.hip", "kernels_gemm_ar.hip", "kernels_gemm_sm.hip")
# k_gemm4s: its LDS-DMA requests are always retired with vmcnt(0); the 256x256 form spills in the epilogue only
SCRATCH_ALLOWED = ("k_attn3ILi80E", "k_gemm4s")
RESOURCES_JSON = os.path.join(HERE, "build", "kernel_resources.json")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _record_resources(source: str, remarks: str) -> None:
    import json
    import re
    import threading
    kernels, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy_waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds_bytes", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    with _RES_LOCK:
        data = {}
        if os.path.exists(RESOURCES_JSON):
            try:
                data = json.load(open(RESOURCES_JSON))
            except Exception:
                data = {}
        data[source] = kernels
        data = {k: v for k, v in data.items() if k.split("/")[-1] in RESOURCE_FILES}      # (sources that no longer exist)
        with open(RESOURCES_JSON, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)


import threading as _threading
_RES_LOCK = _threading.Lock()


def build(force: bool = False, verbose: bool = False, flavours=("bf16", "f16")) -> str:
    """Compile every translation unit once per storage flavour (objects under build/ and build/f16/) and link the two libraries.
    Returns the path of the bf16 library (the default one)."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for fl in flavours:
        objdir = os.path.join(HERE, "build") if fl == "bf16" else os.path.join(HERE, "build", fl)
        os.makedirs(objdir, exist_ok=True)
        for s in SOURCES:
            src = os.path.join(CSRC, s)
            obj = os.path.join(objdir, s.replace(".hip", ".o"))
            if force or _stale(obj, [src] + hdrs):
                jobs.append((src, obj, fl))

    def cc(job):
        src, obj, fl = job
        extra = PER_FILE_FLAGS.get(os.path.basename(src), [])
        base = os.path.basename(src)
        if fl == "f16":
            extra = [*extra, "-DGYRE_STORE_F16"]
        if base in RESOURCE_FILES:
            extra = [*extra, "-Rpass-analysis=kernel-resource-usage"]
        if os.environ.get("GYRE_AR_ABLATIONS") and base == "kernels_gemm_ar.hip":   # tools/ar_ablate.py
            extra = [*extra, "-DGYRE_AR_ABLATIONS"]
        if os.environ.get("GYRE_ATTN_ABLATIONS") and base == "kernels_attn.hip":    # tools/attn_ablate.py
            extra = [*extra, "-DGYRE_ATTN_ABLATIONS"]
        if os.environ.get("GYRE_GEMM_ABLATIONS"):      # tuning builds: the main-loop ablation tests of the pipelined kernel (tools/conv_ablate.py)
            extra = [*extra, "-DGYRE_GEMM_ABLATIONS"]
        cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed for {src} ({fl}):\n{r.stdout}\n{r.stderr}")
        if base in RESOURCE_FILES:
            _record_resources(base if fl == "bf16" else f"{fl}/{base}", r.stderr)
            return ""
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 8)) as ex:
            for warn in ex.map(cc, jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    for fl in flavours:
        objdir = os.path.join(HERE, "build") if fl == "bf16" else os.path.join(HERE, "build", fl)
        lib = LIB if fl == "bf16" else LIB_F16
        objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
        if any(j[2] == fl for j in jobs) or force or _stale(lib, objs):
            # -Bsymbolic: a process may map BOTH flavours (same symbol names); each library binds its own definitions
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", lib]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"link failed ({fl}):\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
