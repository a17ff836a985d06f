#!/bin/bash
# Round 5 evidence: SQ / TCC / LDS counters (tools/pmc_run.sh: three separate --pmc passes, kernel-trace only) of the step's main kernel
# classes on their headline shapes -> gpurun_out/r05_pmc_kernels.txt.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE
# / 8 XCDs x 1024 SIMDs) ; L2 hit rate = TCC_HIT / TCC_REQ.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r05_pmc_kernels.txt; : > $out
run() { echo "== $*" >> $out; bash tools/pmc_run.sh k$RANDOM "$@" 2>&1 | grep -v "amdgpu.ids" >> $out; }
run conv 16 64 64 320 320
run conv 16 32 32 640 640
run conv 16 16 16 1280 1280
run attn 16 8 4096 4096 40 1
run attn 16 8 1024 1024 80 1
run linear 4096 1280 1280
run linear 16384 640 640
run linear 65536 320 320
run geglu 65536 320 1280
rm -rf gpurun_out/pmc_k*
python - <<'PY' >> $out
import re,ast
txt=open("gpurun_out/r05_pmc_kernels.txt").read()
print("\n== derived (per launch): MFMA busy share of the SIMD-cycles of the launch, L2 hit rate")
cur=None; acc={}
for line in txt.splitlines():
    if line.startswith("== "): cur=line[3:]; acc[cur]={}; continue
    m=re.match(r"(void )?(k_\S+.*?) (\{.*\})$", line)
    if m and cur:
        name=m.group(2)[:40]; d=ast.literal_eval(m.group(3)); acc[cur].setdefault(name,{}).update(d)
for shape,ks in acc.items():
    for name,d in ks.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            simd_cycles=d["GRBM_GUI_ACTIVE"]/8.0*1024
            hit=d.get("TCC_HIT_sum",0)/max(1,d.get("TCC_REQ_sum",1))
            print(f"{shape:34s} {name:40s} mfma_busy {d['SQ_VALU_MFMA_BUSY_CYCLES']/simd_cycles:5.3f}  l2_hit {hit:5.3f}  wait_any {d.get('SQ_WAIT_ANY',0)/max(1,d.get('SQ_WAVE_CYCLES',1)):5.3f}")
PY
