#!/bin/bash
# usage: pmc_run.sh <tag> <pmc_one.py args...>   -> prints per-kernel mean counters
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/b -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /dev/null 2>&1
python - <<PY
import csv,collections,glob
for d in "abc":
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if not (n.startswith("void k_") or n.startswith("k_")): continue
            acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
