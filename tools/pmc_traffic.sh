#!/bin/bash
# usage: pmc_traffic.sh <tag> <pmc_one.py args...>  -> mean FETCH_SIZE / WRITE_SIZE (KB, raw) per kernel of one op shape
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmct_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /dev/null 2>&1
python - <<PY
import csv,collections,glob
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in "fw":
    for f in glob.glob("$OUT/%s/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if not (n.startswith("void k_") or n.startswith("k_")): continue
            acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    m={c: sum(x)/len(x) for c,x in v.items()}
    f_,w_=m.get("FETCH_SIZE",0),m.get("WRITE_SIZE",0)
    print(k, "FETCH_KB %.0f WRITE_KB %.0f  corrected HBM bytes (2*FETCH+WRITE, guide) %.1f MB"%(f_,w_,(2*f_+w_)*1024/1e6))
PY
