#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for shape in "4096 1280 1280" "16384 640 640" "65536 320 320"; do
  tag=lin_$(echo $shape | tr ' ' '_')
  echo "== $shape" >> gpurun_out/s3.log
  bash tools/pmc_run.sh $tag linear $shape >> gpurun_out/s3.log 2>&1
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/d -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py linear $shape > /dev/null 2>&1)
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/e -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one.py linear $shape > /dev/null 2>&1)
  python - <<PY >> gpurun_out/s3.log
import csv,collections,glob
for d in "de":
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if not (n.startswith("void k_") or n.startswith("k_")): continue
            acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
rm -rf gpurun_out/pmc_lin_*
