#!/bin/bash
# timing experiments on the attention-backward dK/dV kernel (GYRE_ABW_ABL variants give wrong numbers by design).
# Needs a library built with the variants: add "-DGYRE_ABW_ABLATIONS" to FLAGS in gyre_amd/build.py and rebuild (force=True).
cd /tmp; export TMPDIR=/tmp
for a in ${ABLS:-0 1 2 3 4 12 15}; do
  rm -rf /tmp/ab; GYRE_ABW_ABL=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab -o ab -- python /root/repo/tools/attn_bwd_bench.py > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/ab/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "dkv_dma" in r["Name"]: print("ABL $a", r["Name"][:50], round(float(r["AverageNs"])/1e3,1))
PY
done
