#!/bin/bash
# config-5 (ToMe + CLIP guidance) timeline: kernel trace of one timed request, summarised on the box
set -x
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5 -o c5 -- python $R/bench.py --config tomeclip --steps 1 --warmup 1 --no-cpu-baseline --trace-markers > $O/r06_c5_bench_under_trace.json 2> $O/r06_c5_err.log
python $R/tools/c5_trace.py /tmp/c5 1800 > $O/r06_c5_trace.txt 2>&1
head -30 $O/r06_c5_trace.txt
