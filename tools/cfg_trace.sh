#!/bin/bash
# kernel-class timeline of one timed request of a bench configuration: tools/cfg_trace.sh <config> [extra bench args]
R=$PWD; O=$R/gpurun_out; mkdir -p $O; c=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ct
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o ct -- python $R/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --trace-markers "$@" > $O/trace_bench_$c.json 2> $O/trace_$c.err
python $R/tools/c5_trace.py /tmp/ct > $O/trace_$c.txt 2>&1
head -45 $O/trace_$c.txt
