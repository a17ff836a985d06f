#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (1 step)
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, own runs, kernel-trace only) on a shortened sampler loop
# Outputs land in gpurun_out/prof_$TAG/; tools/summarize_profile.py turns them into profiles/*.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-class-table > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --inference-steps 3 --no-cpu-baseline --no-class-table > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --inference-steps 3 --no-cpu-baseline --no-class-table > $OUT/bench_write.json 2> $OUT/write.log
ls -R $OUT | head -40
# keep only the small files (the merged directory is capped at 64 MiB)
find $OUT -name "*.csv" -size +30M -print -exec sh -c 'head -c 30000000 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh $OUT
