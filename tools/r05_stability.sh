#!/bin/bash
# Stability evidence the round-4 review asked for: tests/test_gpu_threads.py 20 times in a row, and the two-handle thread probe
# (12 threaded UNet calls per repetition) for >= 1000 calls; run through gpurun, writes gpurun_out/r05_stability.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_stability.txt; : > $O
ok=0
for i in $(seq 1 20); do
  if python -m pytest tests/test_gpu_threads.py -x -q -m gpu -p no:cacheprovider > /tmp/thr.log 2>&1; then ok=$((ok+1)); else tail -20 /tmp/thr.log >> $O; fi
  echo "run $i: $(tail -1 /tmp/thr.log)" >> $O
done
echo "test_gpu_threads.py green $ok / 20" >> $O
REPS=90 python tools/thread_probe.py > /tmp/probe.log 2>&1
grep -c "mismatches \[\]" /tmp/probe.log | sed 's/^/thread probe: repetitions without a mismatch: /' >> $O
grep -v "mismatches \[\]" /tmp/probe.log | tail -5 >> $O
tail -3 $O
