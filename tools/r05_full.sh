#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_bench_mid.json 2> gpurun_out/r05_bench_mid.err
