cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 ) > gpurun_out/r05_gpu_tests.log 2>&1
