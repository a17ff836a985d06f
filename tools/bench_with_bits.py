"""bench.py with planner tuning bits set first (gyre_debug_gemm_ablation is per thread; bench.py runs on the main thread):
python tools/bench_with_bits.py 0x800000 --config inpaint768 --steps 1 --warmup 1 --no-cpu-baseline --no-class-table"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gyre_amd import _lib
bits = int(sys.argv[1], 0)
_lib.lib().gyre_debug_gemm_ablation(bits)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
