"""Where the HOST spends a bench step: cProfile around one bench.py step of a config (after one warm-up), top functions by
cumulative and by own time.  python tools/host_profile.py --config tomeclip --inference-steps 20
(the numbers include cProfile's own overhead: use them for shares, not for absolute times)"""
import cProfile, io, os, pstats, runpy, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
real_sync = torch.cuda.synchronize
state = {"steps": 0, "prof": None}
def sync_hook(*a, **k):
    # bench.py synchronises after every timed step: the profiler starts after the warm-up's barrier and stops at the first timed step's
    return real_sync(*a, **k)
torch.cuda.synchronize = sync_hook
prof = cProfile.Profile()
argv = [a for a in sys.argv[1:]]
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-class-table"] + argv
import bench as _b    # noqa: E402  (imports only)
if os.environ.get("BITS"):       # planner tuning bits (per thread; bench runs on this one)
    from gyre_amd import _lib as _l
    _l.lib().gyre_debug_gemm_ablation(int(os.environ["BITS"], 0))
orig_perf = time.perf_counter
src = open(_b.__file__).read()
# run main() with the profiler wrapped around the timed loop: the warm-up ends with barrier(); enable there
src = src.replace("    t0 = time.perf_counter()\n    for i in range(args.steps):", "    import cProfile as _cp; _pr = _cp.Profile(); _pr.enable()\n    t0 = time.perf_counter()\n    for i in range(args.steps):")
src = src.replace("    barrier()\n    elapsed = time.perf_counter() - t0", "    barrier()\n    elapsed = time.perf_counter() - t0\n    _pr.disable(); globals()['_PROFILE'] = _pr")
g = {"__name__": "bench_profiled", "__file__": _b.__file__}
exec(compile(src, _b.__file__, "exec"), g)
g["main"]()
pr = g.get("_PROFILE")
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000], file=sys.stderr)
