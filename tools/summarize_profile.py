"""Turn the raw rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the small tracked
summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 1 --warmup 1`
  <tag>_pmc_summary.csv    per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KB as reported) from the two --pmc passes
  traffic.json             HBM bytes per launch of every GEMM / attention kernel, corrected as MI355X_MICROARCH.md prescribes
                           (gfx950 FETCH_SIZE counts 128-B read requests as 64 B -> x2; WRITE_SIZE taken as reported)
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
csv.field_size_limit(1 << 30)

shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:80]


def pmc(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


fetch = pmc(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(src, "pmc_write", "bench_counter_collection.csv"), "WRITE_SIZE")
stats = {}
with open(os.path.join(src, "trace", "bench_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        stats[short(row["Name"])] = (int(row["Calls"]), float(row["TotalDurationNs"]), float(row["AverageNs"]), float(row["Percentage"]))

rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -stats.get(k, (0, 0, 0, 0))[1]):
    n = fetch.get(k, [0, 0])[0] or write.get(k, [0, 0])[0]
    fk = fetch[k][1] / max(fetch[k][0], 1) if k in fetch else 0.0
    wk = write[k][1] / max(write[k][0], 1) if k in write else 0.0
    st = stats.get(k, (0, 0, 0, 0))
    rows.append((k, st[0], st[2] / 1e3, st[3], n, fk, wk, (2 * fk + wk) * 1024))
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls_in_trace_run", "avg_us_in_trace_run", "pct_of_gpu_time", "launches_in_pmc_run",
                "mean_FETCH_SIZE_KB_raw", "mean_WRITE_SIZE_KB_raw", "hbm_bytes_per_launch_corrected(2*FETCH+WRITE)"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.2f}", f"{r[3]:.2f}", r[4], f"{r[5]:.1f}", f"{r[6]:.1f}", f"{r[7]:.0f}"])
# traffic.json: every MFMA kernel instantiation, keyed by its rocprof name, so bench.py can look up whichever kernel
# class is dominant in its own run (weighted over the instantiations that share the class prefix)
kern = {r[0]: {"hbm_bytes_per_launch": r[7], "mean_FETCH_SIZE_KB_raw": r[5], "mean_WRITE_SIZE_KB_raw": r[6],
               "launches_in_pmc_run": r[4], "avg_launch_us": r[2], "pct_of_gpu_time": r[3]}
        for r in rows if r[0].startswith("k_gemm") or r[0].startswith("k_attn")}
json.dump({"correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests as 64 B), WRITE_SIZE as reported, x1024",
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), tools/profile_bench.sh {tag}",
           "kernels": kern}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
for r in rows[:14]:
    print(f"{r[0][:60]:60s} calls={r[1]:6d} avg={r[2]:9.1f}us {r[3]:5.2f}%  fetchKB={r[5]:10.1f} writeKB={r[6]:10.1f} hbmMB/launch={r[7]/1e6:8.2f}")
