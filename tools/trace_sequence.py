"""Dev tool: launch-by-launch timeline of a UNet forward in the REAL stream (no HIP events between launches), from the
rocprofv3 kernel trace `tools/unet_gap.py run [B]` leaves behind:
  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/q -o q -- python tools/unet_gap.py run 2
  python tools/trace_sequence.py /tmp/q 20 > gpurun_out/seq_b2.txt
Every forward issues the same launch sequence, so position i of the 20 traced forwards is averaged: kernel time, idle gap in
front of it, and start-to-start time (= what the launch costs the call).  Sorted views: by position, and by class.
"""
import csv, glob, os, re, sys
from collections import defaultdict

d, evals = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
marks = [i for i, r in enumerate(rows) if "k_gap_marker" in r[2] or "FillFunctor<double>" in r[2]]
seg = rows[marks[-2] + 1:marks[-1]]
n = len(seg) // evals
assert n * evals == len(seg), (len(seg), evals)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", name)
    base, targs = m.group(1), (m.group(2) or "")
    return base + targs[:34]


dur, gap = [0.0] * n, [0.0] * n
for e in range(evals):
    for i in range(n):
        s, t, _ = seg[e * n + i]
        dur[i] += (t - s) / evals / 1e3
        if i or e:
            gap[i] += (s - seg[e * n + i - 1][1]) / (evals - (0 if i else 1)) / 1e3
names = [short(seg[i][2]) for i in range(n)]
for e in range(1, evals):
    assert all(short(seg[e * n + i][2]) == names[i] for i in range(n)), "launch sequence differs between forwards"
tot = sum(dur) + sum(gap)
print(f"{n} launches per forward, {tot / 1e3:.3f} ms per forward: kernels {sum(dur) / 1e3:.3f} ms, idle {sum(gap) / 1e3:.3f} ms")
print("\n# by class: launches, kernel us, gap us in front, share of the forward")
cls = defaultdict(lambda: [0, 0.0, 0.0])
for i in range(n):
    c = cls[names[i]]
    c[0] += 1; c[1] += dur[i]; c[2] += gap[i]
for k, c in sorted(cls.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:60s} {c[0]:4d} {c[1]:9.1f} {c[2]:9.1f} {100 * (c[1] + c[2]) / tot:6.2f} %")
print("\n# by position: index, gap us, kernel us, name")
for i in range(n):
    print(f"{i:4d} {gap[i]:7.2f} {dur[i]:8.2f}  {names[i]}")
