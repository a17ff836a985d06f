"""Dev tool: where a CLIP-guided request (bench.py --config tomeclip) spends its wall time - GPU busy vs idle, and which launches the
idle time sits in front of.  Input: a rocprofv3 --kernel-trace csv directory of `bench.py --config tomeclip --steps 1 --warmup 1`.
  cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/c5 -o c5 -- python bench.py --config tomeclip --steps 1 --warmup 1 --no-cpu-baseline --trace-markers
  python tools/c5_trace.py /tmp/c5 > gpurun_out/r06_c5_trace.txt
Takes the timed region (between the two fill<complex128> markers of bench.py --trace-markers), prints busy / idle, the classes by busy time and by idle time in
front of them, and one guided step launch by launch (gap, duration, name) between two k_tome_sort-led UNet forwards."""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
marks = [i for i, r in enumerate(rows) if "FillFunctor<c10::complex<double>" in r[2]]
seg = rows[marks[0] + 1:marks[1]]           # bench.py --trace-markers: the timed region


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    if not m:
        return name[:70]
    base, targs = m.group(1), (m.group(2) or "")
    return (base + targs)[:70]


def native(name):
    return re.match(r"^(void )?k_", name) is not None


span = seg[-1][1] - seg[0][0]
busy = sum(t - s for s, t, _ in seg)
print(f"{len(seg)} launches in {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.1f} ms")
cls = defaultdict(lambda: [0, 0.0, 0.0])
end = seg[0][0]
gaps = []
for s, t, n in seg:
    c = cls[short(n)]
    g = max(0, s - end)
    c[0] += 1; c[1] += t - s; c[2] += g
    gaps.append(g)
    end = max(end, t)
nb = sum(c[1] for k, c in cls.items() if k.startswith("k_")); ng = sum(c[2] for k, c in cls.items() if k.startswith("k_"))
nl = sum(c[0] for k, c in cls.items() if k.startswith("k_"))
print(f"native kernels: {nl} launches, busy {nb / 1e6:.1f} ms, idle in front {ng / 1e6:.1f} ms; "
      f"others: {len(seg) - nl} launches, busy {(busy - nb) / 1e6:.1f} ms, idle in front {(span - busy - ng) / 1e6:.1f} ms")
print("\n# classes by busy + idle-in-front: launches, busy ms, idle ms")
for k, c in sorted(cls.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:70]:
    print(f"{k:72s} {c[0]:6d} {c[1] / 1e6:9.2f} {c[2] / 1e6:9.2f}")
# one guided step: from one 'k_tome_sort'-free stretch start ... simply the last 1/20 of the segment, launch by launch
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    print(f"\n# last {n} launches: gap us, kernel us, name")
    for (s, t, nme), g in list(zip(seg, gaps))[-n:]:
        print(f"{g / 1e3:8.1f} {(t - s) / 1e3:8.1f}  {short(nme)}")
