"""rocprofv3 --kernel-trace CSV -> per-kernel time PER STEP in the steady state: the launches before the `skip`-th launch of the marker kernel
(one that runs once per step) are dropped.  usage: steady_trace_summary.py <dir-or-csv> <marker substring> <skip> [top]
Prints launches per step, kernel time per step, and the span of the kept launches per step (kernel time + gaps)."""
import collections, csv, glob, os, sys
path, marker, skip = sys.argv[1], sys.argv[2], int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("qgd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) <= skip + 1:
    sys.exit(f"marker {marker!r}: {len(marks)} launches, cannot skip {skip}")
kept = rows[marks[skip]:marks[-1]]
steps = len(marks) - 1 - skip
g = collections.defaultdict(lambda: [0, 0.0])
for s, e, name in kept:
    g[name][0] += 1; g[name][1] += (e - s) / 1e3
tot = sum(v[1] for v in g.values())
span = (kept[-1][1] - kept[0][0]) / 1e3
print(f"{steps} steps kept: {len(kept) / steps:.1f} launches per step, kernel time {tot / steps / 1e3:.3f} ms per step, span {span / steps / 1e3:.3f} ms per step "
      f"(gaps {100 * (1 - tot / span):.1f} %)")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[:56]:56s} per step {v[0] / steps:6.1f} x {v[1] / v[0]:8.1f} us = {v[1] / steps / 1e3:7.3f} ms {100 * v[1] / tot:5.1f}%")
