"""rocprofv3 --kernel-trace CSV -> time per (kernel, grid size): usage  kernel_trace_summary.py <dir-or-csv> [top]"""
import collections, csv, glob, os, sys
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
g = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("qgd::(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        key = (name, int(r["Grid_Size_X"]))
        g[key][0] += 1; g[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in g.values())
print(f"total kernel time {tot / 1e3:.2f} ms")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0][:44]:44s} grid {k[1]:>10d} n {v[0]:6d} tot {v[1] / 1e3:9.2f} ms avg {v[1] / v[0]:8.1f} us {100 * v[1] / tot:5.1f}%")
