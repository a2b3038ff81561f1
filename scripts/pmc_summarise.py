"""Fold the rocprofv3 outputs of scripts/collect_profiles.sh into the small files kept under profiles/:
<tag>_n<N>_kernel_stats.csv (copy of the --stats kernel table), <tag>_n<N>_pmc_tcc.json (per-kernel averages of the
TCC EA request counters and the HBM-side bytes derived from them) and pmc_traffic.json (what bench.py reads)."""
import csv
import glob
import json
import os
import shutil
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(out_dir, "summaries")
os.makedirs(prof, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return hits[0] if hits else None


traffic = {}
for n in (400, 200):
    stats = find(f"stats_n{n}/**/*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(prof, f"{tag}_n{n}_kernel_stats.csv"))
    per_kernel = {}
    for kind in ("rd", "wr"):
        path = find(f"pmc_{kind}_n{n}/**/*counter_collection.csv")
        if not path:
            continue
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                key = (name, row["Counter_Name"])
                acc.setdefault(key, []).append(float(row["Counter_Value"]))
        # one row per dispatch and counter (summed over XCDs by rocprofv3's *_sum derived counters)
        for (name, counter), vals in acc.items():
            per_kernel.setdefault(name, {})[counter] = sum(vals) / len(vals)
            per_kernel[name]["launches_" + kind] = len(vals)
    for name, c in per_kernel.items():
        if "TCC_EA0_RDREQ_sum" in c:
            other = c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_32B_sum", 0) - c.get("TCC_EA0_RDREQ_64B_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0)
            c["hbm_read_bytes"] = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * max(other, 0)
        if "TCC_EA0_WRREQ_sum" in c:
            c["hbm_write_bytes"] = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c["TCC_EA0_WRREQ_sum"] - c.get("TCC_EA0_WRREQ_64B_sum", 0))
    if per_kernel:
        json.dump(per_kernel, open(os.path.join(prof, f"{tag}_n{n}_pmc_tcc.json"), "w"), indent=1, sort_keys=True)
        # the face pass is the LDS-staged kernel plus the gather kernel on the tiles the former leaves out: one launch of each per step
        # (with QGD_FUSED, the default: ONE kernel, fusedFaceCellKernel, which is the cell update as well)
        face = [(name, c) for name, c in per_kernel.items() if ("faceFluxGvp3" in name or "fusedFaceCell" in name) and "hbm_read_bytes" in c and "hbm_write_bytes" in c]
        if face:
            rd = sum(c["hbm_read_bytes"] for _, c in face)
            wr = sum(c["hbm_write_bytes"] for _, c in face)
            traffic[f"n{n}_gpus1"] = {
                "kernel": " + ".join(sorted(name for name, _ in face)), "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": rd + wr,
                "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ{,_64B}_sum (separate passes, "
                          "scripts/collect_profiles.sh), bytes = sum(size*requests); FETCH_SIZE*1024 reads exactly half of the read "
                          f"side on gfx950 (x2 correction of MI355X_MICROARCH.md), WRITE_SIZE*1024 matches; profiles/{tag}_n{n}_pmc_tcc.json"}
        for name, c in per_kernel.items():
            for short, pattern in (("point", "pointInterpRecKernel"), ("cell", "cellUpdateKernel")):
                if pattern in name and "hbm_read_bytes" in c and "hbm_write_bytes" in c and f"n{n}_gpus1" in traffic:
                    traffic[f"n{n}_gpus1"][short + "_bytes_per_launch"] = c["hbm_read_bytes"] + c["hbm_write_bytes"]
                    traffic[f"n{n}_gpus1"][short + "_read_bytes"] = c["hbm_read_bytes"]
if traffic:
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("summaries:", sorted(os.listdir(prof)))
