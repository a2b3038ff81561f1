"""Folds the counter passes of scripts/collect_secondary_pmc.sh into <tag>_pmc_secondary.json: per kernel the average TCC EA requests of
the LAST 30 dispatches of that name (the launches of the measurement entry) and the HBM-side bytes derived from them."""
import csv, glob, json, os, sys
out_dir, tag = sys.argv[1], sys.argv[2]
WANT = {"qhd": "mgSmoothKernel<float>", "qhd_c5": "mgSmoothKernel<float>", "implicit": "iChebKernel<3, 0>" if os.environ.get("QGD_IMPL_SOLVER", "cheb") == "cheb" else "iApplyKernel<3, 1>"}
res = {}
for which, pattern in WANT.items():
    c = {}
    for kind in ("rd", "wr"):
        hits = glob.glob(os.path.join(out_dir, f"{kind}_{which}/**/*counter_collection.csv"), recursive=True)
        if not hits:
            continue
        rows = [r for r in csv.DictReader(open(hits[0])) if pattern in r["Kernel_Name"]]
        by_counter = {}
        for r in rows:
            by_counter.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, vals in by_counter.items():
            vals.sort()
            last = [v for _, v in vals[-30:]]
            c[name] = sum(last) / len(last)
            c["dispatches_" + kind] = len(last)
    if "TCC_EA0_RDREQ_sum" in c:
        other = c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_32B_sum", 0) - c.get("TCC_EA0_RDREQ_64B_sum", 0) - c.get("TCC_EA0_RDREQ_128B_sum", 0)
        c["hbm_read_bytes"] = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * max(other, 0)
    if "TCC_EA0_WRREQ_sum" in c:
        c["hbm_write_bytes"] = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c["TCC_EA0_WRREQ_sum"] - c.get("TCC_EA0_WRREQ_64B_sum", 0))
    if "hbm_read_bytes" in c and "hbm_write_bytes" in c:
        c["bytes_per_launch"] = c["hbm_read_bytes"] + c["hbm_write_bytes"]
    c["kernel"] = pattern
    res[which if which == "qhd_c5" else which + "_n200"] = c
path = os.path.join(out_dir, f"{tag}_pmc_secondary.json")
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
