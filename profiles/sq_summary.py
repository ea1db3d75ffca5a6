"""Median SQ counter values of the main rp_eval_kernel launches (largest grid) -> per-wave instruction counts."""
import csv, glob, json, sys, collections
wl, out = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "rp_eval_kernel" in r["Kernel_Name"]]
        if not rows:
            continue
        gmax = max(int(r["Grid_Size"]) for r in rows)
        for r in rows:
            if int(r["Grid_Size"]) == gmax:
                vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in vals.items():
    res[k] = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    w = res[k].get("SQ_WAVES")
    if w:
        res[k]["per_wave"] = {c: round(v / w, 1) for c, v in res[k].items() if c.startswith("SQ_INSTS")}
print(json.dumps({"workload": wl, "kernels": res}, indent=1))
