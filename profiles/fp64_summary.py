"""FP64 instruction classes of the main rp_eval_kernel launches (largest grid) -> flops per (candidate, step);
written to profiles/r03_fp64_flops.json, which bench.py reads for the "valu" roofline of production mode."""
import collections, csv, glob, json, os, sys


def _source_hash():
    """hash of the sources of the library these counters were measured on (rp_source_hash)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "commonroad-reactive-planner_amd")]
    from commonroad_rp_amd import _capi
    return _capi.source_hash()

wl, out, mode, bench = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
line = json.load(open(bench))
cand, n1 = float(line["config"]["candidates_per_step"]), int(line["config"]["horizon_steps"]) + 1
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/a/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "rp_eval_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
name, d = max(vals.items(), key=lambda kv: len(kv[1].get("SQ_WAVES", [])))
med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
inst = {k: med.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")}
flops = 64.0 * (inst["ADD"] + inst["MUL"] + inst["TRANS"] + 2.0 * inst["FMA"])
rec = {"workload": wl, "mode": mode, "kernel": name, "candidates": cand, "steps": n1, "wave_instructions": inst, "waves": med.get("SQ_WAVES"),
       "flops_per_launch": flops, "flops_per_candidate_step": flops / (cand * n1),
       "model": "64 lanes x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) wavefront instructions counted by the SQ block (rocprofv3 --pmc), "
                "median launch, / (candidates x (N + 1))"}
print(json.dumps(rec))
path = "profiles/r03_fp64_flops.json"
try:
    allr = json.load(open(path))
except Exception:
    allr = {}
rec["source_hash"] = _source_hash()
allr[wl] = rec
json.dump(allr, open(path, "w"), indent=1)
