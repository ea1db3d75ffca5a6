"""FP64 instruction classes of the launches that evaluate the batch (rp_eval_kernel or rp_cost_kernel, largest grid) -> flops per
(candidate, step); written to profiles/<round>_fp64_flops.json, which bench.py reads for the "valu" roofline of production mode."""
import json
import sys

from _counters import ROUND, by_kernel, main_kernel_of, median, rows_of, source_hash

wl, out, mode, bench = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
line = json.load(open(bench))
cand, n1 = float(line["config"]["candidates_per_step"]), int(line["config"]["horizon_steps"]) + 1
vals = by_kernel(rows_of(f"{out}/a/*/*_counter_collection.csv"), main_kernel_of(bench))
name, d = max(vals.items(), key=lambda kv: len(kv[1].get("SQ_WAVES", [])))
med = {c: median(v) for c, v in d.items()}
inst = {k: med.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")}
flops = 64.0 * (inst["ADD"] + inst["MUL"] + inst["TRANS"] + 2.0 * inst["FMA"])
rec = {"workload": wl, "mode": mode, "kernel": name, "candidates": cand, "steps": n1, "wave_instructions": inst, "waves": med.get("SQ_WAVES"),
       "flops_per_launch": flops, "flops_per_candidate_step": flops / (cand * n1),
       "model": "64 lanes x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) wavefront instructions counted by the SQ block (rocprofv3 --pmc), "
                "median launch, / (candidates x (N + 1))"}
print(json.dumps(rec))
path = f"profiles/{ROUND}_fp64_flops.json"
try:
    allr = json.load(open(path))
except Exception:
    allr = {}
rec["source_hash"] = source_hash()
allr[wl] = rec
json.dump(allr, open(path, "w"), indent=1)
