"""Median SQ counter values of the launches that evaluate the batch (rp_eval_kernel or rp_cost_kernel, largest grid) -> per-wave
instruction counts."""
import json
import sys

from _counters import by_kernel, main_kernel_of, median, rows_of, source_hash

wl, out = sys.argv[1], sys.argv[2]
vals = by_kernel(rows_of(f"{out}/a/*/*_counter_collection.csv") + rows_of(f"{out}/b/*/*_counter_collection.csv"),
                 main_kernel_of(sys.argv[3] if len(sys.argv) > 3 else ""))
res = {}
for k, d in vals.items():
    res[k] = {c: median(v) for c, v in d.items()}
    w = res[k].get("SQ_WAVES")
    if w:
        res[k]["per_wave"] = {c: round(v / w, 1) for c, v in res[k].items() if c.startswith("SQ_INSTS")}
print(json.dumps({"workload": wl, "source_hash": source_hash(), "kernels": res}, indent=1))
