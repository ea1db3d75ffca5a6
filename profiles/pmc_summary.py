"""Median WRITE_SIZE / FETCH_SIZE (KB) of the launches that evaluate the batch (rp_eval_kernel or rp_cost_kernel, largest grid) and
of the kernels around them (rp_lon_kernel, rp_select_kernel / rp_finalize_kernel) -> bytes per launch.
The variant is picked ONCE -- the instance with the largest median WRITE_SIZE (the bench's timed region runs the state-writing
variant; the winner pass and the cost-ordered rounds are other instances or smaller grids) -- and both counters are read from that
same named instance.
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM);
WRITE_SIZE is taken as is (exact for 16-B-per-lane streaming stores; 8-B stores are uncalibrated per the guide)."""
import json
import sys

from _counters import ROUND, SIDE_KERNELS, by_kernel, main_kernel_of, median, rows_of, source_hash

wl, out = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "draw"
family = main_kernel_of(sys.argv[4] if len(sys.argv) > 4 else "")   # (the bench line of the WRITE_SIZE pass)
passes = {"WRITE_SIZE": rows_of(f"{out}/w/*/*_counter_collection.csv"), "FETCH_SIZE": rows_of(f"{out}/r/*/*_counter_collection.csv")}
wv = by_kernel(passes["WRITE_SIZE"], family)
name = max(wv, key=lambda k: median(wv[k]["WRITE_SIZE"]))
rv = by_kernel(passes["FETCH_SIZE"], family)
if name not in rv:
    raise SystemExit(f"{name} has no FETCH_SIZE launches of the largest grid: {sorted(rv)}")
res = {"WRITE_SIZE": {"kernel": name, "median_KB": median(wv[name]["WRITE_SIZE"]), "n": len(wv[name]["WRITE_SIZE"])},
       "FETCH_SIZE": {"kernel": name, "median_KB": median(rv[name]["FETCH_SIZE"]), "n": len(rv[name]["FETCH_SIZE"])}}
w = res["WRITE_SIZE"]["median_KB"] * 1024.0
r = res["FETCH_SIZE"]["median_KB"] * 1024.0 * 2.0
# the other kernels of a step: per instance, median bytes per launch (same corrections), every grid size
side = {}
for tag, factor in (("WRITE_SIZE", 1.0), ("FETCH_SIZE", 2.0)):
    for k, d in by_kernel(passes[tag], SIDE_KERNELS, largest_grid=False).items():
        side.setdefault(k, {})[tag.lower().replace("_size", "_bytes")] = median(d[tag]) * 1024.0 * factor
        side[k]["launches"] = len(d[tag])
summary = {"workload": wl, "mode": mode, "kernel": name, "write_bytes": w, "fetch_bytes_corrected_x2": r, "traffic_bytes": w + r,
           "detail": res, "other_kernels": side}
print(json.dumps(summary))
path = f"profiles/{ROUND}_pmc_traffic.json"   # read back by bench.py (roofline.traffic), key "<workload>:<mode>"
try:
    allr = json.load(open(path))
except Exception:
    allr = {}
summary["source_hash"] = source_hash()
allr[f"{wl}:{mode}"] = summary
json.dump(allr, open(path, "w"), indent=1)
