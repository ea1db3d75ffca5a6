"""GPU-side timeline of bench steps from a rocprofv3 kernel trace: span lon_start -> finalize_end per step,
gaps between kernels inside a step, and idle gap between steps (host-side overhead).
usage: python profiles/step_timeline.py <kernel_trace.csv>"""
import csv, sys, numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
       int(r["Grid_Size_X"])) for r in rows]
steps, cur = [], None
for name, s, e, g in ev:
    if name == "rp_lon_kernel":
        cur = {"lon": (s, e)}
    elif cur is not None and name == "rp_eval_kernel" and "eval" not in cur:
        cur["eval"] = (s, e)
    elif cur is not None and name in ("rp_finalize_kernel", "rp_select_kernel"):
        cur["fin"] = (s, e); steps.append(cur); cur = None
steps = steps[len(steps) // 4:]   # drop warm-up
def us(a): return np.median(a) / 1e3
print("steps", len(steps))
print("lon      %.2f us" % us([s["lon"][1] - s["lon"][0] for s in steps]))
print("gap1     %.2f us" % us([s["eval"][0] - s["lon"][1] for s in steps]))
print("eval     %.2f us" % us([s["eval"][1] - s["eval"][0] for s in steps]))
print("gap2     %.2f us" % us([s["fin"][0] - s["eval"][1] for s in steps]))
print("finalize %.2f us" % us([s["fin"][1] - s["fin"][0] for s in steps]))
print("gpu span %.2f us" % us([s["fin"][1] - s["lon"][0] for s in steps]))
print("between steps (finalize end -> next lon start) %.2f us" % us([b["lon"][0] - a["fin"][1] for a, b in zip(steps[:-1], steps[1:])]))
print("period   %.2f us" % us([b["lon"][0] - a["lon"][0] for a, b in zip(steps[:-1], steps[1:])]))
