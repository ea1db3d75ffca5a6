"""GPU-side timeline of bench steps from a rocprofv3 kernel trace: per step the kernels from the first one (rp_lon_kernel, or the
evaluation kernel of the single-launch path) to the selection epilogue, the gaps between them, and the idle gap between steps
(ticket -> host -> next launch -> first wavefront).   usage: python profiles/step_timeline.py <kernel_trace.csv>"""
import csv, sys, collections, numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def label(r):
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
    one = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) == 1
    # (a one-workgroup evaluation launch behind the epilogue is the re-evaluation of a production-mode plan's winner: part of ITS step)
    return "rp_eval_kernel (winner)" if (name == "rp_eval_kernel" and one) else name
ev = [(label(r), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
steps, cur = [], []
for k, (name, s, e) in enumerate(ev):
    cur.append((name, s, e))
    nxt = ev[k + 1][0] if k + 1 < len(ev) else ""
    if (name in ("rp_finalize_kernel", "rp_select_kernel") and nxt != "rp_eval_kernel (winner)") or \
            (name == "rp_eval_kernel (winner)" and len(cur) > 1 and cur[-2][0] in ("rp_finalize_kernel", "rp_select_kernel")):
        steps.append(cur); cur = []
steps = steps[len(steps) // 4:]   # drop warm-up
# the bench's headline step keeps state rows and has no winner pass: the most common sequence WITHOUT one (argument "production": with);
# others (cost-ordered rounds, the other mode) are reported by count
want_winner = len(sys.argv) > 2 and sys.argv[2] == "production"
shapes = collections.Counter(tuple(k[0] for k in st) for st in steps)
shape, n = next((sh, c) for sh, c in shapes.most_common() if (sh[-1] == "rp_eval_kernel (winner)") == want_winner)
sel = [st for st in steps if tuple(k[0] for k in st) == shape]
us = lambda a: float(np.median(a)) / 1e3
print(f"steps {len(steps)}, of which {n} are {' -> '.join(shape)}")
for i, name in enumerate(shape):
    print(f"  {name:24s} {us([st[i][2] - st[i][1] for st in sel]):8.2f} us")
    if i + 1 < len(shape):
        print(f"  {'(gap)':24s} {us([st[i + 1][1] - st[i][2] for st in sel]):8.2f} us")
print(f"  gpu span                 {us([st[-1][2] - st[0][1] for st in sel]):8.2f} us")
idx = [i for i, st in enumerate(steps) if tuple(k[0] for k in st) == shape]
between = [steps[i + 1][0][1] - steps[i][-1][2] for i in idx if i + 1 < len(steps)]
period = [steps[i + 1][0][1] - steps[i][0][1] for i in idx if i + 1 < len(steps)]
print(f"  between steps            {us(between):8.2f} us   (epilogue end -> next step's first kernel)")
print(f"  period                   {us(period):8.2f} us")
