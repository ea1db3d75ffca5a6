"""Per-kernel medians of the counters collect_kernel_sq.sh collected (largest-grid launches of each kernel) + trace durations."""
import collections, csv, glob, os, sys
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in glob.glob(os.path.join(out, "[abc]", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = max(grid.get(k, 0), int(r["Grid_Size"]))
med = lambda v: sorted(v)[len(v) // 2]
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
for k in sorted(vals):
    c = {n: med(v) for n, v in vals[k].items()}
    w = c.get("SQ_WAVES", 0) or 1
    d = med(dur[k]) if dur.get(k) else float("nan")
    lines.append(f"{k[:70]:70s} grid {grid[k]:8d} dur {d:8.1f} us | waves {w:8.0f} | per wave: VALU {c.get('SQ_INSTS_VALU',0)/w:8.1f} SALU {c.get('SQ_INSTS_SALU',0)/w:7.1f} "
                 f"LDS {c.get('SQ_INSTS_LDS',0)/w:6.1f} VMEM_RD {c.get('SQ_INSTS_VMEM_RD',0)/w:6.1f} VMEM_WR {c.get('SQ_INSTS_VMEM_WR',0)/w:6.1f} SMEM {c.get('SQ_INSTS_SMEM',0)/w:6.1f} FLAT {c.get('SQ_INSTS_FLAT',0)/w:6.1f} | "
                 f"wave_cycles/wave {c.get('SQ_WAVE_CYCLES',0)/w:9.0f} busy_cycles {c.get('SQ_BUSY_CYCLES',0):9.0f} active_valu/wave {c.get('SQ_ACTIVE_INST_VALU',0)/w:8.0f} "
                 f"wait_inst_any/wave {c.get('SQ_WAIT_INST_ANY',0)/w:9.0f} wait_any/wave {c.get('SQ_WAIT_ANY',0)/w:9.0f} inst_cycles_vmem/wave {c.get('SQ_INST_CYCLES_VMEM',0)/w:8.0f} active_vmem/wave {c.get('SQ_ACTIVE_INST_VMEM',0)/w:8.0f} "
                 f"active_lds/wave {c.get('SQ_ACTIVE_INST_LDS',0)/w:7.0f} wait_lds/wave {c.get('SQ_WAIT_INST_LDS',0)/w:7.0f}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
