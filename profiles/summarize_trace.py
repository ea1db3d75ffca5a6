"""Per-(kernel, grid size) duration statistics from a rocprofv3 --kernel-trace CSV.
usage: python profiles/summarize_trace.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    agg[(name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':60s} {'blocks':>7s} {'vgpr':>5s} {'lds':>6s} {'scr':>4s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for (name, blocks, vg, lds, scr), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:60]:60s} {blocks:7d} {vg:>5s} {lds:>6s} {scr:>4s} {len(d):6d} {sum(d)/len(d)/1e3:9.2f} {min(d)/1e3:9.2f} {max(d)/1e3:9.2f}")
