"""Merge rocprofv3 kernel and memory-copy traces into one timeline: start (us since the first record), duration, gap to the previous end.
usage: python profiles/timeline.py OUT_DIR [last_n]"""
import csv, glob, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
t0 = rows[0][0]; prev = None
for s, e, n in rows[-last:]:
    print(f"{(s - t0) / 1e3:12.1f} us  {(e - s) / 1e3:8.1f} us  gap {((s - prev) / 1e3 if prev else 0):7.1f}  {n}")
    prev = e
