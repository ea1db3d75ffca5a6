#!/bin/bash
# rp_lon_kernel's broad phase with 4 / 8 (default) / 16 obstacle centres in flight (-DRP_LON_BROAD_BATCH): kernel-trace durations of the
# profile kernel and the step on cfg3 / cfg4 / cfg5obs, same box.   usage (GPU box): bash profiles/ab_lon_batch.sh <outdir>
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in default bb4 bb16; do
  for wl in cfg3 cfg4; do
    if [ $lib = default ]; then unset RP_AMD_LIBRARY; else export RP_AMD_LIBRARY=$ROOT/commonroad-reactive-planner_amd/lib/ab/librp_amd_$lib.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${lib}_$wl -- python3 $ROOT/bench.py --workload $wl --mode draw --steps 40 --warmup 10 --min-seconds 0.2 --sequence 16 --main-only > $OUT/${lib}_$wl.json 2> $OUT/${lib}_$wl.err
    python3 - $OUT/${lib}_$wl $lib $wl <<'PY'
import csv, glob, sys, collections, json
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
ms = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])["ms_per_step"]
print(sys.argv[2], sys.argv[3], "step %.1f us |" % (ms * 1e3), " ".join(f"{k} {med(v):.1f}" for k, v in sorted(d.items())))
PY
  done
done
