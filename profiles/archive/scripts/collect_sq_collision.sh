#!/bin/bash
# Instruction counts of the evaluation kernel with and without its collision test (profiles/probe_collision_cost.py).
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/sq_collision
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -- python3 $ROOT/profiles/probe_collision_cost.py "$@" > /dev/null 2>&1
cd $ROOT && python3 - <<'PY'
import csv, glob, collections, os
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sq_collision/a/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rp_eval_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) > 4096:
            vals[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(vals.items()):
    m = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    w = m.get("SQ_WAVES", 1)
    print(k[0].replace("void ", ""), "grid", k[1], "waves", int(w), " per wave:", {c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in m.items() if c != "SQ_WAVES"})
PY
