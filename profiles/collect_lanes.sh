#!/bin/bash
# Lane utilisation of the evaluation kernel's vector instruction stream: thread-cycles of VALU work against
# the wave-level cycles it was issued in.  Separate --pmc passes, no trace domains.
#   SQ_INSTS_VALU           vector ALU instructions issued (wave level)
#   SQ_ACTIVE_INST_VALU     cycles waves spent executing VALU instructions (wave level, 4-cycle quads)
#   SQ_THREAD_CYCLES_VALU   the same counted per ACTIVE LANE: = SQ_ACTIVE_INST_VALU x 64 when every lane is on
#   (derived, as rocprof's VALUUtilization) lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
# usage (GPU box): bash profiles/collect_lanes.sh <workload> <steps> [mode]
set -e
WL=${1:-cfg3}; STEPS=${2:-20}; MODE=${3:-fused}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/lanes_${WL}_$MODE
# <workload>rb: the workload with its road boundary (bench.py --road-boundary)
BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > $OUT.bench.json 2> $OUT.err
cd $ROOT && python3 - "$WL" "$MODE" "$OUT" <<'PY'
import json, sys
sys.path[:0] = ["profiles"]
from _counters import by_kernel, main_kernel_of, rows_of
wl, mode, out = sys.argv[1:4]
vals = by_kernel(rows_of(f"{out}/a/*/*_counter_collection.csv"), main_kernel_of(out + ".bench.json"))
res = {}
for k, d in vals.items():
    m = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    if m.get("SQ_ACTIVE_INST_VALU"):
        m["lane_utilisation"] = round(m.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * m["SQ_ACTIVE_INST_VALU"]), 4)
    if m.get("SQ_WAVES"):
        m["valu_insts_per_wave"] = round(m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"], 1)
    res[k] = m
sys.path[:0] = [".", "commonroad-reactive-planner_amd"]
from commonroad_rp_amd import _capi
print(json.dumps({"workload": wl, "mode": mode, "source_hash": _capi.source_hash(), "kernels": res}, indent=1))
PY
rm -rf $OUT $OUT.bench.json
