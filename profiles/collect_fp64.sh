#!/bin/bash
# FP64 instructions executed by the evaluation kernel (SQ block, per class) -> flops per (candidate, step):
#   flops = 64 lanes x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64) per wavefront instruction, all lanes counted (a lane that is
#   masked off still occupies its slot of the FP64 pipe, which is what the "valu" roofline of bench.py prices).
# One pass (4 SQ counters), no trace domains.  usage (GPU box): bash profiles/collect_fp64.sh <workload> <steps> [mode]
set -e
WL=${1:-cfg5}; STEPS=${2:-6}; MODE=${3:-fused}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/fp64_${WL}_$MODE
# <workload>rb: the workload with its road boundary (bench.py --road-boundary)
BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > $OUT.bench.json 2> $OUT.err
cd $ROOT && python3 profiles/fp64_summary.py $WL $OUT $MODE $OUT.bench.json
