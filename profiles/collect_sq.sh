#!/bin/bash
# Instruction-issue counters of the eval kernel (SQ block), separate passes, no trace domains.
# usage (GPU box): bash profiles/collect_sq.sh <workload> <steps> [mode]
set -e
WL=${1:-cfg2}; STEPS=${2:-20}; MODE=${3:-draw}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/sq_${WL}_$MODE
# <workload>rb: the workload with its road boundary (bench.py --road-boundary)
BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > $OUT.bench.json 2> $OUT.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > /dev/null 2>&1
cd $ROOT && python3 profiles/sq_summary.py $WL $OUT $OUT.bench.json
