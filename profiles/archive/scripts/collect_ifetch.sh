#!/bin/bash
# Instruction-fetch and wait counters of the eval kernel, separate passes, no trace domains.
# usage (GPU box): bash profiles/collect_ifetch.sh <workload> <steps> [mode]
set -e
WL=${1:-cfg2}; STEPS=${2:-20}; MODE=${3:-draw}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/ifetch_${WL}_$MODE
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --workload $WL --mode $MODE --steps $STEPS --warmup 3 --main-only > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --workload $WL --mode $MODE --steps $STEPS --warmup 3 --main-only > /dev/null 2>&1
cd $ROOT && python3 profiles/sq_summary.py $WL $OUT
