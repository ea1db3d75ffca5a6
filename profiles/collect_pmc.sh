#!/bin/bash
# HBM traffic of the evaluation kernel (rp_eval_kernel / rp_cost_kernel) and the kernels around it from PMC counters (separate passes for WRITE_SIZE and FETCH_SIZE, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; no trace domains combined with --pmc).
# usage (GPU box): bash profiles/collect_pmc.sh <workload> <steps> [mode]
set -e
WL=${1:-cfg2}; STEPS=${2:-20}; MODE=${3:-draw}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_${WL}_$MODE
# <workload>rb: the workload with its road boundary (bench.py --road-boundary)
BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > $OUT.bench.json 2> $OUT.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps $STEPS --warmup 3 --min-seconds 0 --sequence 8 --main-only > /dev/null 2>&1
cd $ROOT && python3 profiles/pmc_summary.py $WL $OUT $MODE $OUT.bench.json
