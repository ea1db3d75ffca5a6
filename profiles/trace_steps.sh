#!/bin/bash
# Step timelines (profiles/step_timeline.py) of workloads from rocprofv3 kernel traces of the headline region.
# usage (GPU box): bash profiles/trace_steps.sh <tag> <workload[:mode]> ...
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  WL=${spec%%:*}; MODE=${spec#*:}; [ "$MODE" = "$spec" ] && MODE=draw
  BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$WL -- python3 $ROOT/bench.py --workload $BASE $RB --mode $MODE --steps 50 --warmup 10 --min-seconds 0.1 --sequence 16 --main-only > /dev/null 2> $OUT/trace_$WL.err
  echo "== $WL $MODE" >> $OUT/step_timelines.txt
  python3 $ROOT/profiles/step_timeline.py $(ls $OUT/trace_$WL/*/*kernel_trace.csv | head -1) >> $OUT/step_timelines.txt
  rm -rf $OUT/trace_$WL
done
cat $OUT/step_timelines.txt
