#!/bin/bash
# rocprofv3 kernel trace of the headline timed region for one library build -> per-kernel duration summary.
# usage (GPU box): bash profiles/trace_lib.sh librp_amd.so [workload] [mode]
LIB=${1:-librp_amd.so}; WL=${2:-cfg2}; MODE=${3:-draw}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/trace_${LIB%.so}_${WL}_$MODE
export RP_AMD_LIBRARY=$ROOT/commonroad-reactive-planner_amd/lib/$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload $WL --mode $MODE --steps 50 --warmup 10 --min-seconds 0.1 --sequence 16 --main-only > /dev/null 2> $OUT.err
python3 $ROOT/profiles/summarize_trace.py $(ls $OUT/*/*kernel_trace.csv | head -1)
rm -rf $OUT
