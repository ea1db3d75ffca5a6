#!/bin/bash
# Round 4 diagnosis of the cfg3 draw step: (1) GPU-side step timeline from a rocprofv3 kernel trace (kernel durations and the gaps
# between them), (2) per-workgroup start / end distribution of the evaluation kernel (library built with -DRP_TIMELINE),
# (3) in-kernel phase stamps of one workgroup (-DRP_STAMPS).   usage (GPU box): bash profiles/r04_cfg3_diag.sh [workload] [mode]
WL=${1:-cfg3}; MODE=${2:-draw}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r04/diag_${WL}_$MODE; LIBDIR=$ROOT/commonroad-reactive-planner_amd/lib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --workload $WL --mode $MODE --steps 50 --warmup 10 --min-seconds 0.1 --sequence 16 --main-only > $OUT/bench_under_trace.json 2> $OUT/trace.err
TR=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 $ROOT/profiles/summarize_trace.py $TR > $OUT/trace_summary.txt
python3 $ROOT/profiles/step_timeline.py $TR > $OUT/step_timeline.txt
rm -rf $OUT/trace
cd $ROOT
[ -f $LIBDIR/librp_amd_tl.so ] && RP_AMD_LIBRARY=$LIBDIR/librp_amd_tl.so RP_AMD_PRINT_STAMPS=1 python3 profiles/probe_stamps_cfg3.py $WL > $OUT/wg_timeline.txt 2>&1
[ -f $LIBDIR/librp_amd_st.so ] && RP_AMD_LIBRARY=$LIBDIR/librp_amd_st.so RP_AMD_PRINT_STAMPS=1 python3 profiles/probe_stamps_cfg3.py $WL > $OUT/stamps.txt 2>&1
python3 bench.py --workload $WL --mode $MODE --no-cpu-baseline --no-configs --min-seconds 0.3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/step_timeline.txt $OUT/trace_summary.txt; tail -4 $OUT/wg_timeline.txt; tail -6 $OUT/stamps.txt
