#!/bin/bash
# rocprofv3 kernel trace of bench.py's corridor-sampling record (rp_plan_coeffs on a CorridorSampling level) -> per-kernel summary.
# usage (GPU box): bash profiles/trace_corridor.sh [workload]
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/trace_corridor
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload ${1:-cfg2} --steps 20 --warmup 5 --min-seconds 0.05 --sequence 8 --no-configs --no-cpu-baseline > /dev/null 2> $OUT.err
python3 $ROOT/profiles/summarize_trace.py $(ls $OUT/*/*kernel_trace.csv | head -1)
rm -rf $OUT
