#!/bin/bash
# One measurement pass for a round tag: the default bench line, rocprofv3 kernel stats + trace summary of the SAME command,
# PMC traffic (WRITE_SIZE / FETCH_SIZE, separate passes) and FP64 instruction counters per workload.
# usage (GPU box): bash profiles/collect_round.sh r03a [quick]      -> gpurun_out/<tag>/...
set -e
TAG=${1:-r02x}; QUICK=${2:-}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# counters first: the bench line below then carries `traffic` and the FP64 fraction of this very code (bench.py reports the
# numbers of profiles/r05_*.json only when their source hash is the library's)
if [ -z "$QUICK" ]; then
for wl in cfg1 cfg2 cfg2rb cfg3 cfg3rb cfg3f cfg3frb cfg4 cfg4rb cfg5; do
  steps=20; [ $wl = cfg5 ] && steps=6; [ $wl = cfg4 ] && steps=8; [ $wl = cfg4rb ] && steps=8
  bash profiles/collect_pmc.sh $wl $steps draw > $OUT/pmc_${wl}_draw.json 2> $OUT/pmc_${wl}_draw.err
  bash profiles/collect_fp64.sh $wl $steps fused > $OUT/fp64_${wl}.json 2> $OUT/fp64_${wl}.err
  rm -rf $ROOT/gpurun_out/pmc_${wl}_draw $ROOT/gpurun_out/fp64_${wl}_fused
  echo "$wl counters done"
done
cp profiles/r05_pmc_traffic.json profiles/r05_fp64_flops.json $OUT/
fi
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cp gpurun_out/bench_detail_n1.json $OUT/${TAG}_bench_detail.json
# cfg2 as the headline workload (the headline of rounds 1-2; carries plan() latency on cfg2)
python3 bench.py --workload cfg2 --no-configs --no-cpu-baseline > $OUT/${TAG}_bench_cfg2.json 2>> $OUT/bench.err
cp gpurun_out/bench_detail_n1.json $OUT/${TAG}_bench_cfg2_detail.json
# plan() in closed loop with the cycle's call made by the binding's extension module / through ctypes, on this box
python3 profiles/probe_plan_latency_r05.py cfg2 cfg1 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_plan_latency_ab.txt
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof.err
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $ROOT/profiles/summarize_trace.py $(ls $OUT/prof/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_kernel_trace_summary.txt
rm -rf $OUT/prof
# the headline alone (no side configurations: cfg3rb / cfg3frb launch the same kernel instance on the same grid as the headline, so
# the averages of the full line's trace mix three workloads): the kernel durations bench.py's headline `kernel_ms` has to agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline --no-configs > $OUT/${TAG}_headline_bench_under_rocprof.json 2>> $OUT/prof.err
python3 $ROOT/profiles/summarize_trace.py $(ls $OUT/prof/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_headline_kernel_trace_summary.txt
python3 $ROOT/profiles/step_timeline.py $(ls $OUT/prof/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_headline_step_timeline.txt 2>&1 || true
python3 $ROOT/profiles/step_timeline.py $(ls $OUT/prof/*/*kernel_trace.csv | head -1) production > $OUT/${TAG}_production_step_timeline.txt 2>&1 || true
rm -rf $OUT/prof
echo "trace done"
cd $ROOT
[ -n "$QUICK" ] && exit 0
bash profiles/collect_sq.sh cfg3 20 draw > $OUT/${TAG}_sq_cfg3.json 2> $OUT/sq_cfg3.err
bash profiles/collect_sq.sh cfg5 6 fused > $OUT/${TAG}_sq_cfg5_fused.json 2> $OUT/sq_cfg5.err
rm -rf $ROOT/gpurun_out/sq_cfg3_draw $ROOT/gpurun_out/sq_cfg5_fused
echo done; ls $OUT
