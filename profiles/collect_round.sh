#!/bin/bash
# One measurement pass for a round tag: bench lines, rocprofv3 kernel stats + trace summaries, PMC traffic, SQ counters.
# usage (GPU box): bash profiles/collect_round.sh r01g      -> gpurun_out/<tag>/...
set -e
TAG=${1:-r01x}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 bench.py --workload cfg1 --cpu-seconds 6 > $OUT/${TAG}_bench_cfg1.json 2> $OUT/bench_cfg1.err
for wl in cfg3 cfg4 cfg5; do
  python3 bench.py --workload $wl --steps 20 --warmup 3 --cpu-seconds 6 > $OUT/${TAG}_bench_$wl.json 2> $OUT/bench_$wl.err
done
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg5; do
  steps=200; [ $wl = cfg5 ] && steps=20
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -- python3 $ROOT/bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cp $(ls $OUT/prof_$wl/*/*kernel_stats.csv | head -1) $OUT/${TAG}_${wl}_kernel_stats.csv
  python3 $ROOT/profiles/summarize_trace.py $(ls $OUT/prof_$wl/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_${wl}_kernel_trace_summary.txt
  rm -rf $OUT/prof_$wl
done
cd $ROOT
bash profiles/collect_pmc.sh cfg2 20 > $OUT/pmc_cfg2.json 2> $OUT/pmc_cfg2.err
bash profiles/collect_pmc.sh cfg5 8 > $OUT/pmc_cfg5.json 2> $OUT/pmc_cfg5.err
cp profiles/r01_pmc_traffic.json $OUT/r01_pmc_traffic.json
bash profiles/collect_sq.sh cfg2 20 draw > $OUT/${TAG}_sq_cfg2.json 2> $OUT/sq_cfg2.err
bash profiles/collect_sq.sh cfg5 6 fused > $OUT/${TAG}_sq_cfg5_fused.json 2> $OUT/sq_cfg5.err
rm -rf $ROOT/gpurun_out/pmc_cfg2 $ROOT/gpurun_out/pmc_cfg5 $ROOT/gpurun_out/sq_cfg2_draw $ROOT/gpurun_out/sq_cfg5_fused
echo done; ls $OUT
