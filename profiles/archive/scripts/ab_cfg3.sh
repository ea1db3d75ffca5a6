#!/bin/bash
# A/B of library builds on ONE box, large-batch step: for every library named on the command line (files under
# commonroad-reactive-planner_amd/lib/), twice in alternation, step and kernel time of one workload (default cfg3) in draw and
# production mode.   usage (GPU box): bash profiles/ab_cfg3.sh [-w cfg4] librp_amd_r02.so librp_amd.so
cd $GRAFT_REPO_ROOT
WL=cfg3
if [ "$1" = "-w" ]; then WL=$2; shift 2; fi
BASE=$WL; RB=""; case $WL in *rb) BASE=${WL%rb}; RB=--road-boundary;; esac   # <workload>rb: with its road boundary
for rep in 1 2; do
for lib in "$@"; do
  RP_AMD_LIBRARY=$GRAFT_REPO_ROOT/commonroad-reactive-planner_amd/lib/$lib python bench.py --workload $BASE $RB --no-cpu-baseline --no-configs --min-seconds 0.3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); f=r['fused_mode']
print('$lib'.ljust(24), '$WL draw step=%.1f us kernel=%.1f us frac=%.3f | production step=%.1f us kernel=%.1f us | plan p50 %s us' % (r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3, r['roofline']['frac'], f['ms_per_step']*1e3, f['kernel_ms']*1e3, r['plan_latency_ms'] and round(r['plan_latency_ms']['p50']*1e3, 1)))"
done
done
