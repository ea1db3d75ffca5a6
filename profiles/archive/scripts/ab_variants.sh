#!/bin/bash
# A/B of library builds on ONE box (box-to-box differences are larger than most kernel changes): for every library
# named on the command line (files under commonroad-reactive-planner_amd/lib/), twice in alternation, the bench line's
# kernel and step times of cfg2 (headline) and of cfg3 / cfg4 / cfg5 in both modes.
# usage (GPU box): bash profiles/ab_variants.sh librp_amd_base.so librp_amd.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  RP_AMD_LIBRARY=$GRAFT_REPO_ROOT/commonroad-reactive-planner_amd/lib/$lib python bench.py --no-cpu-baseline --min-seconds 0.3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read())
out=['$lib'.ljust(22), 'cfg2 draw k=%.2f step=%.2f us | fused k=%.2f step=%.2f | plan p50 %.1f us' % (r['roofline']['kernel_ms']*1e3, r['ms_per_step']*1e3, r['fused_mode']['kernel_ms']*1e3, r['fused_mode']['ms_per_step']*1e3, r['plan_latency_ms']['p50']*1e3)]
for k,v in r.get('configs',{}).items(): out.append('%s draw k=%.3f fused k=%.3f ms' % (k, v['draw']['kernel_ms'], v['fused']['kernel_ms']))
print(' | '.join(out))"
done
done
