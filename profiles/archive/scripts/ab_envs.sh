#!/bin/bash
# A/B of ONE library under environment settings, on one box, twice in alternation: step and kernel time of one workload in draw and
# production mode.   usage (GPU box): bash profiles/ab_envs.sh <workload> "" "RP_AMD_X=1" "RP_AMD_Y=2 RP_AMD_Z=3" ...
cd $GRAFT_REPO_ROOT
WL=$1; shift
for rep in 1 2; do
for setting in "$@"; do
  env $setting python bench.py --workload $WL --no-cpu-baseline --no-configs --min-seconds 0.3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); f=r['fused_mode']
print('${setting:-default}'.ljust(34), '$WL draw step=%.1f us kernel=%.1f us (%s) frac=%.3f | production step=%.1f us kernel=%.1f us (%s) | plan p50 %s' % (r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3, r['roofline']['kernel'], r['roofline']['frac'], f['ms_per_step']*1e3, f['kernel_ms']*1e3, f['roofline']['kernel'], r['plan_latency_ms'] and round(r['plan_latency_ms']['p50']*1e3,1)))"
done
done
