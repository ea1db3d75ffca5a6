#!/bin/bash
# A/B of ONE library under a diagnostic environment switch, on one box, twice in alternation (cfg2 headline + side configs).
# usage (GPU box): bash profiles/ab_env.sh RP_AMD_NO_ROW_PADDING
cd $GRAFT_REPO_ROOT
SW=$1
for rep in 1 2; do
for val in "" 1; do
  if [ -n "$val" ]; then export $SW=1; tag="$SW=1"; else unset $SW; tag="default"; fi
  python bench.py --no-cpu-baseline --min-seconds 0.3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read())
out=['$tag'.ljust(26), 'cfg2 draw k=%.2f step=%.2f us | fused k=%.2f step=%.2f | plan p50 %.1f us' % (r['roofline']['kernel_ms']*1e3, r['ms_per_step']*1e3, r['fused_mode']['kernel_ms']*1e3, r['fused_mode']['ms_per_step']*1e3, r['plan_latency_ms']['p50']*1e3)]
for k,v in r.get('configs',{}).items(): out.append('%s draw k=%.3f fused k=%.3f ms' % (k, v['draw']['kernel_ms'], v['fused']['kernel_ms']))
print(' | '.join(out))"
done
done
