#!/bin/bash
# A/B on one box: threads per workgroup of the evaluation kernel (RP_AMD_EVAL_BLOCK=256|64), step and kernel time per workload,
# draw and production mode, twice in alternation.   usage (GPU box): bash profiles/ab_block.sh cfg3 [cfg4 ...]
cd $GRAFT_REPO_ROOT
for WL in "$@"; do
for rep in 1 2; do
for blk in 256 64; do
  RP_AMD_EVAL_BLOCK=$blk python bench.py --workload $WL --no-cpu-baseline --no-configs --min-seconds 0.3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); f=r['fused_mode']
print('block $blk'.ljust(12), '$WL draw step=%.1f us kernel=%.1f us frac=%.3f | production step=%.1f us kernel=%.1f us' % (r['ms_per_step']*1e3, r['roofline']['kernel_ms']*1e3, r['roofline']['frac'], f['ms_per_step']*1e3, f['kernel_ms']*1e3))"
done
done
done
