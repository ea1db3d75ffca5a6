#!/bin/bash
# A/B the library variants on cfg2 and cfg5 (draw mode): prints kernel_ms and ms_per_step
cd $GRAFT_REPO_ROOT
for lib in librp_amd.so librp_amd_w3.so librp_amd_w2.so; do
  for wl in cfg2 cfg5; do
    steps=200; [ $wl = cfg5 ] && steps=10
    RP_AMD_LIBRARY=$GRAFT_REPO_ROOT/commonroad-reactive-planner_amd/lib/$lib python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$lib','$wl','kernel_ms=%.4f'%r['roofline']['kernel_ms'],'frac=%.3f'%r['roofline']['frac'],'ms_per_step=%.4f'%r['ms_per_step'],'fused_kernel_ms=%.4f'%r['fused_mode']['kernel_ms'])"
  done
done
