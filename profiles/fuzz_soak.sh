#!/bin/bash
# A longer run of the fuzz comparison (tests/sweeps/fuzz_parity.py: random small plans, HIP path against the oracle, every label / reason /
# cost / winner / counter) on seeds the round's final pass did not use.   usage (GPU box): bash profiles/fuzz_soak.sh > gpurun_out/<dir>/soak.txt
cd $GRAFT_REPO_ROOT
HASH=$(python3 -c "
import sys; sys.path[:0]=['commonroad-reactive-planner_amd']
from commonroad_rp_amd import _capi; print(_capi.source_hash())")
echo "Fuzz soak on MI355X, library source hash $HASH"
echo "default launch paths, seeds 300000 .. 399999 (four runs of 25 000: a line every two minutes):"
for s0 in 300000 325000 350000 375000; do python3 tests/sweeps/fuzz_parity.py $s0 25000 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "cost-ordered stage forced, seeds 400000 .. 429999:"
RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 400000 30000 2>&1 | grep -v amdgpu.ids | tail -1
echo "rp_chunk_kernel forced, seeds 430000 .. 459999:"
RP_AMD_NO_FUSED_LON=1 RP_AMD_CHUNK_KERNEL=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 430000 30000 2>&1 | grep -v amdgpu.ids | tail -1
echo "bounded sweep forced, seeds 460000 .. 479999:"
RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_SWEEP=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 460000 20000 2>&1 | grep -v amdgpu.ids | tail -1
echo "rp_cost_kernel forced, seeds 480000 .. 499999:"
RP_AMD_NO_FUSED_LON=1 RP_AMD_COST_KERNEL=1 RP_AMD_CHUNK_KERNEL=0 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 480000 20000 2>&1 | grep -v amdgpu.ids | tail -1
