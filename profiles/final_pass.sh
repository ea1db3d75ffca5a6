#!/bin/bash
# The last act of a round, on the library that is committed: counters + bench + kernel trace (collect_round.sh), then the parity
# sweeps outside pytest (tests/sweeps/) on the default launch paths and with the cost-ordered stage / rp_cost_kernel forced -- every
# output names the source hash of the library it ran on.  (The RP_AMD_* variables are the DEFAULTS rp_create gives a context's options:
# set for the whole process here.)   usage (GPU box): bash profiles/final_pass.sh r05 [quick]
TAG=${1:-r05}; QUICK=${2:-}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
HASH=$(python3 -c "
import sys; sys.path[:0]=['commonroad-reactive-planner_amd']
from commonroad_rp_amd import _capi; print(_capi.source_hash())")
echo "library source hash $HASH"
bash profiles/collect_round.sh $TAG $QUICK || echo "collect_round failed"
cd $ROOT
{
  echo "Fuzz sweeps (tests/sweeps/fuzz_parity.py) on MI355X, library source hash $HASH"
  echo "default launch paths, seeds 100000 .. 129999:"
  python3 tests/sweeps/fuzz_parity.py 100000 30000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "cost-ordered stage forced (RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_NO_AUTO_MATERIALIZE=1), seeds 130000 .. 139999:"
  RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 130000 10000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "rp_cost_kernel forced (RP_AMD_NO_FUSED_LON=1 RP_AMD_COST_KERNEL=1 RP_AMD_CHUNK_KERNEL=0 RP_AMD_NO_AUTO_MATERIALIZE=1), seeds 140000 .. 144999:"
  RP_AMD_NO_FUSED_LON=1 RP_AMD_COST_KERNEL=1 RP_AMD_CHUNK_KERNEL=0 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 140000 5000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "rp_chunk_kernel forced (RP_AMD_NO_FUSED_LON=1 RP_AMD_CHUNK_KERNEL=1 RP_AMD_NO_AUTO_MATERIALIZE=1), seeds 145000 .. 154999:"
  RP_AMD_NO_FUSED_LON=1 RP_AMD_CHUNK_KERNEL=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 145000 10000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "bounded sweep forced (RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_SWEEP=1 RP_AMD_NO_AUTO_MATERIALIZE=1), seeds 155000 .. 159999:"
  RP_AMD_NO_FUSED_LON=1 RP_AMD_LAZY=1 RP_AMD_SWEEP=1 RP_AMD_NO_AUTO_MATERIALIZE=1 python3 tests/sweeps/fuzz_parity.py 155000 5000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "two-kernel path, 16 lanes, one wavefront per workgroup (RP_AMD_NO_FUSED_LON=1 RP_AMD_G=16 RP_AMD_EVAL_BLOCK=64 RP_AMD_CHUNK_KERNEL=0), seeds 160000 .. 162999:"
  RP_AMD_NO_FUSED_LON=1 RP_AMD_G=16 RP_AMD_EVAL_BLOCK=64 RP_AMD_CHUNK_KERNEL=0 python3 tests/sweeps/fuzz_parity.py 160000 3000 2>&1 | grep -v amdgpu.ids | tail -1
} > $OUT/${TAG}_fuzz_parity.txt 2>&1
echo "fuzz done"
{
  echo "Full benchmark workloads against the oracle's brute force (tests/sweeps/full_scale_parity.py) on MI355X, library source hash $HASH"
  python3 tests/sweeps/full_scale_parity.py 2>&1 | grep -v amdgpu.ids
} > $OUT/${TAG}_full_scale_parity.txt 2>&1
echo "full scale done"
tail -n 3 $OUT/${TAG}_fuzz_parity.txt; tail -n 3 $OUT/${TAG}_full_scale_parity.txt
