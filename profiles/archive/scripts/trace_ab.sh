#!/bin/bash
# rocprofv3 kernel trace of one workload's timed region for several library builds -> per-kernel duration summaries, one box.
# usage (GPU box): bash profiles/trace_ab.sh cfg3 fused librp_amd_r02.so librp_amd.so
WL=$1; MODE=$2; shift 2
for lib in "$@"; do echo "== $lib $WL $MODE"; bash $GRAFT_REPO_ROOT/profiles/trace_lib.sh $lib $WL $MODE | cut -c1-150; done
