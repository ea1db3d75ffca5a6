#!/bin/bash
# rp_lon_kernel on one workload: rocprofv3 duration, per-workgroup timeline (-DRP_TIMELINE -DRP_TIMELINE_LON), in-kernel stamps.
WL=${1:-cfg3}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r04/lon_${WL}; LIBDIR=$ROOT/commonroad-reactive-planner_amd/lib
mkdir -p $OUT; cd $ROOT
bash profiles/trace_lib.sh librp_amd.so $WL draw > $OUT/trace.txt 2>&1
[ -f $LIBDIR/librp_amd_tll.so ] && RP_AMD_LIBRARY=$LIBDIR/librp_amd_tll.so RP_AMD_PRINT_STAMPS=1 python3 profiles/probe_stamps_cfg3.py $WL > $OUT/wg_timeline.txt 2>&1
[ -f $LIBDIR/librp_amd_st.so ] && RP_AMD_LIBRARY=$LIBDIR/librp_amd_st.so RP_AMD_PRINT_STAMPS=1 python3 profiles/probe_stamps_cfg3.py $WL > $OUT/stamps.txt 2>&1
head -6 $OUT/trace.txt; tail -3 $OUT/wg_timeline.txt; grep "profile kernel" $OUT/stamps.txt | tail -2
