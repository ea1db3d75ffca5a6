#!/bin/bash
# SQ counters of the batch's evaluation kernel for one (workload, mode, kernel) -- separate passes, counters only.
# usage (GPU box): bash profiles/collect_kernel_sq.sh <workload> <nocoll|eager|prod> <eval16|lane|chunk|auto> <outdir>
set -e
WL=$1; MODE=$2; KERN=$3; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$4/${WL}_${MODE}_${KERN}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 $ROOT/profiles/run_plans.py $WL $MODE $KERN 6 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $ROOT/profiles/run_plans.py $WL $MODE $KERN 6 > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_INST_LDS --output-format csv -d $OUT/c -- python3 $ROOT/profiles/run_plans.py $WL $MODE $KERN 6 > $OUT/c.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/profiles/run_plans.py $WL $MODE $KERN 20 > $OUT/t.log 2>&1
cd $ROOT && python3 profiles/kernel_sq_summary.py $OUT
