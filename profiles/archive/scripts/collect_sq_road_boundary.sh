set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/sq_rb
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --road-boundary --steps 20 --warmup 3 --main-only > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --road-boundary --steps 20 --warmup 3 --main-only > /dev/null 2>&1
cd $ROOT && python3 profiles/sq_summary.py cfg2rb $OUT
rm -rf $OUT
