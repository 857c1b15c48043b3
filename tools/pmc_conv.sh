#!/bin/bash
# PMC passes for the conv kernels (run on the GPU box through gpurun). Separate passes per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
CFG=${1:-C2}
OUT=$R/gpurun_out/pmc_$CFG
mkdir -p $OUT
CMD="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
[ "$CFG" = C5 ] && CMD="$CMD --batch 16"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o tcc1 -- $CMD > $OUT/tcc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o tcc2 -- $CMD > $OUT/tcc2.log 2>&1
ls -la $OUT | head -30
tail -2 $OUT/sq1.log | cut -c1-200
