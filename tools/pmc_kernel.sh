#!/bin/bash
# SQ counter passes (separate runs) for one command; prints per-kernel per-launch averages of kernels matching $1.
#   tools/pmc_kernel.sh <kernel substring> <command...>      (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=$1; shift
OUT=$R/gpurun_out/pmc_kernel
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o sq1 -- "$@" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT -o sq2 -- "$@" > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $OUT -o sq3 -- "$@" > $OUT/sq3.log 2>&1
python - "$OUT" "$PAT" <<'PY'
import csv, sys, collections, glob
out, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + '/*_counter_collection.csv')):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        d = int(r['Dispatch_Id']); per[d][r['Counter_Name']] += float(r['Counter_Value']); names[d] = (r['Kernel_Name'][:60], int(r['Grid_Size']) // int(r['Workgroup_Size']))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, c in per.items():
        for k, v in c.items(): agg[names[d]][k].append(v)
    for key, c in agg.items():
        print(key, {k: round(sum(v) / len(v)) for k, v in c.items()}, flush=True)
PY
