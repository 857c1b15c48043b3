#!/bin/bash
# HBM traffic passes (separate runs) for one command; prints per-launch FETCH_SIZE / WRITE_SIZE of kernels matching $1, corrected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE under-counts by 2x on gfx950).
#   tools/pmc_traffic.sh <kernel substring> <command...>      (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=$1; shift
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o tcc1 -- "$@" > $OUT/tcc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o tcc2 -- "$@" > $OUT/tcc2.log 2>&1
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
        for k, v in c.items():
            mean = sum(v) / len(v)
            gb = mean * 1024 * (2 if k == 'FETCH_SIZE' else 1) / 1e9
            print(key, k, f'{gb:.2f} GB per launch (corrected)', flush=True)
PY
