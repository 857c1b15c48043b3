#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs) for one command; prints per-launch averages (MB, fetch with the gfx950 x2 correction) of kernels matching $1.
#   tools/pmc_traffic_cmd.sh <kernel substring> <command...>      (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=$1; shift
OUT=$R/gpurun_out/pmc_traffic_cmd
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o f -- "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o w -- "$@" > $OUT/w.log 2>&1
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
        print(key, {k: '%.0f MB' % ((2 if k == 'FETCH_SIZE' else 1) * sum(v) / len(v) * 1024 / 1e6) for k, v in c.items()}, flush=True)
PY
