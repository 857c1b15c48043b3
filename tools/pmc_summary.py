"""Summarise the rocprofv3 PMC passes of tools/pmc_conv.sh (gpurun_out/pmc/*_counter_collection.csv) into
profiles/<round>_pmc_bench_C2_B32.csv (per kernel x grid size, per-launch averages) and profiles/<round>_dominant_kernel.json
(what bench.py reports as roofline.traffic).

    python tools/pmc_summary.py [pmc_dir] [round_tag]

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): every pass is its own run,
counter values are already summed over the 8 XCDs, FETCH_SIZE is reported in KB and under-counts by 2x on gfx950 (so fetch
bytes = 2 * FETCH_SIZE * 1024), WRITE_SIZE is KB; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (per-XCD
GRBM_GUI_ACTIVE * 1024 SIMDs) with GRBM_GUI_ACTIVE / 8 = cycles of one XCD.
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
pmc_dir = Path(sys.argv[1]) if len(sys.argv) > 1 else REPO / 'gpurun_out' / 'pmc'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r02'
cfg_name = sys.argv[3] if len(sys.argv) > 3 else 'C2'
DOMINANT = 'k_conv3_up_split_pp'          # rf_conv3d_up_split_presplit_pm 32+64->56 @8^3 (C1-C4); persistent: one workgroup per CU walks the 8192 samples of the B = 32 step


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name


def load(prefix):
    """-> {(kernel, workgroups): {counter: [values per launch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    path = pmc_dir / (prefix + '_counter_collection.csv')
    if not path.exists():
        return out
    per_dispatch = collections.defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(path)):
        d = int(r['Dispatch_Id'])
        per_dispatch[d][r['Counter_Name']] = per_dispatch[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        meta[d] = (short(r['Kernel_Name']), int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
    for d, ctrs in per_dispatch.items():
        for c, v in ctrs.items():
            out[meta[d]][c].append(v)
    return out


sq1, sq2, tcc1, tcc2 = load('sq1'), load('sq2'), load('tcc1'), load('tcc2')


def avg(table, key, ctr):
    v = table.get(key, {}).get(ctr)
    return sum(v) / len(v) if v else float('nan')


rows = []
for key in sq1:
    kern, wgs = key
    if not (kern.startswith('k_') or 'k_conv3' in kern or 'k_conv3_split' in kern or 'k_linear' in kern or 'k_l2' in kern or 'k_convv' in kern):
        continue
    gui = avg(sq1, key, 'GRBM_GUI_ACTIVE')
    cycles = gui / 8.0
    mfma = 100.0 * avg(sq1, key, 'SQ_VALU_MFMA_BUSY_CYCLES') / (cycles * 1024.0) if cycles else float('nan')
    wave_cycles = avg(sq1, key, 'SQ_WAVE_CYCLES')
    wait_any = 100.0 * avg(sq1, key, 'SQ_WAIT_ANY') / wave_cycles if wave_cycles else float('nan')
    wait_inst = 100.0 * avg(sq1, key, 'SQ_WAIT_INST_ANY') / wave_cycles if wave_cycles else float('nan')
    lds_act = avg(sq2, key, 'SQ_LDS_IDX_ACTIVE')
    lds_conf = avg(sq2, key, 'SQ_LDS_BANK_CONFLICT') / lds_act if lds_act and lds_act == lds_act else float('nan')
    fetch_mb = 2.0 * avg(tcc1, key, 'FETCH_SIZE') * 1024.0 / 1e6
    write_mb = avg(tcc2, key, 'WRITE_SIZE') * 1024.0 / 1e6
    rows.append((cycles * len(sq1[key]['GRBM_GUI_ACTIVE']), kern, wgs, len(sq1[key]['GRBM_GUI_ACTIVE']), cycles, mfma, wait_any, wait_inst, lds_conf, fetch_mb, write_mb))
rows.sort(reverse=True)

out_csv = REPO / 'profiles' / ('%s_pmc_bench_%s_B32.csv' % (tag, cfg_name))
with open(out_csv, 'w') as f:
    f.write('# rocprofv3 --pmc <group> --kernel-trace -- python bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-extras  (B=32 chunks/step; tools/pmc_conv.sh, summarised by tools/pmc_summary.py)\n' % cfg_name)
    f.write('# separate passes: {SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE}, {SQ_LDS_* SQ_INSTS_*}, {FETCH_SIZE}, {WRITE_SIZE}\n')
    f.write('# per-launch averages; counters are summed over the 8 XCDs; cycles = GRBM_GUI_ACTIVE/8; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs);\n')
    f.write('# fetch_MB = 2 x FETCH_SIZE KB (gfx950 correction of MI355X_MICROARCH.md, HBM section), write_MB = WRITE_SIZE KB; rows ordered by total cycles\n')
    f.write('kernel,workgroups,launches,cycles,mfma_busy_pct,wait_any_pct,wait_inst_pct,lds_conflict_per_active,fetch_MB,write_MB\n')
    for _, kern, wgs, launches, cycles, mfma, wa, wi, lc, fm, wm in rows[:40]:
        f.write('"%s",%d,%d,%.0f,%.1f,%.1f,%.1f,%.2f,%.0f,%.0f\n' % (kern, wgs, launches, cycles, mfma, wa, wi, lc, fm, wm))
print('wrote', out_csv)

dom = [r for r in rows if r[1] == DOMINANT]
if dom and cfg_name == 'C2':
    _, kern, wgs, launches, cycles, mfma, wa, wi, lc, fm, wm = max(dom, key=lambda r: r[2])
    n_patches = 8192 if kern == 'k_conv3_up_split_pp' else wgs      # (k_conv3_up_split<4>: one patch per workgroup; the persistent form: the bench's B = 32 -> 8192 patches)
    j = {'kernel': kern, 'entry': 'rf_conv3d_up_split_presplit_pm' if kern == 'k_conv3_up_split_pp' else 'rf_conv3d_up_split_k3_gn_relu', 'shape': [32, 64, 8, 56], 'batch': n_patches // 256, 'n_patches': n_patches,
         'fetch_MB_per_launch': fm, 'write_MB_per_launch': wm,
         'traffic_bytes_per_sample': (fm + wm) * 1e6 / n_patches, 'mfma_busy_pct': mfma, 'cycles_per_launch': cycles,
         'source': 'profiles/%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)' % out_csv.name}
    (REPO / 'profiles' / ('%s_dominant_kernel.json' % tag)).write_text(json.dumps(j, indent=1))
    print(json.dumps(j, indent=1))
else:
    print('dominant kernel %s not found in the PMC passes' % DOMINANT)
