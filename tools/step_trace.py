"""Dev tool (GPU box): per-queue, per-(kernel, grid) mean durations of the pipelined step from a rocprofv3 kernel trace, and the back-end queue's kernels in
launch order for one steady-state step.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-extras --no-cpu-baseline --steps 20 --repeats 0
    python tools/step_trace.py $OUT/*/t_kernel_trace.csv [steps]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dom = [i for i, r in enumerate(rows) if 'k_conv3_up_split_pp' in r['Kernel_Name'] or 'k_conv3_up_split<' in r['Kernel_Name']]
first, last = (dom[-steps] if len(dom) >= steps else dom[0]), dom[-1]
win = rows[first:last]
t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['Start_Timestamp'])
nst = steps - 1
print('window: %d kernels, %.3f ms per step' % (len(win), (t1 - t0) / 1e6 / nst))
def grid(r):
    return 'x'.join(str(int(r['Grid_Size_' + d]) // max(1, int(r['Workgroup_Size_' + d]))) for d in 'XYZ')
byq = collections.defaultdict(list)
for r in win:
    byq[r['Queue_Id']].append(r)
main_q = max(byq, key=lambda q: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in byq[q]))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6 / nst
    print('--- queue %s%s: %.1f kernels/step, busy %.3f ms/step' % (q, ' (back end)' if q == main_q else '', len(rs) / nst, busy))
    agg = collections.OrderedDict()
    for r in rs:
        a = agg.setdefault((r['Kernel_Name'][:64], grid(r)), [0, 0.0])
        a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for (name, g), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%8.1f us/step  %7.1f us x %4.1f  wg %-12s %s' % (us / nst, us / c, c / nst, g, name))
# one step of the back-end queue in launch order (the second to last dominant launch to the last)
rs = byq[main_q]
di = [i for i, r in enumerate(rs) if 'k_conv3_up_split_pp' in r['Kernel_Name'] or 'k_conv3_up_split<' in r['Kernel_Name']]
if len(di) >= 2:
    print('--- back-end queue, one step in launch order (start offset us, duration us, gap before us)')
    seg = rs[di[-2]:di[-1]]
    base, prev_end = int(seg[0]['Start_Timestamp']), None
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f %8.1f %7.1f  wg %-12s %s' % ((s - base) / 1e3, (e - s) / 1e3, ((s - prev_end) / 1e3 if prev_end else 0.0), grid(r), r['Kernel_Name'][:70]))
        prev_end = e
