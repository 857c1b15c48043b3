"""Dev tool (GPU box): GPU-side timeline of the pipelined step from a rocprofv3 kernel trace -- per queue: busy time, idle gaps between consecutive kernels,
the largest gaps and what ran around them; overall: the fraction of the step during which NO kernel runs.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-extras --no-cpu-baseline --steps 20 --repeats 0
    python tools/trace_gaps.py $OUT/*/t_kernel_trace.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last `steps` steps: find by the dominant kernel's launches
dom = [i for i, r in enumerate(rows) if 'k_conv3_up_split<' in r['Kernel_Name']]
first = dom[-steps] if len(dom) >= steps else 0
last = dom[-1]
rows = rows[first:last]
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
span = (t1 - t0) / 1e6
print('window: %d kernels, %.2f ms, %.3f ms per step' % (len(rows), span, span / (steps - 1)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rs, rs[1:])]
    small = [g for g in gaps if 0 <= g < 50]
    print('queue %s: %d kernels, busy %.2f ms (%.2f ms/step); gaps < 50 us: %d, mean %.1f us, sum %.2f ms/step' % (q, len(rs), busy, busy / (steps - 1), len(small), sum(small) / max(1, len(small)), sum(small) / 1e3 / (steps - 1)))
# union of busy intervals over all queues
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        covered += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
covered += cur_e - cur_s
print('some kernel running: %.1f %% of the window; nothing running: %.3f ms per step' % (100 * covered / (t1 - t0), (t1 - t0 - covered) / 1e6 / (steps - 1)))
# per queue and kernel name: mean duration in this window
for q, rs in byq.items():
    print('--- queue', q)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        a = agg[r['Kernel_Name'][:70]]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for name, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
        print('%7.1f us/step  %5.1f us x %5.1f  %s' % (us / (steps - 1), us / c, c / (steps - 1), name))
