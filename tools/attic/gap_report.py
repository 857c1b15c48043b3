"""Dev tool: per-stream busy time and launch gaps of the last complete step in a rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_query_windows' in r['Kernel_Name']]
a, b = idx[-4], idx[-2]
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
print('step wall %.1f us, %d kernels' % ((t1 - t0) / 1e3, len(step)))
by = collections.defaultdict(list)
for r in step:
    by[r.get('Stream_Id', '?')].append(r)
for sid, rs in by.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
    gaps = [(int(rs[i + 1]['Start_Timestamp']) - int(rs[i]['End_Timestamp'])) / 1e3 for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print('stream %s: %d kernels, busy %.1f us, sum of gaps %.1f us, median gap %.2f us, max gap %.1f us' % (sid, len(rs), busy, sum(pos), sorted(pos)[len(pos) // 2] if pos else 0, max(pos) if pos else 0))
main = max(by.items(), key=lambda kv: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kv[1]))[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in main:
    k = r['Kernel_Name'].split('(')[0][-60:]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%9.1f us  x%-3d %s' % (t, n, k))
