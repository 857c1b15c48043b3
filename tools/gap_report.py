"""Dev tool: timeline of the first N kernels of the last complete step in a rocprofv3 kernel trace (gaps = GPU idle, all streams)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_query_windows' in r['Kernel_Name']]
start = idx[-4] if len(idx) >= 4 else idx[0]
t0 = int(rows[start]['Start_Timestamp'])
end_max = 0
for r in rows[start:start + N]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - end_max) / 1e3 if end_max else 0.0
    print('%9.1f us  gap %7.1f  dur %8.1f  st %-3s %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, r.get('Stream_Id', '?'), r['Kernel_Name'][:60]))
    end_max = max(end_max, e)
