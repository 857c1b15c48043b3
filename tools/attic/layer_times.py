"""Dev tool: per-layer conv timings of the retrieval path from a rocprofv3 kernel trace of bench.py (main stream, last step)."""
import csv, sys, collections
path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_query_windows' in r['Kernel_Name']]
step = rows[idx[-4]:idx[-2]]            # one complete step (two k_query_windows launches per step)
n = 256 * B
layers = [('1->8@16', 1, 8, 16, n), ('8->16@16', 8, 16, 16, n), ('16->16@8', 16, 16, 8, n), ('16->32@8', 16, 32, 8, n), ('32->32@4', 32, 32, 4, n),
          ('32->64@4', 32, 64, 4, n), ('64->64@2', 64, 64, 2, n), ('64->128@2', 64, 128, 2, n), ('192->64@4', 192, 64, 4, n), ('64->64@4', 64, 64, 4, n),
          ('96->56@8', 96, 56, 8, n), ('56->16@8', 56, 16, 8, n), ('dec16->16@64', 16, 16, 64, B), ('dec16->16@64', 16, 16, 64, B)]
streams = collections.Counter(r['Stream_Id'] for r in step if 'conv3' in r['Kernel_Name'])
for sid in streams:
    cs = [r for r in step if 'conv3' in r['Kernel_Name'] and r['Stream_Id'] == sid]
    if len(cs) != len(layers):
        continue
    tot = 0
    for (name, ci, co, e, nn), r in zip(layers, cs):
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += dur
        print('%-14s %8.1f us %7.1f TF/s   %s' % (name, dur, 2 * 27 * ci * co * e ** 3 * nn / dur / 1e6, r['Kernel_Name'].split('(')[0][-44:]))
    print('sum %.1f us' % tot)
