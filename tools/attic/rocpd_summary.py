"""Dev tool: per-kernel totals of a rocprofv3 run kept in rocpd format (the default output, <dir>/<name>_results.db).

    python tools/rocpd_summary.py results.db [steps] [rows]
"""
import sqlite3, collections, sys, re
db=sqlite3.connect(sys.argv[1]); steps=int(sys.argv[2]) if len(sys.argv)>2 else 8
c=db.cursor()
rows=list(c.execute("select name, start, end from kernels order by start"))
agg=collections.defaultdict(lambda:[0,0.0])
for n,s,e in rows:
    agg[n][0]+=1; agg[n][1]+=(e-s)
tot=sum(v[1] for v in agg.values())
tt=sum(v[1] for k,v in agg.items() if k.startswith('void at::') or 'rocclr' in k)
print('kernel ms/step', tot/steps/1e6, 'launches/step', len(rows)/steps, 'torch ms/step', tt/steps/1e6)
for n,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[3]) if len(sys.argv)>3 else 30]:
    print(f"{re.sub('at::native::','',n)[:110]:110s} {v[0]/steps:7.1f} {v[1]/steps/1e3:9.1f}us/step")
