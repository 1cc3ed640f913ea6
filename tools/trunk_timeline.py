#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace database of `bench.py --no-overlap`, aggregate the kernels of ONE training step's
trunk part (everything after the last stem kernel up to clip_adam) by total time: the dependent chain of the main stream."""
import collections
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
adam = [i for i, r in enumerate(rows) if 'clip_adam' in r[2]]
step = rows[adam[-3] + 1:adam[-2] + 1]
last_stem = max(i for i, r in enumerate(step) if 'Li1ELi2EEEv' in r[2] or 'conv_c64' in r[2])
tr = step[last_stem + 1:]
wall = (tr[-1][1] - tr[0][0]) / 1e6
busy = sum(e - s for s, e, _ in tr) / 1e6
print("step %.3f ms; trunk chain %.3f ms in %d kernels (busy %.3f, gaps %.3f)" % ((step[-1][1] - step[0][0]) / 1e6, wall, len(tr), busy, wall - busy))
agg = collections.OrderedDict()
for s, e, n in tr:
    a = agg.setdefault(n[:100], [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f us  n=%3d  %s" % (us, cnt, k))
if len(sys.argv) > 3:        # per-launch durations (us) of kernels whose name contains argv[3], in launch order
    print([round((e - s) / 1e3, 1) for s, e, n in tr if sys.argv[3] in n])
