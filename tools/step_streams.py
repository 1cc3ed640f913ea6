#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace database of an OVERLAPPED bench.py run: every kernel of ONE training step (between two clip_adam
launches) in start order with its queue, start offset and duration — where the trunk's chain waits while the stem co-runs."""
import glob
import re
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute("select d.start, d.end, s.kernel_name, d.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (q, kd, ks)).fetchall()
mc = [t for t in tabs if 'memory_copy' in t and 'rocpd' in t]
for t in mc[:1]:          # (--memory-copy-trace: copies as rows of a pseudo queue "copy")
    mcols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    if "start" in mcols and "end" in mcols:
        size = "size" if "size" in mcols else "0"
        rows += [(s_, e_, "MEMCPY %d bytes" % (b or 0), "copy") for s_, e_, b in c.execute("select start, end, %s from %s" % (size, t)).fetchall()]
rows.sort(key=lambda r: r[0])
import os
mark = os.environ.get("MARK", "clip_adam")          # the kernel that ends a step (inference has no clip_adam: MARK=ce_loss)
adam = [i for i, r in enumerate(rows) if mark in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # consecutive steps to list
step = rows[adam[-back - nsteps + 1] + 1:adam[-back + 1] + 1]
t0 = step[0][0]
qs = sorted(set(r[3] for r in step), key=str)
print("step %.3f ms, %d kernels, queues %s" % ((step[-1][1] - t0) / 1e6, len(step), qs))
busy = dict((k, 0.0) for k in qs)
for s, e, n, k in step:
    busy[k] += (e - s) / 1e3
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)
    name = re.sub(r"^_ZN2at6native\d*", "at::", name)[:70]
    print("q%-3s %9.1f +%8.1f us  %s" % (qs.index(k), (s - t0) / 1e3, (e - s) / 1e3, name))
print("busy us per queue:", dict((qs.index(k), round(v, 1)) for k, v in busy.items()))
