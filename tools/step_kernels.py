#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace database of an overlapped bench.py run: kernel time per training step BY KERNEL NAME, taken over
whole steps of the timed region only (between clip_adam launches; priming, warm-up and the stem-alone passes after the region are
not in it — ADVICE r4: tools/prof_precision.sh divided everything by the step count).

  python tools/step_kernels.py <rocprof dir> [steps back from the end, default 8] [steps to average, default 4]"""
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
adam = [i for i, r in enumerate(rows) if "clip_adam" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lo, hi = adam[-back - n], adam[-back]
span = rows[lo + 1:hi + 1]
agg = {}
for s, e, name, k in span:
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"void |\(anonymous namespace\)::|at::native::", "", name)[:96]
    a = agg.setdefault(name, [0.0, 0])
    a[0] += (e - s) / 1e6
    a[1] += 1
wall = (span[-1][1] - span[0][0]) / 1e6 / n
tot = sum(v[0] for v in agg.values()) / n
print("%d steps: %.3f ms/step wall, %.3f ms/step summed kernel time, %d launches/step" % (n, wall, tot, len(span) // n))
for name, (ms, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print("%8.3f ms/step %6.1f calls/step avg %8.1f us  %s" % (ms / n, calls / n, ms / calls * 1e3, name))
