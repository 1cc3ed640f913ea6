#!/usr/bin/env python3
"""From a rocprofv3 --hip-trace --memory-copy-trace --kernel-trace database of `bench.py --h2d`: for every large host-to-device copy,
WHEN the launch thread called the copy API and when the copy engine started / finished it, next to the clip_adam kernels (step marks):
is a late clip the launch thread's doing or the copy queue's?"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t):
    return [r[1] for r in c.execute("pragma table_info(%s)" % t)]
mc = [t for t in tabs if 'memory_cop' in t]
print("tables:", [t for t in tabs if any(k in t for k in ('memory_cop', 'region', 'api', 'kernel_dispatch'))][:12])
t = [x for x in mc if 'rocpd' in x][0]
cp = c.execute("select start, end, %s from %s order by start" % ("size" if "size" in cols(t) else "0", t)).fetchall()
big = [(s, e, b) for s, e, b in cp if (b or 0) > 50e6]
kd = [x for x in tabs if 'kernel_dispatch' in x][0]
ks = [x for x in tabs if 'kernel_symbol' in x][0]
adam = [r[0] for r in c.execute("select d.start from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%clip_adam%%' order by d.start" % (kd, ks))]
reg = [x for x in tabs if x.startswith('rocpd_region') and 'name' not in x]
api = []
for rt in reg[:1]:
    rc = cols(rt)
    print(rt, rc)
    st = [x for x in tabs if x.startswith('rocpd_string')][0]
    try:
        api = c.execute("select r.start, r.end, s.string from %s r join %s s on r.name_id = s.id where s.string like 'hipMemcpy%%' order by r.start" % (rt, st)).fetchall()
    except Exception as e:
        print("api query failed:", e)
t0 = adam[0] if adam else big[0][0]
print("clip_adam starts (ms):", [round((a - t0) / 1e6, 2) for a in adam])
print("large H2D copies: start / end (ms), and the API call that started at most 50 ms before it:")
for s, e, b in big:
    calls = [(a, z, n) for a, z, n in api if a <= s and s - a < 50e6 and (z - a) >= 0]
    near = calls[-1] if calls else None
    print("  copy %8.2f .. %8.2f  (%d MB)   api %s" % ((s - t0) / 1e6, (e - t0) / 1e6, b / 1e6,
          ("%s called %.2f, returned %.2f" % (near[2], (near[0] - t0) / 1e6, (near[1] - t0) / 1e6)) if near else "-"))
# the launch thread's long HIP calls (> 0.2 ms): where it blocks
if reg:
    rt = reg[0]
    st = [x for x in tabs if x.startswith('rocpd_string')][0]
    long_calls = c.execute("select r.start, r.end, s.string, r.tid from %s r join %s s on r.name_id = s.id where r.end - r.start > 200000 order by r.start" % (rt, st)).fetchall()
    print("HIP calls longer than 0.2 ms inside the timed steps (start ms, duration ms, name, thread):")
    lo, hi = (adam[6], adam[-1]) if len(adam) > 8 else (t0, t0 + 10**12)
    for a, z, n, tid in long_calls:
        if lo <= a <= hi:
            print("  %8.2f  %6.2f  %s  tid %s" % ((a - t0) / 1e6, (z - a) / 1e6, n, tid))
# one timed step of the launch thread (between two consecutive clip uploads): HIP calls by total time
if reg and len(api) > 12:
    ups = [a for a, z, n in api if n == 'hipMemcpyAsync' and (z - a) > 20000][-4:-2]
    if len(ups) == 2:
        rows = c.execute("select r.start, r.end, s.string from %s r join %s s on r.name_id = s.id where r.start >= %d and r.start < %d order by r.start" % (rt, st, ups[0], ups[1])).fetchall()
        import collections
        agg = collections.OrderedDict()
        for a, z, n in rows:
            e = agg.setdefault(n, [0, 0.0, 0.0])
            e[0] += 1; e[1] += (z - a) / 1e3; e[2] = max(e[2], (z - a) / 1e3)
        print("launch thread between two uploads: %.2f ms, %d HIP calls, %.2f ms inside them" % ((ups[1] - ups[0]) / 1e6, len(rows), sum(v[1] for v in agg.values()) / 1e3))
        for n, (cnt, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print("   %5d calls %9.1f us total  max %7.1f us  %s" % (cnt, tot, mx, n))
        # the largest gaps between consecutive API calls (time spent in Python / elsewhere)
        gaps = sorted(((rows[i + 1][0] - rows[i][1]) / 1e3, rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1))[-6:]
        print("   largest gaps between consecutive HIP calls (us, after, before):", [(round(g), a, b) for g, a, b in gaps])
