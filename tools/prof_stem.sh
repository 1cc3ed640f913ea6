#!/bin/bash
# ON THE GPU BOX: per-kernel split of the frozen stem alone (tools/stem_only.py under rocprofv3 --kernel-trace --stats)
R=$PWD; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $R/tools/stem_only.py --iters 20 $STEM_ARGS > /tmp/ps.out 2> /tmp/ps.err
cd $R
F=$(find /tmp/ps -name '*kernel_stats.csv' | head -1)
tail -1 /tmp/ps.out
python - $F <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
it = 23.0
for r in rows[:16]:
    print("%8.3f ms/pass %5.1f calls/pass avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / it, int(r["Calls"]) / it, float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "")[:100]))
print("total %.3f ms/pass" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / it))
PY
