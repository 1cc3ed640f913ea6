#!/bin/bash
# ON THE GPU BOX: kernel-only durations of the weight-gradient op's launches (tools/bench_wgrad.py under rocprofv3 --kernel-trace --stats, csv)
R=$PWD; export PYTHONPATH=$R; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/wg
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/wg -- python3 $R/tools/bench_wgrad.py --headline-only > /tmp/wg.out 2>/dev/null < /dev/null
cat /tmp/wg.out
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/wg/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no trace"); raise SystemExit
agg = {}
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"][:70], r.get("Grid_Size") or r.get("Grid_Size_X") or "?")
    agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print("%-72s grid %9s  n %4d  median %8.1f us  min %8.1f" % (k[0], k[1], len(v), v2[len(v2) // 2] / 1e3, v2[0] / 1e3))
PY
