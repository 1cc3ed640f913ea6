#!/bin/bash
# ON THE GPU BOX: kernel + memory-copy trace of `bench.py --h2d [--clip-dtype u8]` (where do the H2D copies sit relative to the stem
# kernels, and what stalls?).  Prints the copies of the last steps with the stem kernels around them.
R=$PWD; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ph
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ph -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-fp16-leg --h2d $H2D_ARGS --steps 6 --warmup 2 --repeats 1 > /tmp/ph.out 2> /tmp/ph.err
cd $R
tail -c 300 /tmp/ph.out | head -c 200; echo
K=$(find /tmp/ph -name '*kernel_trace.csv' | head -1); M=$(find /tmp/ph -name '*memory_copy_trace.csv' | head -1)
python - $K $M <<'PY'
import csv, sys
ks = list(csv.DictReader(open(sys.argv[1])))
ms = list(csv.DictReader(open(sys.argv[2])))
print("kernel cols", list(ks[0].keys())[:12]); print("copy cols", list(ms[0].keys()))
big = [m for m in ms if int(m["End_Timestamp"]) - int(m["Start_Timestamp"]) > 200_000 and "HOST_TO_DEVICE" in m.get("Direction", "").upper().replace("MEMORY_COPY_", "")]
if not big:
    big = [m for m in ms if int(m["End_Timestamp"]) - int(m["Start_Timestamp"]) > 200_000]
import collections
print("copy directions:", collections.Counter(m.get("Direction") for m in ms))
print("large copies:", len(big))
t0 = min(int(k["Start_Timestamp"]) for k in ks)
def name(k): return k["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]
for m in big[-4:]:
    s, e = int(m["Start_Timestamp"]), int(m["End_Timestamp"])
    print("COPY %s  start %.3f ms  dur %.3f ms  stream %s" % (m.get("Direction", "?"), (s - t0) / 1e6, (e - s) / 1e6, m.get("Stream_Id")))
    around = [k for k in ks if int(k["End_Timestamp"]) > s - 2_000_000 and int(k["Start_Timestamp"]) < e + 2_000_000 and int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) > 150_000]
    for k in around[:14]:
        print("    %-42s s%-3s start %9.3f dur %7.3f" % (name(k), k.get("Stream_Id", "?"), (int(k["Start_Timestamp"]) - t0) / 1e6, (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e6))
PY
