#!/bin/bash
# ON THE GPU BOX: per-kernel split of a whole training step in another precision (PREC=fp16x|fp16w|fp16, default fp16x):
# rocprofv3 --kernel-trace --stats over a short bench run without the side legs.
R=$PWD; export PYTHONPATH=$R; PREC=${PREC:-fp16x}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --precision $PREC --steps 10 --warmup 2 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg --no-fp16-leg > /tmp/pp.out 2> /tmp/pp.err
cd $R
F=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
python - $F <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
it = 15.0 + 10.0 * 0      # 3 priming + 2 warm-up + 10 timed steps (the stem-alone passes after the region add 10 stem passes)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%8.3f ms/step %6.1f calls/step avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / it, int(r["Calls"]) / it, float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("at::native::", "")[:110]))
print("total %.3f ms/step of kernel time" % (tot / 1e6 / it))
PY
