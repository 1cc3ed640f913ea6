#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of `bench.py --model mac` (serial and overlapped), top kernels by time and the
# trunk's dependent chain -> gpurun_out/mac_*.txt.   gpurun -- 'bash tools/profile_mac.sh'
ROOT=$PWD; export PYTHONPATH=$ROOT
mkdir -p gpurun_out
ARGS="--model mac --steps 8 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/macS /tmp/macO
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/macS -- python3 $ROOT/bench.py $ARGS --no-overlap > $ROOT/gpurun_out/mac_serial_bench.json 2> $ROOT/gpurun_out/mac_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/macO -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/mac_overlap_bench.json 2> $ROOT/gpurun_out/mac_overlap.err
cd $ROOT
for m in S O; do
  f=$(find /tmp/mac$m -name '*kernel_stats.csv' | head -1)
  python - "$f" > gpurun_out/mac_kernels_$m.txt <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
steps = 11.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("total GPU time %.3f ms/step over %d kernel names, %.0f launches/step" % (tot, len(rows), sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:45]:
    print("%8.3f ms/step  %7.1f calls/step  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps,
                                                                  float(r["AverageNs"]) / 1e3, r["Name"][:120]))
EOF
done
tail -1 gpurun_out/mac_serial_bench.json | cut -c1-300
tail -1 gpurun_out/mac_overlap_bench.json | cut -c1-300
head -30 gpurun_out/mac_kernels_S.txt
