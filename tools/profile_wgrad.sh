#!/bin/bash
# Runs ON THE GPU BOX: the trunk's weight-gradient launch alone (tools/bench_wgrad.py) under rocprofv3 -> kernel-only time
# next to the whole op's (kernel + slab reduce + bias column sums) -> gpurun_out/r03_wgrad.txt
ROOT=$PWD; export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/wg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg -- python3 $ROOT/tools/bench_wgrad.py > /tmp/wg.out 2>/dev/null
cd $ROOT
{
  echo "tools/bench_wgrad.py (280 images x 14x14, 512 -> 512, 3x3, bf16; 259 GFLOP on valid pixels) under rocprofv3 --kernel-trace --stats"
  grep wgrad /tmp/wg.out
  python3 tools/kstats.py $(find /tmp/wg -name '*kernel_stats.csv' | head -1) wgrad slab_reduce colsum | sed 's/(anonymous namespace):://g' | cut -c1-150
  python3 - $(find /tmp/wg -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_wgrad_kernel" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("conv_wgrad_kernel alone: %.1f us = %.0f TFLOP/s on valid pixels" % (us, 2.0 * 280 * 196 * 512 * 512 * 9 / us / 1e6))
PY
} > gpurun_out/r03_wgrad.txt
cat gpurun_out/r03_wgrad.txt
