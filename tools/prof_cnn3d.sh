#!/bin/bash
# ON THE GPU BOX: kernel-trace summary of config 2 (tools/bench_cnn3d.py) -> gpurun_out/cnn3d_kstats.txt
R=$PWD; mkdir -p gpurun_out
python tools/bench_cnn3d.py ${1:-bf16} 2>/dev/null | tee gpurun_out/cnn3d_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/tools/bench_cnn3d.py ${1:-bf16} > /tmp/p3.out 2> /tmp/p3.err
cd $R
F=$(find /tmp/p3 -name '*kernel_stats.csv' | head -1)
if [ -z "$F" ]; then tail -5 /tmp/p3.err; exit 1; fi
python tools/kstats.py $F | head -${2:-45} | tee gpurun_out/cnn3d_kstats.txt
