#!/bin/bash
# ON THE GPU BOX: BASELINE config 2 (tools/bench_cnn3d.py: v_only_cnn3d, bs = 32, 16x3x112x112, fwd + bwd + Adam) un-profiled and
# under rocprofv3 --kernel-trace --stats -> gpurun_out/rNN_cnn3d.md (copy to profiles/).   gpurun -- 'bash tools/prof_cnn3d.sh 3'
R=${1:-3}; TAG=$(printf "r%02d" $R)
ROOT=$PWD; export PYTHONPATH=$ROOT; mkdir -p gpurun_out
python tools/bench_cnn3d.py bf16 2>/dev/null | tail -1 > gpurun_out/cnn3d_bench.json
VNQA_CNN3D_GENERIC=1 python tools/bench_cnn3d.py bf16 2>/dev/null | tail -1 > gpurun_out/cnn3d_bench_generic.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $ROOT/tools/bench_cnn3d.py bf16 > /tmp/p3.out 2> /tmp/p3.err
cd $ROOT
F=$(find /tmp/p3 -name '*kernel_stats.csv' | head -1)
if [ -z "$F" ]; then tail -5 /tmp/p3.err; exit 1; fi
python tools/cnn3d_profile_md.py $F gpurun_out/cnn3d_bench.json gpurun_out/cnn3d_bench_generic.json $R > gpurun_out/${TAG}_cnn3d.md
head -40 gpurun_out/${TAG}_cnn3d.md
