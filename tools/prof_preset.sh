#!/bin/bash
# ON THE GPU BOX: the eval.sh preset (5 blocks x 1024 channels, bs 32) with the stem NOT overlapped — the trunk chain's kernels of one
# step with the chip to themselves, aggregated by total time, plus the per-launch durations of the conv kernels in launch order.
#   gpurun -- 'bash tools/prof_preset.sh [extra bench.py args]'
ROOT=$PWD; export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace -d /tmp/pp -- python3 $ROOT/bench.py --blocks 5 --channels 1024 --batch 32 --steps 4 --warmup 2 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg --no-overlap "$@" > /tmp/pp_bench.json 2> /tmp/pp.err
tail -1 /tmp/pp_bench.json | cut -c100-200
python3 $ROOT/tools/trunk_timeline.py /tmp/pp 34 conv_
