#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the default bench + the per-kernel table (profiles/rNN_kernel_stats.*)
mkdir -p gpurun_out
ROOT=$PWD; export PYTHONPATH=$ROOT; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg > $ROOT/gpurun_out/prof_bench.json 2> $ROOT/gpurun_out/prof_bench.err
cd $ROOT
python bench.py --no-cpu-baseline --no-parity --no-eval-leg > gpurun_out/r04_bench_quick.json 2>/dev/null
python bench.py --no-overlap --no-cpu-baseline --no-parity --no-eval-leg > gpurun_out/r04_bench_nooverlap.json 2>/dev/null
python tools/profile_summary.py /tmp/prof --steps 28 --round 4 --bench gpurun_out/r04_bench_quick.json --bench-nooverlap gpurun_out/r04_bench_nooverlap.json --profiled gpurun_out/prof_bench.json
cp profiles/r04_kernel_stats.* gpurun_out/
head -32 profiles/r04_kernel_stats.md | tail -24 | cut -c1-170
