#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): regenerates everything under profiles/ for round $1
# (default 1) from the current build and leaves copies in gpurun_out/ for the merge back.  ~6 minutes.
#   bench.py default / --no-overlap / layer-by-layer stem, rocprofv3 kernel trace of the bench, two PMC passes.
R=${1:-2}; TAG=$(printf "r%02d" $R)
ROOT=$PWD; export PYTHONPATH=$ROOT
mkdir -p gpurun_out profiles
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err
python bench.py --no-overlap --no-cpu-baseline --no-parity > gpurun_out/${TAG}_bench_nooverlap.json 2>> gpurun_out/bench.err
VNQA_STEM_COMPOSE=0 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${TAG}_bench_layer_by_layer_stem.json 2>> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmcF /tmp/pmcW
rocprofv3 --kernel-trace --stats -d /tmp/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.json 2> $ROOT/gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-overlap > /dev/null 2>&1
cd $ROOT
for f in bench bench_nooverlap bench_layer_by_layer_stem; do tail -1 gpurun_out/${TAG}_$f.json > profiles/${TAG}_$f.json; done
python tools/profile_summary.py /tmp/prof --steps 28 --round $R --bench profiles/${TAG}_bench.json \
    --bench-nooverlap profiles/${TAG}_bench_nooverlap.json --profiled gpurun_out/prof_bench.json
python tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW > profiles/${TAG}_pmc_traffic.json
python tools/bench_hbm_kernels.py > gpurun_out/hbm_kernels.txt 2>&1
cp profiles/${TAG}_* gpurun_out/
tail -1 profiles/${TAG}_bench.json
