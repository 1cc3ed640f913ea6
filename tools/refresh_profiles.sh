#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): regenerates everything under profiles/ for round $1
# (default 1) from the current build and leaves copies in gpurun_out/ for the merge back.  ~6 minutes.
#   bench.py default / --no-overlap / layer-by-layer stem, rocprofv3 kernel trace of the bench, two PMC passes.
R=${1:-5}; TAG=$(printf "r%02d" $R)
ROOT=$PWD; export PYTHONPATH=$ROOT
mkdir -p gpurun_out profiles
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err
python bench.py --no-overlap --no-cpu-baseline --no-parity > gpurun_out/${TAG}_bench_nooverlap.json 2>> gpurun_out/bench.err
VNQA_STEM_COMPOSE=0 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${TAG}_bench_layer_by_layer_stem.json 2>> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmcF /tmp/pmcW
rocprofv3 --kernel-trace --stats -d /tmp/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg > $ROOT/gpurun_out/prof_bench.json 2> $ROOT/gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-overlap > /dev/null 2>&1
rm -rf /tmp/pmcM /tmp/pmcS
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/pmcM -- python3 $ROOT/tools/stem_only.py --precision fp16h --iters 5 > /dev/null 2>$ROOT/gpurun_out/pmcM.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcS -- python3 $ROOT/tools/stem_only.py --precision fp16h --iters 5 > /dev/null 2>$ROOT/gpurun_out/pmcS.err
cd $ROOT
for f in bench bench_nooverlap bench_layer_by_layer_stem; do tail -1 gpurun_out/${TAG}_$f.json > profiles/${TAG}_$f.json; done
python tools/profile_summary.py /tmp/prof --steps 28 --round $R --bench profiles/${TAG}_bench.json \
    --bench-nooverlap profiles/${TAG}_bench_nooverlap.json --profiled gpurun_out/prof_bench.json
python tools/pmc_traffic.py /tmp/pmcF /tmp/pmcW > profiles/${TAG}_pmc_traffic.json
python tools/pmc_mfma.py /tmp/pmcM /tmp/pmcS > profiles/${TAG}_pmc_mfma.json 2> gpurun_out/pmc_mfma.err
(cd /tmp && rm -rf /tmp/profT && rocprofv3 --kernel-trace -d /tmp/profT -- python3 $ROOT/bench.py --steps 6 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg --no-overlap > /dev/null 2>&1)
python tools/trunk_timeline.py /tmp/profT > profiles/${TAG}_trunk_timeline.txt 2> gpurun_out/trunk_timeline.err
python tools/bench_hbm_kernels.py > gpurun_out/hbm_kernels.txt 2>&1
cp profiles/${TAG}_* gpurun_out/
tail -1 profiles/${TAG}_bench.json
