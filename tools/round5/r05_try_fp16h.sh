#!/bin/bash
# first GPU contact of precision 'fp16h': kernel tests, error over seeds, throughput
mkdir -p gpurun_out
export VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16
timeout 1500 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "not twelve and not calibration_frames" > gpurun_out/r05_pair_tests.txt 2>&1
tail -15 gpurun_out/r05_pair_tests.txt
unset VNQA_TEST_LOW_PRECISION VNQA_HALF
for seed in 0 1 2 3; do
  timeout 600 python tools/error_budget.py --precision fp16h --seed $seed "COH=1" >> gpurun_out/r05_fp16h_err.txt 2>> gpurun_out/r05_fp16h_err.err
done
timeout 300 python tools/error_budget.py --precision fp16h --data smooth "COH=1" >> gpurun_out/r05_fp16h_err.txt 2>> gpurun_out/r05_fp16h_err.err
cat gpurun_out/r05_fp16h_err.txt; tail -5 gpurun_out/r05_fp16h_err.err
for prec in fp16h fp16 fp16h fp16; do
  timeout 600 python bench.py --precision $prec --no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 1 2>> gpurun_out/r05_bench_try.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'].get('precision', ''), d['value'], d['ms_per_step'], d['config'].get('stem_alone_ms'))" >> gpurun_out/r05_bench_try.txt
done
cat gpurun_out/r05_bench_try.txt; tail -5 gpurun_out/r05_bench_try.err
