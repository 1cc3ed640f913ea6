#!/bin/bash
mkdir -p gpurun_out
bash tools/h2d_power.sh > /dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_gpu_tests_b.txt 2>&1
tail -5 gpurun_out/r05_gpu_tests_b.txt; cat gpurun_out/r05_h2d_power.txt
