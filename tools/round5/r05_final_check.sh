#!/bin/bash
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_gpu_tests_c.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_gpu_tests_c.txt | tail -3
bash tools/refresh_profiles.sh 5 2>&1 | tail -3 | cut -c1-300
PRECS="fp16h" bash tools/round5/r05_pooling_heads.sh > gpurun_out/r05_pooling_heads_c.txt 2>&1; cat gpurun_out/r05_pooling_heads_c.txt
