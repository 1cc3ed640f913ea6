#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_wgrad.txt; : > $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py -k "wgrad" 2>&1 | tail -4 >> $O
timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py -k "wgrad" 2>&1 | tail -4 >> $O
timeout 600 python tools/bench_wgrad.py >> $O 2>&1
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "4-wave ring wgrad: $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "8-wave wgrad     : $(python tools/bench_with.py kernels.WGRAD_EIGHT_WAVES=1 -- $A 2>/dev/null | q)" >> $O
done
cat $O
