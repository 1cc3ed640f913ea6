#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_ring_edges.txt; : > $O
timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py -k "ring" 2>&1 | tail -3 >> $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py tests/test_gpu_models.py -k "ring or stem or golden" 2>&1 | tail -3 >> $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "fp16h: $(python bench.py $A 2>/dev/null | q)   bf16: $(python bench.py $A --precision bf16 2>/dev/null | q)" >> $O
done
echo "stem_only: $(python tools/stem_only.py 2>/dev/null | tail -1)" >> $O
cat $O
