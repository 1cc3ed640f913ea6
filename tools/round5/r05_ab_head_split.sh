#!/bin/bash
# ON THE GPU BOX: conv_init's output kept as hi + lo into its BatchNorm (ops.HEAD_SPLIT_OUT) — kernel test, logits error per weight seed,
# throughput A/B on one box.  (module attribute switched by tools/error_budget.py's module.ATTR=int words and tools/bench_with.py)
mkdir -p gpurun_out; O=gpurun_out/r05_head_split.txt; : > $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "split_out or bnstats or goldens" 2>&1 | tail -3 >> $O
for seed in 0 1 2 3; do
  echo "seed $seed" >> $O; timeout 900 python tools/error_budget.py --precision fp16h --seed $seed ops.HEAD_SPLIT_OUT=1 ops.HEAD_SPLIT_OUT=0 2>/dev/null | tail -2 >> $O
done
echo "smooth" >> $O; timeout 600 python tools/error_budget.py --precision fp16h --data smooth ops.HEAD_SPLIT_OUT=1 ops.HEAD_SPLIT_OUT=0 2>/dev/null | tail -2 >> $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "head split on : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "head split off: $(python tools/bench_with.py ops.HEAD_SPLIT_OUT=0 -- $A 2>/dev/null | q)" >> $O
done
cat $O
