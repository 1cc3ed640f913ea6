#!/bin/bash
# ON THE GPU BOX: second-order rounding of the frozen stem's weights + conv31 / conv32 as two products (precision 'fp16h')
mkdir -p gpurun_out; O=gpurun_out/r05_second_order.txt; : > $O
python - >> $O 2>&1 <<'PY'
import time, torch, argparse, sys
sys.path.insert(0, ".")
import bench
from videonavqa_amd import _lib as L
L.set_half("f16")
args = argparse.Namespace(precision="fp16h", model="film_attn_pt", batch=8, frames=35, height=224, width=224, blocks=1, channels=512, tail_channels=0, seed=0)
for i in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    model, stem, vgg, od = bench.build(args, torch.device("cuda", 0))
    torch.cuda.synchronize(); print("build (model + stem with calibration) %.2f s, second order %s" % (time.time() - t0, stem.second_order))
PY
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 2400 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "not twelve and not calibration_frames" 2>&1 | tail -4 >> $O
for seed in 0 1 2 3; do
  echo "seed $seed: $(timeout 900 python tools/error_budget.py --precision fp16h --seed $seed 2>/dev/null | tail -1)" >> $O
done
echo "smooth: $(timeout 600 python tools/error_budget.py --precision fp16h --data smooth 2>/dev/null | tail -1)" >> $O
echo "fp16 seed 0: $(timeout 600 python tools/error_budget.py --precision fp16 --seed 0 2>/dev/null | tail -1)" >> $O
echo "fp16 seed 3: $(timeout 600 python tools/error_budget.py --precision fp16 --seed 3 2>/dev/null | tail -1)" >> $O
echo "bf16 seed 0: $(timeout 600 python tools/error_budget.py --precision bf16 --seed 0 2>/dev/null | tail -1)" >> $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "fp16h second order, conv31 / conv32 two products : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "fp16h calibration off (three products, timing)   : $(VNQA_COHERENT_ROUND=0 python bench.py $A 2>/dev/null | q)" >> $O
done
cat $O
