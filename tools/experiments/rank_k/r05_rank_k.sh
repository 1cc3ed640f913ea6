#!/bin/bash
# ON THE GPU BOX: conv_init as two products on [hi | lo] features against per-step rank-32 rounded weights (vnqa_rank_k_round)
mkdir -p gpurun_out; O=gpurun_out/r05_rank_k.txt; : > $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 2400 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "not twelve and not calibration_frames" 2>&1 | tail -4 >> $O
python - >> $O 2>&1 <<'PY'
import time, torch, argparse, sys
sys.path.insert(0, ".")
import bench
from videonavqa_amd import _lib as L, kernels as K
L.set_half("f16")
args = argparse.Namespace(precision="fp16h", model="film_attn_pt", batch=8, frames=35, height=224, width=224, blocks=1, channels=512, tail_channels=0, seed=0)
torch.cuda.synchronize(); t0 = time.time()
model, stem, vgg, od = bench.build(args, torch.device("cuda", 0))
torch.cuda.synchronize(); print("build %.2f s, moments %s" % (time.time() - t0, stem.feature_moments is not None))
wt = K.pack_conv_weight(model.conv_init.weight, torch.float32)
for _ in range(3): K.rank_k_round(wt, stem.feature_moments, 2)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.rank_k_round(wt, stem.feature_moments, 2)
e1.record(); torch.cuda.synchronize(); print("rank_k_round 512 x 4608: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
PY
for seed in 0 1 2 3; do
  echo "seed $seed: $(timeout 900 python tools/error_budget.py --precision fp16h --seed $seed 2>/dev/null | tail -1)" >> $O
done
echo "smooth: $(timeout 600 python tools/error_budget.py --precision fp16h --data smooth 2>/dev/null | tail -1)" >> $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "fp16h: $(python bench.py $A 2>/dev/null | q)   bf16: $(python bench.py $A --precision bf16 2>/dev/null | q)" >> $O
done
echo "eval: $(python bench.py $A --mode eval 2>/dev/null | q)" >> $O
cat $O
