#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_round2.txt; : > $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 2400 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "second_order or split_features or goldens" 2>&1 | tail -4 >> $O
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
for seed in 0 3; do
  echo "seed $seed: $(timeout 900 python tools/error_budget.py --precision fp16h --seed $seed 2>/dev/null | tail -1)" >> $O
done
echo "smooth: $(timeout 600 python tools/error_budget.py --precision fp16h --data smooth 2>/dev/null | tail -1)" >> $O
cat $O
bash tools/round5/r05_dbg_pmc.sh 2>&1 | tail -8
