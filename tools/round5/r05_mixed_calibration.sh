#!/bin/bash
# ON THE GPU BOX: default calibration frames = half uniform noise, half smooth (stem.default_calibration_frames); logits error of
# precision 'fp16h' per weight seed and on smooth / blocks clips (blocks: a kind the calibration frames contain nothing of)
mkdir -p gpurun_out; O=gpurun_out/r05_mixed_calibration.txt; : > $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 2400 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "not twelve and not calibration_frames" 2>&1 | tail -3 >> $O
for seed in 0 1 2 3; do
  echo "seed $seed noise : $(timeout 900 python tools/error_budget.py --precision fp16h --seed $seed 2>/dev/null | tail -1)" >> $O
done
echo "seed 0 smooth: $(timeout 600 python tools/error_budget.py --precision fp16h --data smooth 2>/dev/null | tail -1)" >> $O
echo "seed 0 blocks: $(timeout 600 python tools/error_budget.py --precision fp16h --data blocks 2>/dev/null | tail -1)" >> $O
echo "seed 3 blocks: $(timeout 600 python tools/error_budget.py --precision fp16h --data blocks --seed 3 2>/dev/null | tail -1)" >> $O
echo "fp16 seed 0 blocks: $(timeout 600 python tools/error_budget.py --precision fp16 --data blocks 2>/dev/null | tail -1)" >> $O
cat $O
