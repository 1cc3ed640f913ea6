#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_h2d_ablation2.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  regions %s" % (d["value"], d["ms_per_step"], d["repeats"]["clips_per_s"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --precision bf16 --repeats 3 --steps 200"
for r in 1 2; do
  echo "resident            : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "--h2d               : $(python bench.py $A --h2d 2>/dev/null | q)" >> $O
  echo "--h2d stem_no_wait  : $(python bench.py $A --h2d --h2d-ablation stem_no_wait 2>/dev/null | q)" >> $O
  echo "--h2d copy_no_wait  : $(python bench.py $A --h2d --h2d-ablation copy_no_wait 2>/dev/null | q)" >> $O
  echo "--h2d no_waits      : $(python bench.py $A --h2d --h2d-ablation no_waits 2>/dev/null | q)" >> $O
done
cat $O
timeout 1500 python tools/experiments/gptq_stem_weights.py --seeds 0 1 --batches 6 > gpurun_out/r05_gptq_noise_calib.txt 2>&1
tail -25 gpurun_out/r05_gptq_noise_calib.txt
