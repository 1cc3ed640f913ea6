#!/bin/bash
# ON THE GPU BOX: --h2d with the runtime's copies done by blit KERNELS instead of the copy engine (HSA_ENABLE_SDMA=0): does the 2.5 % that
# appears when the stem reads copy-engine-written buffers go away?
mkdir -p gpurun_out; O=gpurun_out/r05_h2d_blit.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  regions %s" % (d["value"], d["ms_per_step"], d["repeats"]["clips_per_s"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --precision bf16 --repeats 3 --steps 200"
for r in 1 2; do
  echo "resident                      : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "--h2d (copy engine)           : $(python bench.py $A --h2d 2>/dev/null | q)" >> $O
  echo "--h2d, HSA_ENABLE_SDMA=0      : $(HSA_ENABLE_SDMA=0 python bench.py $A --h2d 2>/dev/null | q)" >> $O
  echo "--h2d u8, HSA_ENABLE_SDMA=0   : $(HSA_ENABLE_SDMA=0 python bench.py $A --h2d --clip-dtype u8 2>/dev/null | q)" >> $O
  echo "resident, HSA_ENABLE_SDMA=0   : $(HSA_ENABLE_SDMA=0 python bench.py $A 2>/dev/null | q)" >> $O
done
cat $O
