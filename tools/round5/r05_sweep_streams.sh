#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_sweep_streams.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2; do
  echo "default                    : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "VNQA_STEM_RESERVE_CUS=8    : $(VNQA_STEM_RESERVE_CUS=8 python bench.py $A 2>/dev/null | q)" >> $O
  echo "VNQA_STEM_RESERVE_CUS=16   : $(VNQA_STEM_RESERVE_CUS=16 python bench.py $A 2>/dev/null | q)" >> $O
  echo "VNQA_TRUNK_PRIO=none       : $(VNQA_TRUNK_PRIO=none python bench.py $A 2>/dev/null | q)" >> $O
  echo "VNQA_TRUNK_PRIO=0          : $(VNQA_TRUNK_PRIO=0 python bench.py $A 2>/dev/null | q)" >> $O
  echo "--feature-slots 3          : $(python bench.py $A --feature-slots 3 2>/dev/null | q)" >> $O
done
cat $O
