#!/bin/bash
# ON THE GPU BOX: the PCIe-inclusive rate against the resident one per precision and per REGION LENGTH (the round-4 record, 0.95-0.96, was
# taken on 20-step regions of the bf16 precision) + telemetry of the long bf16 regions.  Appends to gpurun_out/r05_h2d_power.txt.
mkdir -p gpurun_out; O=gpurun_out/r05_h2d_followup.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f  regions %s" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"], d["repeats"]["clips_per_s"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity"
for prec in bf16 fp16h; do
  for r in 1 2; do
    echo "$prec  20-step regions x5  resident: $(python bench.py $A --precision $prec --repeats 5 2>/dev/null | q)" >> $O
    echo "$prec  20-step regions x5  --h2d   : $(python bench.py $A --precision $prec --repeats 5 --h2d 2>/dev/null | q)" >> $O
  done
  echo "$prec  200-step regions x3 resident: $(python bench.py $A --precision $prec --repeats 3 --steps 200 2>/dev/null | q)" >> $O
  echo "$prec  200-step regions x3 --h2d   : $(python bench.py $A --precision $prec --repeats 3 --steps 200 --h2d 2>/dev/null | q)" >> $O
done
TAG=r05b STEPS=1500 bash -c 'sed "s/python bench.py \$A/python bench.py --precision bf16 \$A/" tools/h2d_power.sh > /tmp/h2d_bf16.sh; bash /tmp/h2d_bf16.sh' > /dev/null 2>&1
echo "bf16, 1500-step regions with telemetry:" >> $O; cat gpurun_out/r05b_h2d_power.txt >> $O
cat $O
