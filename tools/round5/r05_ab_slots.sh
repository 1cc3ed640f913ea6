#!/bin/bash
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
B="python bench.py --no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2; do
echo "fp16h slots 2: $($B 2>/dev/null | q)"
echo "fp16h slots 3: $($B --feature-slots 3 2>/dev/null | q)"
echo "fp16  slots 2: $($B --precision fp16 2>/dev/null | q)"
done
PREC=fp16h bash tools/prof_step_streams.sh > gpurun_out/r05_step_streams_fp16h.txt 2>&1
grep -c . gpurun_out/r05_step_streams_fp16h.txt
