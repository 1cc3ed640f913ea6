#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_ring_edges_ab.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  for prec in fp16h bf16; do
    echo "$prec  four 3-tap launches: $(python bench.py $A --precision $prec 2>/dev/null | q)" >> $O
    echo "$prec  one 9-tap launch   : $(python tools/bench_with.py stem.RING_EDGE_LAUNCHES=0 -- $A --precision $prec 2>/dev/null | q)" >> $O
  done
done
cat $O
