#!/bin/bash
# ON THE GPU BOX: side-stream FiLM generator on / off over batch sizes (and the eval.sh 5 x 1024 preset at bs 32)
export PYTHONPATH=$PWD
B="python bench.py --steps 8 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg"
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%7.1f clips/s %6.2f ms" % (d["value"], d["ms_per_step"]))'; }
for bs in 16 32; do
  for r in 1 2; do
    echo "bs $bs  side on: $($B --batch $bs 2>/dev/null | q)   off: $(VNQA_SIDE_LSTM=0 $B --batch $bs 2>/dev/null | q)"
  done
done
echo "5x1024 bs32  side on: $($B --batch 32 --blocks 5 --channels 1024 2>/dev/null | q)   off: $(VNQA_SIDE_LSTM=0 $B --batch 32 --blocks 5 --channels 1024 2>/dev/null | q)"
