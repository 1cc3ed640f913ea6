#!/bin/bash
# ON THE GPU BOX: the other ladder rows on the final build (default precision fp16h unless stated), one bench line each
mkdir -p gpurun_out; O=gpurun_out/r05_ladder.txt; : > $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%9.1f %s  %.3f ms/step  %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("dtype","")[:5]))'; }
A="--no-cpu-baseline --no-parity --no-fp16-leg --no-eval-leg"
run() { echo "$(printf '%-64s' "$*") $(timeout 600 python bench.py $A "$@" 2>/dev/null | q)" >> $O; }
run
run --precision bf16
run --precision fp16
run --mode eval
run --height 160 --width 208
run --model film_gp_pt
run --model time_multi_hop --frames 70
run --model mac
run --model mac --precision bf16
run --model v_only_cnn3d
run --batch 32 --blocks 5 --channels 1024
run --batch 32 --blocks 5 --channels 1024 --precision bf16
run --h2d
cat $O
