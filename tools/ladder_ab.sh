#!/bin/bash
# ON THE GPU BOX: the workload ladder under two settings of one environment knob, interleaved:  tools/ladder_ab.sh "VNQA_TRUNK_PRIO=none"
export PYTHONPATH=$PWD
B="python bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg"
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%7.1f clips/s %6.2f ms" % (d["value"], d["ms_per_step"]))'; }
row() { name=$1; shift; echo "$name  default: $($B "$@" 2>/dev/null | q)   [$ALT]: $(env $ALT $B "$@" 2>/dev/null | q)"; }
ALT=$1
row "headline          "
row "160x208           " --height 160 --width 208
row "film_gp_pt        " --model film_gp_pt
row "time_multi_hop T70" --model time_multi_hop --frames 70
row "mac               " --model mac
row "5x1024 bs8        " --blocks 5 --channels 1024
row "bs32              " --batch 32 --steps 8 --warmup 3
row "fp16              " --precision fp16
