#!/bin/bash
# Same-box interleaved A/B of two builds of the HIP library on the headline bench:
#   tools/ab_bench.sh libvnqa_hip.so libvnqa_alt.so [rounds] [extra bench args...]
# prints clips/s per run; box-to-box variance is +-10 %, so only same-box interleaved runs are comparable.
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for L in $A $B; do
    V=$(VNQA_LIB=$PWD/videonavqa_amd/lib/$L timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["stem_alone_ms"])')
    echo "$L $V"
  done
done
