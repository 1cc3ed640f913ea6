#!/bin/bash
# Same-box interleaved A/B of environment knobs on the headline bench:
#   tools/ab_env.sh [rounds] "VAR=1" "OTHER=2 THIRD=3" ...     (the empty setting "" = defaults is always included)
# prints clips/s, stem_alone_ms, roofline.frac and the stem igemm's average launch per run.
R=${1:-2}; shift
for i in $(seq $R); do
  for S in "" "$@"; do
    V=$(env $S timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-fp16-leg --repeats 1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["stem_alone_ms"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])')
    echo "[${S:-default}] $V"
  done
done
