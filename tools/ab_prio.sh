for S in "" "VNQA_STEM_PRIO=-1" "VNQA_TRUNK_PRIO=-1" "VNQA_STEM_PRIO=-1 VNQA_TRUNK_PRIO=0"; do
  V=$(env $S timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-fp16-leg --repeats 1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["stem_alone_ms"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])')
  echo "[${S:-default}] $V"
done
