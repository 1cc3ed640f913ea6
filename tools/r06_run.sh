O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_fp16h.py -k "f32_epilogue or mean_shifted" > $O/t_shift.txt 2>&1; tail -3 $O/t_shift.txt
for cfg in "MEAN_SHIFT=1 SPLIT_DEPTH=3" "MEAN_SHIFT=0 SPLIT_DEPTH=3" "MEAN_SHIFT=1 SPLIT_DEPTH=1"; do
  echo "==== stem alone: $cfg"; STEM_ARGS="--set $cfg" bash tools/prof_stem.sh 2>&1 | cut -c1-180
done > $O/stem_kernels_shift_ab.txt 2>&1; cat $O/stem_kernels_shift_ab.txt
R=$O/errors_depth1.txt; : > $R
for seed in 0 3; do
  for data in noise blocks; do
    echo "== 224x224 $data seed $seed" >> $R; timeout 900 python tools/error_budget.py --data $data --seed $seed --batches 12 stem.SPLIT_DEPTH=1 2>/dev/null | tail -1 >> $R
  done
  echo "== 160x208 noise seed $seed" >> $R; timeout 900 python tools/error_budget.py --height 160 --width 208 --seed $seed --batches 12 stem.SPLIT_DEPTH=1 2>/dev/null | tail -1 >> $R
  echo "== 224x224 noise seed $seed precision fp16" >> $R; timeout 900 python tools/error_budget.py --seed $seed --batches 12 --precision fp16 COH=1 2>/dev/null | tail -1 >> $R
  echo "== 224x224 blocks seed $seed precision fp16" >> $R; timeout 900 python tools/error_budget.py --seed $seed --data blocks --batches 12 --precision fp16 COH=1 2>/dev/null | tail -1 >> $R
done
cat $R
for d in 1 3; do
  timeout 600 python tools/bench_with.py stem.SPLIT_DEPTH=$d -- --no-parity --no-fp16-leg --no-cpu-baseline --no-eval-leg --steps 20 --warmup 5 > $O/bench_d$d.json 2> $O/bench_d$d.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_d$d.json").read().strip().splitlines()[-1]); print("depth $d", d["value"], d["ms_per_step"], d["config"].get("stem_alone_ms"))
except Exception as e: print("depth $d failed", e)
PY
done
timeout 600 python bench.py --precision fp16 --no-parity --no-fp16-leg --no-cpu-baseline --no-eval-leg --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16', d['value'], d['ms_per_step'], d['config'].get('stem_alone_ms'))"
