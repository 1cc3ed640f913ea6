O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fp16h.py tests/test_gpu_trainer.py tests/test_gpu_models.py tests/test_eval_cli.py -m gpu -q -p no:cacheprovider -rA > $O/gpu_tests_twin.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|fp16h vs fp32" $O/gpu_tests_twin.txt | tail -40
timeout 600 python bench.py --no-parity --no-fp16-leg --no-cpu-baseline --no-eval-leg --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16h', d['value'], d['ms_per_step'], d['config'].get('stem_alone_ms'), d['config'].get('host_enqueue_idle_queue_ms_per_step'))"
timeout 600 python tools/bench_with.py stem.FEATURE_TWIN=0 -- --no-parity --no-fp16-leg --no-cpu-baseline --no-eval-leg --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16h one-product conv_init', d['value'], d['ms_per_step'], d['config'].get('stem_alone_ms'))"
timeout 600 python bench.py --precision fp16 --no-parity --no-fp16-leg --no-cpu-baseline --no-eval-leg --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16', d['value'], d['ms_per_step'], d['config'].get('stem_alone_ms'))"
R=$O/pooling_heads_twin.txt; : > $R
for seed in 0 1; do
  echo "== film_gp_pt 224x224 T=35 seed $seed" >> $R; timeout 900 python tools/error_budget.py --model film_gp_pt --seed $seed --batches 3 COH=1 2>/dev/null | tail -1 >> $R
  echo "== time_multi_hop 224x224 T=70 seed $seed" >> $R; timeout 1200 python tools/error_budget.py --model time_multi_hop --frames 70 --seed $seed --batches 3 COH=1 2>/dev/null | tail -1 >> $R
done
cat $R
