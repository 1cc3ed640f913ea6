O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -rA --durations=12 > $O/gpu_tests_full.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests_full.txt | tail -20
for p in fp16 fp16h; do timeout 600 python tools/measure_smallnet_tol.py $p 2>/dev/null | tail -1; done > $O/smallnet_tol.txt; cat $O/smallnet_tol.txt
VNQA_TEST_LOW_PRECISION=bf16 VNQA_HALF=bf16 timeout 600 python tools/measure_smallnet_tol.py bf16 2>/dev/null | tail -1 >> $O/smallnet_tol.txt; tail -1 $O/smallnet_tol.txt
bash tools/prof_corun.sh > $O/corun_attribution.txt 2>&1; head -40 $O/corun_attribution.txt | cut -c1-150
