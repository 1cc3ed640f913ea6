O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fp16h.py -m gpu -q -p no:cacheprovider -k "golden or goldens" > $O/t_tol.txt 2>&1; tail -3 $O/t_tol.txt
for s in 0 1; do timeout 900 python tools/diag_gp_tail.py --seed $s 2>/dev/null | tail -3; done > $O/gp_tail.txt; cat $O/gp_tail.txt
bash tools/refresh_profiles.sh 6 > $O/refresh.log 2>&1; tail -2 $O/refresh.log | cut -c1-600
bash tools/prof_corun.sh > $O/corun_attribution.txt 2>&1; head -30 $O/corun_attribution.txt | cut -c1-150
bash tools/ladder.sh > $O/ladder.txt 2>&1; cat $O/ladder.txt
