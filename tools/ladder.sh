#!/bin/bash
# Runs ON THE GPU BOX: the README's workload ladder on the current build (short bench lines, no parity / CPU legs).
export PYTHONPATH=$PWD
B="python bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg"
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.2f ms/step" % (d["value"], d["ms_per_step"]))'; }
echo "headline            $($B 2>/dev/null | q)"
echo "160x208             $($B --height 160 --width 208 2>/dev/null | q)"
echo "film_gp_pt          $($B --model film_gp_pt 2>/dev/null | q)"
echo "time_multi_hop T70  $($B --model time_multi_hop --frames 70 2>/dev/null | q)"
echo "mac                 $($B --model mac 2>/dev/null | q)"
echo "5x1024 bs8          $($B --blocks 5 --channels 1024 2>/dev/null | q)"
echo "5x1024 bs32         $($B --blocks 5 --channels 1024 --batch 32 --steps 8 --warmup 3 2>/dev/null | q)"
echo "bs32                $($B --batch 32 --steps 8 --warmup 3 2>/dev/null | q)"
echo "bf16                $($B --precision bf16 2>/dev/null | q)"
echo "fp16                $($B --precision fp16 2>/dev/null | q)"
echo "eval (inference)    $($B --mode eval 2>/dev/null | q)"
echo "film_gp_pt eval     $($B --model film_gp_pt --mode eval 2>/dev/null | q)"
echo "v_only_cnn3d        $($B --model v_only_cnn3d 2>/dev/null | q)"
echo "fp32                $(python bench.py --steps 5 --warmup 2 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg --precision fp32 2>/dev/null | q)"
