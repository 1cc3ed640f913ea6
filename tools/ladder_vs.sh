#!/bin/bash
# Same-box regression sweep of the CURRENT tree against an older revision of this repo, whole ladder + config 2:
#   1. here (no GPU):   bash tools/ladder_vs.sh prepare <git-rev>     # worktree _old/ at <git-rev>, its libraries built in it
#   2. on the GPU box:  gpurun -- 'bash tools/ladder_vs.sh run'        # every ladder row for both trees, interleaved
#   3. here:            bash tools/ladder_vs.sh clean
# (_old/ travels with the gpurun snapshot; it is excluded from git through .git/info/exclude.)  This is how round 3 found a 5.5 %
# stem regression (an ELU branch in a shared conv epilogue) and an 8 % one in VideoOnlyCNN3D (a fused store loop's registers
# halving the occupancy of an unrelated tile instantiation) that no test and no single-config bench had shown.
set -e
case "$1" in
  prepare)
    git worktree add -f _old "$2"
    grep -qx "_old/" .git/info/exclude 2>/dev/null || echo "_old/" >> .git/info/exclude
    (cd _old && PYTHONPATH=$PWD python -m videonavqa_amd.build | tail -2) ;;
  clean)
    git worktree remove --force _old; git worktree prune; sed -i '/^_old\/$/d' .git/info/exclude ;;
  run)
    R=$PWD
    B="--steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline"
    grep -q -- "--no-fp16-leg" _old/bench.py && B="$B --no-fp16-leg"
    one() { (cd $1 && PYTHONPATH=$1 python bench.py $B $([ $1 = $R ] && echo --no-fp16-leg) "${@:2}" 2>/dev/null | tail -1 | python -c 'import sys,json
try: print("%8.1f" % json.loads(sys.stdin.read())["value"])
except Exception: print("nan")'); }       # (nan: the older tree's bench.py does not know the row's flags)
    row() { name=$1; shift; a=$(one $R "$@"); b=$(one $R/_old "$@"); a2=$(one $R "$@"); b2=$(one $R/_old "$@")
            python3 -c "import sys; n,a,b,a2,b2=sys.argv[1:]; a,b,a2,b2=map(float,(a,b,a2,b2)); print('%-22s new %8.1f %8.1f   old %8.1f %8.1f   new/old %.3f' % (n,a,a2,b,b2,(a+a2)/(b+b2)))" "$name" $a $b $a2 $b2; }
    row headline
    row 160x208 --height 160 --width 208
    row film_gp_pt --model film_gp_pt
    row "time_multi_hop T70" --model time_multi_hop --frames 70
    row mac --model mac
    row "5x1024 bs8" --blocks 5 --channels 1024
    row "5x1024 bs32" --blocks 5 --channels 1024 --batch 32 --steps 8 --warmup 3
    row bs32 --batch 32 --steps 8 --warmup 3
    row fp16 --precision fp16
    row v_only_cnn3d --model v_only_cnn3d ;;
  *) echo "usage: $0 prepare <rev> | run | clean"; exit 2 ;;
esac
