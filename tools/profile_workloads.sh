#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of bench.py for the other BASELINE configs and §8 rows
# (config 3 film_gp_pt, config 5 time_multi_hop at 70 frames, eval.sh's 5 x 1024 preset, MACNetwork, the 160x208 geometry)
# -> gpurun_out/rNN_other_workloads.md (copy to profiles/).   gpurun -- 'bash tools/profile_workloads.sh 3'
R=${1:-3}; TAG=$(printf "r%02d" $R)
ROOT=$PWD; export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/${TAG}_other_workloads.md
mkdir -p $ROOT/gpurun_out
echo "# Round $R — other workloads: bench line + rocprofv3 kernel trace (top kernels by GPU time)" > $OUT
echo "" >> $OUT
echo "Each section: \`python3 bench.py <args> --steps 10 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg\` under" >> $OUT
echo "\`rocprofv3 --kernel-trace --stats\`.  Under the profiler the launch thread becomes the bound (host enqueue = the step time), so the clips/s here are 5-30 % below the un-profiled ladder in README.md; the per-kernel durations are what this file is for.  ms/step = total kernel time / 16 steps" >> $OUT
echo "(3 priming + 3 warm-up + 10 timed); the stem-alone passes bench.py runs after the timed region inflate the stem rows by ≈1.3×." >> $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf /tmp/pw
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -- python3 $ROOT/bench.py "$@" --steps 10 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg > /tmp/pw_bench.json 2> /tmp/pw.err
  f=$(find /tmp/pw -name '*kernel_stats.csv' | head -1)
  python3 - "$name" "$f" /tmp/pw_bench.json "$*" >> $OUT <<'EOF'
import csv, json, sys
name, f, bj, args = sys.argv[1:5]
d = json.loads(open(bj).read().strip().splitlines()[-1])
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
steps = 16.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("\n## %s  (`bench.py %s`)\n" % (name, args))
print("profiled: **%.1f clips/s, %.2f ms/step**; roofline.frac %.3f; stem alone %.2f ms; host enqueue %.2f ms/step; %d launches/step, %.2f ms/step of kernel time\n"
      % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["stem_alone_ms"], d["config"]["host_enqueue_ms_per_step"],
         sum(int(r["Calls"]) for r in rows) / steps, tot))
foreign = [r for r in rows if r["Name"].lstrip().startswith(("void at::native", "at::native", "Cijk_", "MIOpen", "void at::cuda"))]
ftot = sum(float(r["TotalDurationNs"]) for r in foreign) / 1e6 / steps
worst = max(foreign, key=lambda r: float(r["TotalDurationNs"])) if foreign else None
print("kernels NOT from this library (ATen `at::native::*`, rocBLAS `Cijk_*`, MIOpen): **%.3f ms/step = %.2f %% of kernel time** over %d names"
      % (ftot, 100.0 * ftot / tot, len(foreign))
      + ("; largest: `%s` %.3f ms/step (%.2f %%)\n" % (worst["Name"][:70], float(worst["TotalDurationNs"]) / 1e6 / steps,
                                                       100.0 * float(worst["TotalDurationNs"]) / 1e6 / steps / tot) if worst else "\n"))
print("| kernel | calls/step | ms/step | avg µs |\n|---|---|---|---|")
for r in rows[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "h16")
    print("| `%s` | %.1f | %.3f | %.1f |" % (n[:90], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
EOF
}
run "BASELINE config 3: FiLM global pooling" --model film_gp_pt
run "BASELINE config 5: time multi-hop FiLM, 70 frames" --model time_multi_hop --frames 70
run "eval.sh preset: 5 blocks x 1024 channels (bs 8)" --blocks 5 --channels 1024
run "MACNetwork (dim 512, 12 steps)" --model mac
run "the reference's 160x208 geometry" --height 160 --width 208
run "fp16 storage precision" --precision fp16
cd $ROOT
tail -5 $OUT
