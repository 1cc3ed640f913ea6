#!/bin/bash
# ON THE GPU BOX: where does the PCIe-inclusive rate's 4-5 % go?  The SMU's own telemetry (rocm-smi: socket power, gfx clock, memory clock,
# GPU use) sampled every ~0.25 s beside a LONG timed region (default 1500 steps ~ 15 s) of the resident-input bench and of `--h2d`
# (clips start in pinned host memory; 168 MB per minibatch over PCIe under the step), twice each, interleaved, on one box.
# A chip that is power-limited in both modes and clocks lower with the copy engine + PCIe PHY + extra HBM writes active says "power";
# equal clocks say "look elsewhere" (tools/prof_h2d.sh).  Writes gpurun_out/${TAG}_h2d_power.txt.
TAG=${TAG:-r05}; STEPS=${STEPS:-1500}; O=gpurun_out/${TAG}_h2d_power.txt; mkdir -p gpurun_out; : > $O
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 1 --steps $STEPS --warmup 20"
sampler() { while :; do rocm-smi -P -c --showuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.15; done; }
run() {   # $1 label, rest: bench arguments
  local label=$1; shift
  sampler > /tmp/smi_$label.jsonl & local spid=$!
  python bench.py $A "$@" 2>/dev/null | tail -1 > /tmp/bench_$label.json
  kill $spid; wait $spid 2>/dev/null
  python tools/h2d_power_summary.py $label /tmp/smi_$label.jsonl /tmp/bench_$label.json >> $O
}
for r in 1 2; do
  run resident_$r
  run h2d_$r --h2d
done
run h2d_u8 --h2d --clip-dtype u8
cat $O
