#!/bin/bash
# ON THE GPU BOX: which ingredient of --h2d costs the bf16 step its 3.5 %?  (fp16h: 0.3 %.)  200-step regions, interleaved twice:
#   resident | pinonly (pinned staging clips exist) | copyonly (+ one 168-MB H2D copy per step into scratch, no dependency) | --h2d
# then PMC passes (program directly behind `rocprofv3 ... --`) of the stem kernels in the resident and the --h2d process.
mkdir -p gpurun_out; O=gpurun_out/r05_h2d_ablation.txt; : > $O; R=$PWD
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f  regions %s" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"], d["repeats"]["clips_per_s"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --precision bf16 --repeats 3 --steps 200"
for r in 1 2; do
  echo "resident : $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "pinonly  : $(python bench.py $A --h2d-ablation pinonly 2>/dev/null | q)" >> $O
  echo "copyonly : $(python bench.py $A --h2d-ablation copyonly 2>/dev/null | q)" >> $O
  echo "--h2d    : $(python bench.py $A --h2d 2>/dev/null | q)" >> $O
done
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
B="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --precision bf16 --repeats 1 --steps 8 --warmup 2"
for mode in resident h2d; do
  X=""; if [ $mode = h2d ]; then X="--h2d"; fi
  for pm in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pc; rocprofv3 --kernel-trace --pmc $pm --output-format csv -d /tmp/pc -- python3 $R/bench.py $B $X > /dev/null 2>&1
    python3 - $mode >> $R/$O <<'PY'
import csv, glob, sys
mode = sys.argv[1]
tot, dur = {}, {}
names = {"conv_igemm_kernel<unsigned short, 256, 256, 2, 4, 1": "composed 5x5", "conv_ps_kernel<28": "conv_ps<28>", "conv_first_c64": "fused conv1",
         "conv_wreg_kernel<128": "conv2_2", "conv_ps_kernel<14": "conv_ps<14>"}
def key(n):
    for k, v in names.items():
        if k in n:
            return v
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            tot.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob("/tmp/pc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(tot):
    print("%-8s %-13s avg %8.1f us (n=%d)  %s" % (mode, k, sum(dur.get(k, [0])) / max(1, len(dur.get(k, []))), len(dur.get(k, [])),
          "  ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(tot[k].items()))))
PY
  done
done
cd $R; cat $O
