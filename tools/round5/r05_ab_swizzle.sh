#!/bin/bash
# ON THE GPU BOX: LDS key row & 7 (this build) against row & 6 (build/ab/libvnqa_hip_f16_oldswz.so, -DVNQA_C64_OLD_SWIZZLE) of the C_in = 64
# kernels: tests of those kernels, stem alone, the step, and the conflict counter of the fused conv1.
mkdir -p gpurun_out; O=gpurun_out/r05_swizzle_ab.txt; : > $O; R=$PWD
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 1200 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py tests/test_gpu_models.py -k "c64 or first or wreg or stem or objdet or vgg or golden" 2>&1 | tail -2 >> $O
timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py 2>&1 | tail -2 >> $O
q() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.1f clips/s  %.3f ms/step  stem alone %.3f" % (d["value"], d["ms_per_step"], d["config"]["stem_alone_ms"]))'; }
A="--no-cpu-baseline --no-fp16-leg --no-eval-leg --no-parity --repeats 3"
for r in 1 2 3; do
  echo "row & 7: $(python tools/stem_only.py 2>/dev/null | tail -1)   step: $(python bench.py $A 2>/dev/null | q)" >> $O
  echo "row & 6: $(VNQA_LIB=$R/build/ab/libvnqa_hip_f16_oldswz.so python tools/stem_only.py 2>/dev/null | tail -1)   step: $(VNQA_LIB=$R/build/ab/libvnqa_hip_f16_oldswz.so python bench.py $A 2>/dev/null | q)" >> $O
done
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for v in new old; do
  rm -rf /tmp/pmc_$v /tmp/kt_$v
  if [ $v = old ]; then export VNQA_LIB=$R/build/ab/libvnqa_hip_f16_oldswz.so; fi
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$v -- python3 $R/tools/stem_only.py --iters 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python3 $R/tools/stem_only.py --iters 20 > /dev/null 2>&1
  echo "== $v: counters of conv_first_c64_wide (sum over dispatches) and its average duration" >> $R/$O
  python3 - $v >> $R/$O <<'PY'
import csv, glob, sys
v = sys.argv[1]
tot = {}
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_first_c64" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print({k: "%.4g" % x for k, x in sorted(tot.items())})
if tot.get("SQ_LDS_IDX_ACTIVE"):
    print("lds_bank_conflict_frac %.4f" % (tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"]))
for f in glob.glob("/tmp/kt_%s/**/*kernel_stats.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_first_c64" in r["Name"] or "conv_wreg" in r["Name"]:
            print("%-70s calls %s avg %.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cd $R; cat $O
