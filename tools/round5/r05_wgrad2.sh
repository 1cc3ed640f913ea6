#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_wgrad2.txt; : > $O; R=$PWD
timeout 300 python tools/experiments/dbg_wgrad.py 2>&1 | grep "max diff" | head -8 >> $O
VNQA_TEST_LOW_PRECISION=fp16 VNQA_HALF=f16 timeout 900 python -m pytest -q -m gpu -x -p no:cacheprovider tests/test_gpu_conv.py -k "wgrad" 2>&1 | tail -3 >> $O
echo "== full" >> $O; timeout 600 python tools/bench_wgrad.py 2>/dev/null | grep "ring" >> $O
echo "== no global->LDS transfers in the loop" >> $O; VNQA_LIB=$R/build/ab/libvnqa_hip_f16_wg4_nodma.so timeout 600 python tools/bench_wgrad.py 2>/dev/null | grep "ring" >> $O
echo "== no MFMAs" >> $O; VNQA_LIB=$R/build/ab/libvnqa_hip_f16_wg4_nomfma.so timeout 600 python tools/bench_wgrad.py 2>/dev/null | grep "ring" >> $O
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -- python3 $R/tools/bench_wgrad.py > /dev/null 2>&1
python3 - >> $R/$O <<'PY'
import csv, glob
for f in glob.glob("/tmp/pw/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Name"] or "slab" in r["Name"] or "colsum" in r["Name"]:
            print("%-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf /tmp/pc; rocprofv3 --kernel-trace --pmc $pm --output-format csv -d /tmp/pc -- python3 $R/tools/bench_wgrad.py > /dev/null 2>&1
  python3 - >> $R/$O <<'PY'
import csv, glob
tot = {}
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "wgrad4" if "wgrad4" in n else ("wgrad8" if "conv_wgrad_kernel" in n else None)
        if k and int(r["Grid_Size"]) in (252 * 256, 252 * 512):
            tot.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in sorted(tot.items()):
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(d.items())}, "n=%d" % len(next(iter(d.values()))))
PY
done
cd $R; cat $O
