#!/bin/bash
# ON THE GPU BOX: A/B of the composed 5x5 on the igemm stem tile (VNQA_COMPOSED_PS=0) against the patch-stationary 2-D tiles (=1; VNQA_COMPOSED_PS_XCD=1:
# one cout half per XCD): stem alone (event-timed), the headline step, and the PS forms' HBM traffic (two PMC passes each, csv).
R=$PWD; export PYTHONPATH=$R; O=$R/gpurun_out/r06ps; mkdir -p $O
V="0,0 1,0 1,1"
for i in 1 2; do for v in $V; do
  echo "PS,XCD=$v $(VNQA_COMPOSED_PS=${v%,*} VNQA_COMPOSED_PS_XCD=${v#*,} timeout 120 python3 tools/stem_only.py --iters 20 2>/dev/null < /dev/null | tail -1)"
done; done | tee $O/stem_ab.txt
for i in 1 2; do for v in $V; do
  VNQA_COMPOSED_PS=${v%,*} VNQA_COMPOSED_PS_XCD=${v#*,} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-fp16-leg --no-eval-leg --no-robustness 2>/dev/null < /dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"].get("stem_alone_ms"))' | sed "s/^/PS,XCD=$v /"
done; done | tee $O/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
export VNQA_COMPOSED_PS=1
for x in 0 1; do
export VNQA_COMPOSED_PS_XCD=$x
rm -rf /tmp/pF /tmp/pW
timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pF -- python3 $R/tools/stem_only.py --iters 3 > /dev/null 2>&1 < /dev/null
timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pW -- python3 $R/tools/stem_only.py --iters 3 > /dev/null 2>&1 < /dev/null
python3 - <<'PY' > $O/traffic_ps_xcd$x.txt
import csv, glob
for d, c in (("/tmp/pF", "FETCH_SIZE"), ("/tmp/pW", "WRITE_SIZE")):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        print("no counter csv under", d); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c and ("conv_ps_kernel<28, 2" in r["Kernel_Name"]):
            agg.setdefault((r["Kernel_Name"][:60], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(c, k, len(v), sum(v) / len(v) * 1024 / 1e6, "MB raw avg")
PY
echo "XCD=$x"; cat $O/traffic_ps_xcd$x.txt
done
