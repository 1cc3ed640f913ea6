#!/bin/bash
# ON THE GPU BOX: FETCH_SIZE of the composed 5x5 on the 2-D tiles when the tiles that read the same input lines start ~3 us apart
# (libvnqa_stag1.so: the odd cout half sleeps, no XCD split; libvnqa_stag2.so: odd pixel tiles sleep, XCD split) — bf16 builds.
R=$PWD; export PYTHONPATH=$R; O=$R/gpurun_out/r06ps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name lib xcd
  rm -rf /tmp/pF
  VNQA_LIB=$2 VNQA_NO_REBUILD=1 VNQA_HALF=bf16 VNQA_COMPOSED_PS=1 VNQA_COMPOSED_PS_XCD=$3 timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pF -- python3 $R/tools/stem_only.py --iters 3 --precision bf16 > $O/stag_$1.log 2>&1 < /dev/null
  python3 - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/pF/**/*_counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no csv"); sys.exit()
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == "FETCH_SIZE" and "conv_ps_kernel<28, 2, 1>" in r["Kernel_Name"]]
print(sys.argv[1], len(v), "launches, FETCH_SIZE raw avg %.1f MB" % (sum(v) / max(len(v), 1) * 1024 / 1e6))
PY
}
run base_nosplit $R/videonavqa_amd/lib/libvnqa_hip.so 0
run stag1_nosplit $R/videonavqa_amd/lib/libvnqa_stag1.so 0
run base_split $R/videonavqa_amd/lib/libvnqa_hip.so 1
run stag2_split $R/videonavqa_amd/lib/libvnqa_stag2.so 1
