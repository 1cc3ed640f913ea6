#!/bin/bash
# ON THE GPU BOX: HBM-side traffic of the frozen stem's C_out = 512 kernels (FETCH_SIZE / WRITE_SIZE, separate --pmc passes over
# tools/stem_only.py) for VNQA_STEM_XCD_SPLIT = 0 and 1 -> gpurun_out/pmc_traffic_xcd{0,1}.json
ROOT=$PWD; export PYTHONPATH=$ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export VNQA_STEM_XCD_SPLIT=$v
  rm -rf /tmp/pF$v /tmp/pW$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pF$v -- python3 $ROOT/tools/stem_only.py --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pW$v -- python3 $ROOT/tools/stem_only.py --iters 3 > /dev/null 2>&1
  python3 $ROOT/tools/pmc_traffic.py /tmp/pF$v /tmp/pW$v "VNQA_STEM_XCD_SPLIT=$v rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 tools/stem_only.py --iters 3 (two separate passes; the frozen stem alone, 280 frames)" > $ROOT/gpurun_out/pmc_traffic_xcd$v.json
  python3 - $ROOT/gpurun_out/pmc_traffic_xcd$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, r in d["kernels"].items():
    print("xcd_split=%s %-20s read %7.1f MB  write %6.1f MB  total %7.1f MB  algorithmic %6.1f MB  ratio %.2f" % (
        sys.argv[2], k, r["hbm_read_bytes_per_launch"] / 1e6, r["hbm_write_bytes_per_launch"] / 1e6, r["hbm_bytes_per_launch"] / 1e6,
        r["algorithmic_bytes_per_launch"] / 1e6, r["traffic_over_algorithmic"]))
PY
done
