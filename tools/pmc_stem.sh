#!/bin/bash
# ON THE GPU BOX: the two PMC passes over the frozen stem alone -> gpurun_out/pmc_stem.json (MFMA-busy, stall split, LDS conflicts per kernel)
ROOT=$PWD; export PYTHONPATH=$ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcM /tmp/pmcS
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/pmcM -- python3 $ROOT/tools/stem_only.py --iters 5 > /dev/null 2>$ROOT/gpurun_out/pmcM.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcS -- python3 $ROOT/tools/stem_only.py --iters 5 > /dev/null 2>$ROOT/gpurun_out/pmcS.err
cd $ROOT
python tools/pmc_mfma.py /tmp/pmcM /tmp/pmcS > gpurun_out/pmc_stem.json
python - <<'PY'
import json
m = json.load(open("gpurun_out/pmc_stem.json"))
for r in m["kernels"][:9]:
    print("%-52s grid %8d  %.3f ms  mfma %.3f  cu_busy %.2f  wait %.2f  stall %.2f  lds_conflict %.3f" % (r["kernel"][:52], r["grid"], r["avg_ms_profiled"], r["mfma_util"], r["cu_busy_frac"], r["wait_any_frac"], r["issue_stall_frac"], r["lds_bank_conflict_frac"]))
PY
