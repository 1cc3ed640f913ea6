#!/bin/bash
# Runs ON THE GPU BOX: the MFMA / LDS counter passes of tools/pmc_mfma.py over a serial (--no-overlap) training step,
# i.e. including the trunk's kernels (wgrad, dgrad, fused epilogues, LSTM) -> gpurun_out/pmc_mfma_step.json
ROOT=$PWD; export PYTHONPATH=$ROOT
ARGS="--steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-fp16-leg --no-overlap"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tM /tmp/tS
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/tM -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/tS -- python3 $ROOT/bench.py $ARGS > /dev/null 2>&1
cd $ROOT
python tools/pmc_mfma.py /tmp/tM /tmp/tS 40 > gpurun_out/pmc_mfma_step.json
