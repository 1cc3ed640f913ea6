#!/bin/bash
# ON THE GPU BOX: the co-run attribution of one training step (tools/corun_attribution.py): an overlapped and a --no-overlap kernel trace
# of the same bench.py workload (precision PREC, default fp16h); extra bench.py arguments are passed through.
R=$PWD; export PYTHONPATH=$R; PREC=${PREC:-fp16h}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pco /tmp/pcs
A="--precision $PREC --steps 6 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg --no-fp16-leg"
rocprofv3 --kernel-trace -d /tmp/pco -- python3 $R/bench.py $A "$@" > /tmp/pco.out 2> /tmp/pco.err
rocprofv3 --kernel-trace -d /tmp/pcs -- python3 $R/bench.py $A --no-overlap "$@" > /tmp/pcs.out 2> /tmp/pcs.err
cd $R
python3 tools/corun_attribution.py /tmp/pco /tmp/pcs 2
