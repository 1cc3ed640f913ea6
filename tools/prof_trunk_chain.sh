#!/bin/bash
# ON THE GPU BOX: the trunk chain's kernels of ONE step of another precision (PREC, default fp16h) with the stem NOT overlapped,
# aggregated by total time (tools/trunk_timeline.py; in fp16h the rows up to the last x3_post before lstm_seq_fwd are stem layers).
R=$PWD; export PYTHONPATH=$R; PREC=${PREC:-fp16h}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pt -- python3 $R/bench.py --precision $PREC --steps 6 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg --no-fp16-leg --no-overlap > /tmp/pt.out 2> /tmp/pt.err
cd $R
python3 tools/trunk_timeline.py /tmp/pt 70
