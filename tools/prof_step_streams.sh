#!/bin/bash
# ON THE GPU BOX: one overlapped training step of precision PREC (default fp16h), kernel by kernel with queues (tools/step_streams.py);
# extra bench.py arguments (e.g. --h2d) are passed through; MEMCPY=1 adds the memory-copy trace (H2D copies as rows of the timeline).
R=$PWD; export PYTHONPATH=$R; PREC=${PREC:-fp16h}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps
TR="--kernel-trace"; if [ "$MEMCPY" = "1" ]; then TR="--kernel-trace --memory-copy-trace"; fi
rocprofv3 $TR -d /tmp/ps -- python3 $R/bench.py --precision $PREC --steps 6 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg --no-fp16-leg "$@" > /tmp/ps.out 2> /tmp/ps.err
cd $R
python3 tools/step_streams.py /tmp/ps 2 ${NSTEPS:-1}
