#!/bin/bash
# ON THE GPU BOX: kernel time per step by kernel name for precision PREC (default fp16h), from whole timed-region steps only
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R; PREC=${PREC:-fp16h}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk
rocprofv3 --kernel-trace -d /tmp/pk -- python3 $R/bench.py --precision $PREC --steps 12 --warmup 3 --repeats 1 --no-parity --no-cpu-baseline --no-eval-leg --no-fp16-leg "$@" > /tmp/pk.out 2> /tmp/pk.err
cd $R
tail -1 /tmp/pk.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under the profiler:', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')"
python3 tools/step_kernels.py /tmp/pk 8 4 ${TOP:-34}
