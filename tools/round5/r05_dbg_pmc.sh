#!/bin/bash
R=$PWD; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcF
( time rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-parity --no-cpu-baseline --no-overlap ) > /tmp/f.out 2> /tmp/f.err
echo "rc $?"; tail -5 /tmp/f.out | cut -c1-300; grep -v "^W2026\|^E2026.*Opened" /tmp/f.err | tail -25 | cut -c1-300; find /tmp/pmcF -name "*.csv" | head
