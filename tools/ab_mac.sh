#!/bin/bash
# A/B of the MAC reasoning step implementations and of the stem's CU reservation (runs on the GPU box)
run() { python bench.py --model mac --no-cpu-baseline --no-parity --repeats 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%7.1f clips/s %7.3f ms/step host %6.2f ms' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))"; }
echo "C-ABI step, stem leaves 32 CUs (default):"; run
echo "C-ABI step, stem leaves 64 CUs          :"; VNQA_STEM_RESERVE_CUS=64 run
echo "C-ABI step, no reservation              :"; VNQA_STEM_RESERVE_CUS=0 run
echo "torch / rocBLAS step, 32 CUs            :"; VNQA_MAC_CORE_TORCH=1 run
echo "C-ABI, FMA sgemm, 32 CUs                :"; VNQA_SGEMM_FMA=1 run
echo "C-ABI, no overlap                       :"; run --no-overlap
