#!/bin/bash
# A/B of the MAC reasoning step implementations (runs on the GPU box)
run() { python bench.py --model mac --no-cpu-baseline --no-parity --repeats 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%7.1f clips/s %7.3f ms/step host %6.2f ms' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))"; }
echo "C-ABI step, split-K      :"; VNQA_MAC_CORE_CABI=1 run
echo "C-ABI step, no split-K   :"; VNQA_MAC_CORE_CABI=1 VNQA_MAC_SPLITK=0 run
echo "torch step (default)     :"; run
echo "C-ABI, no overlap        :"; VNQA_MAC_CORE_CABI=1 run --no-overlap
echo "C-ABI no split, no overl.:"; VNQA_MAC_CORE_CABI=1 VNQA_MAC_SPLITK=0 run --no-overlap
echo "torch, no overlap        :"; run --no-overlap
