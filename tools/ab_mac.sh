#!/bin/bash
# A/B of the MAC reasoning step implementations and of the stem's CU reservation (runs on the GPU box)
run() { python bench.py --model mac --no-cpu-baseline --no-parity --repeats 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%7.1f clips/s %7.3f ms/step host %6.2f ms' % (d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))"; }
echo "C-ABI step (default)                    :"; run
for n in 0 32 64 96 128 160; do echo "C-ABI step, stem leaves $n CUs          :"; VNQA_STEM_RESERVE_CUS=$n run; done
echo "C-ABI step, 64 CUs, trunk on caller stream:"; VNQA_STEM_RESERVE_CUS=64 VNQA_TRUNK_PRIO=none run
echo "torch / rocBLAS step                    :"; VNQA_MAC_CORE_TORCH=1 run
echo "torch / rocBLAS step, 64 CUs            :"; VNQA_MAC_CORE_TORCH=1 VNQA_STEM_RESERVE_CUS=64 run
echo "C-ABI, no overlap                       :"; run --no-overlap
echo "question encoder on the caller's stream :"; VNQA_MAC_SIDE_QUESTION=0 run
echo "ELUs as separate passes                 :"; VNQA_MAC_ELU_FUSED=0 run
echo "plain convs on the patch-stationary tile:"; VNQA_PLAIN_PS=1 run
echo "C-ABI step (default) again              :"; run
