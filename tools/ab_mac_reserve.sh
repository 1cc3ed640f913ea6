#!/bin/bash
# ON THE GPU BOX: `bench.py --model mac` over the stem's CU reservation (high-priority trunk stream on)
run() { python bench.py --model mac --no-cpu-baseline --no-parity --repeats 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%7.1f clips/s %7.3f ms/step stem alone %.2f ms' % (d['value'], d['ms_per_step'], d['config']['stem_alone_ms']))"; }
for n in ${@:-0 96 128 160 192 128 0}; do echo "reserve $n: $(VNQA_STEM_RESERVE_CUS=$n run)"; done
