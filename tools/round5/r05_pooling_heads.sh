#!/bin/bash
# ON THE GPU BOX: the pooling heads (BASELINE configs 3 and 5) at full size, 16-bit precision vs exact fp32: logits, arg-max frame flips, gradients
mkdir -p gpurun_out
for cfg in "film_gp_pt 35" "time_multi_hop 70"; do
  set -- $cfg
  for prec in ${PRECS:-fp16h fp16 bf16}; do
    timeout 600 python bench.py --model $1 --frames $2 --precision $prec --parity-only 2>/dev/null | tail -1 | python -c "
import sys, json
p = json.loads(sys.stdin.read())
k = [x for x in p if x.endswith('_logits_rel_err')][0]
ph = p.get('pooling_head') or {}
print('%-15s T=%s %-6s logits %s  argmax %s  flips %.4f  grad_rel_l2 %.4f  routed %.4f' % ('$1', '$2', '$prec', ['%.2e' % v for v in p[k + '_per_batch']], p['argmax_equal_at_init'], ph.get('argmax_frame_flip_frac', -1), p['grad_rel_l2_err'], ph.get('grad_rel_l2_err_routed_by_fp32_argmax', -1)))"
  done
done
