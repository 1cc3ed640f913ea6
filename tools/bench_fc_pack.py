#!/usr/bin/env python3
"""fc_embed_attn weight re-layout kernels at the headline size (128 x 512 x 14 x 14)."""
import torch
from videonavqa_amd import kernels as K
w = torch.randn(128, 512 * 196, device="cuda")
def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("pack nat+nat_t %.1f us" % timed(lambda: K.pack_fc_weight(w, 512, 14, 14, 512, 128, torch.bfloat16)))
print("pack nat only  %.1f us" % timed(lambda: K.pack_fc_weight(w, 512, 14, 14, 512, 128, torch.bfloat16, want_t=False)))
g = torch.randn(128, 256 * 512, device="cuda")
print("unpack grad    %.1f us" % timed(lambda: K.unpack_fc_wgrad(g, 128, 512, 14, 14, 512)))
