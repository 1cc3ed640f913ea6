#!/usr/bin/env python3
"""The composed 5x5 conv (128 -> 512 on 280 x 56 x 56 maps, ReLU + pool) alone: with / without the border correction,
tile variants."""
import sys
import torch
from videonavqa_amd import kernels as K, _lib as L

N, H, W, Ci, Co = 280, 56, 56, 128, 512
dt = torch.bfloat16
x = torch.zeros(N, H + 4, W + 4, Ci, dtype=dt, device="cuda")
x[:, 2:-2, 2:-2] = torch.randn(N, H, W, Ci, device="cuda").to(dt)
w = torch.randn(Co, Ci, 5, 5, device="cuda") / (Ci * 25) ** 0.5
b = torch.randn(Co, device="cuda") * 0.1
ring = (torch.randn(N, 2 * W + 2 * (H - 2), Co, device="cuda") * 0.1).to(dt)
out = torch.zeros(N, H // 2 + 2, W // 2 + 2, Co, dtype=dt, device="cuda")
flops = 2.0 * N * H * W * Ci * Co * 25


def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

for rep in range(2):
    for tile, tiled in ((6, True), (21, False), (19, True)):
        wt = K.pack_conv_weight_tiled(w, dt, tile, c_out_pad=Co, c_in_pad=Ci) if tiled else K.pack_conv_weight(w, dt, c_out_pad=Co, c_in_pad=Ci)
        for sub in (ring, None):
            ms = timed(lambda: K.conv2d_igemm(x, wt, bias=b, relu=True, pool2=True, x_halo=2, y_halo=1, out=out, tile=tile, border_sub=sub))
            print("tile %2d %s %s: %.3f ms  %.0f TFLOP/s" % (tile, "tiled  " if tiled else "k-major", "border_sub" if sub is not None else "no corr   ", ms, flops / ms / 1e9))
