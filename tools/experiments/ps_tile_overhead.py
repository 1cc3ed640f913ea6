#!/usr/bin/env python3
"""ON THE GPU BOX: per-K-step time and per-tile overhead of the patch-stationary conv kernel, from launches that differ ONLY in C_in
(the K loop's length): time = rounds x (ksteps x t + T_o).  5x5 on 56x56 maps (the composed pair's geometry, pool) and 3x3 on 28x28."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videonavqa_amd import kernels as K, _lib as L   # noqa: E402

L.set_half("f16")
dt = L.half_dtype()


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


N = 280
for name, hw, k, halo, pool, cins in (("5x5 56x56 pool", 56, 5, 2, True, (64, 128, 256, 512)), ("3x3 28x28", 28, 3, 1, False, (128, 256, 512, 1024))):
    rows = []
    for cin in cins:
        x = torch.zeros(N, hw + 2 * halo, hw + 2 * halo, cin, dtype=dt, device="cuda")
        x[:, halo:-halo, halo:-halo] = torch.randn(N, hw, hw, cin, device="cuda").to(dt)
        w = torch.randn(512, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        wt = K.pack_conv_weight(w, dt, c_out_pad=512, c_in_pad=cin)
        ho = hw // 2 if pool else hw
        out = torch.zeros(N, ho + 2, ho + 2, 512, dtype=dt, device="cuda")
        us = timed(lambda: K.conv2d_igemm(x, wt, relu=True, pool2=pool, x_halo=halo, y_halo=1, out=out, tile=L.TILE_STEM_PS_224x256,
                                          desc_flags=L.CONV_XCD_SPLIT_N if k == 5 else 0))
        tiles = N * hw * hw // 224 * 2
        rounds = tiles / 256.0
        ksteps = cin // 64 * k * k
        rows.append((cin, ksteps, us, rounds))
        print("%s  C_in %4d  %3d K-steps/tile  %8.1f us  %6.2f rounds of 256 tiles  %7.1f us/round  %6.0f TFLOP/s"
              % (name, cin, ksteps, us, rounds, us / rounds, 2.0 * N * hw * hw * cin * 512 * k * k / us / 1e6))
    for (c0, k0, u0, r), (c1, k1, u1, _) in zip(rows, rows[1:]):
        t = (u1 - u0) / r / (k1 - k0)
        print("   C_in %d -> %d: t = %.3f us per K-step (MFMA-bound: %.3f at 1.9 GHz), T_o = %.1f us per tile" % (c0, c1, t, 1792 / 1900.0, u0 / r - k0 * t))

# the composed pair's launch as the stem issues it: bias, per-channel ReLU floor, border correction (16-bit ring), pool
hw, cin, k, halo = 56, 128, 5, 2
x = torch.zeros(N, hw + 4, hw + 4, cin, dtype=dt, device="cuda")
x[:, 2:-2, 2:-2] = torch.randn(N, hw, hw, cin, device="cuda").to(dt)
wt = K.pack_conv_weight(torch.randn(512, cin, 5, 5, device="cuda") / (cin * 25) ** 0.5, dt, c_out_pad=512, c_in_pad=cin)
out = torch.zeros(N, 30, 30, 512, dtype=dt, device="cuda")
bias = torch.randn(512, device="cuda")
floor = -torch.rand(512, device="cuda")
ring = (torch.randn(N, 2 * hw + 2 * (hw - 2), 512, device="cuda") * 0.1).to(dt)
for name, kw in (("plain", {}), ("+ bias", dict(bias=bias)), ("+ bias + floor", dict(bias=bias, relu_floor=floor)),
                 ("+ bias + floor + border ring", dict(bias=bias, relu_floor=floor, border_sub=ring))):
    us = timed(lambda: K.conv2d_igemm(x, wt, relu=True, pool2=True, x_halo=2, y_halo=1, out=out, tile=L.TILE_STEM_PS_224x256,
                                      desc_flags=L.CONV_XCD_SPLIT_N, **kw))
    print("5x5 56x56 pool C_in 128  %-30s %8.1f us  %6.0f TFLOP/s" % (name, us, 2.0 * N * hw * hw * cin * 512 * 25 / us / 1e6))
