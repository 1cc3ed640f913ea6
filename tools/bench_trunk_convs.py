#!/usr/bin/env python3
"""The trunk's five conv launches alone (280 images x 14x14, 512 channels): plain igemm tiles vs the fused-epilogue forms vs the
patch-stationary kernel, us per launch and TFLOP/s — what a port of the fused epilogues to conv_ps could buy.
VNQA_BTC_N / VNQA_BTC_C set other sizes (the eval.sh preset at bs 32: 1120 images, 1024 channels; the 1x1 conv over the tile ids)."""
import os
import torch
from videonavqa_amd import kernels as K, _lib as L

N, H, W, C = int(os.environ.get("VNQA_BTC_N", 280)), 14, 14, int(os.environ.get("VNQA_BTC_C", 512))
dt = torch.bfloat16
def padded(c=C):
    t = torch.zeros(N, H + 2, W + 2, c, dtype=dt, device="cuda")
    t[:, 1:-1, 1:-1] = torch.randn(N, H, W, c, device="cuda").to(dt)
    return t
x, res, dout = padded(), padded(), padded()
w3 = torch.randn(C, C, 3, 3, device="cuda") / (C * 9) ** 0.5
w1 = torch.randn(C, C, 1, 1, device="cuda") / C ** 0.5
b = torch.randn(C, device="cuda") * 0.1
wt3, wt1 = K.pack_conv_weight(w3, dt), K.pack_conv_weight(w1, dt)
film = torch.randn(N, 2 * C, device="cuda")
frame_of = torch.arange(N, dtype=torch.int32, device="cuda") // 8
frame_off = torch.arange(0, N + 1, 8, dtype=torch.int32, device="cuda")
F3, F1 = 2.0 * N * H * W * C * C * 9, 2.0 * N * H * W * C * C

def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

rows = [("3x3 plain, tile auto (256x256)", lambda: K.conv2d_igemm(x, wt3, bias=b, relu=True), F3),
        ("3x3 plain, patch-stationary (tile 20)", lambda: K.conv2d_igemm(x, wt3, bias=b, relu=True, tile=L.TILE_PS_224x256), F3),
        ("3x3 + BNSTATS epilogue", lambda: K.conv2d_igemm_bnstats(x, wt3, b, True, frame_of, frame_off, N // 8, 8), F3),
        ("3x3 + FILM_RES epilogue", lambda: K.conv2d_igemm_film_res(x, wt3, b, film[:, :C], film[:, C:], C, res), F3),
        ("3x3 dgrad + ADD_MASK epilogue", lambda: K.conv2d_igemm_add_mask(x, wt3, dout, res), F3),
        ("1x1 plain", lambda: K.conv2d_igemm(x, wt1, bias=b, relu=True), F1)]
for tid, tname in ((L.TILE_256x128, "256x128"), (L.TILE_128x128, "128x128"), (7, "P4 256x256"), (16, "P3 256x128"), (18, "I5 256x256"),
                   (15, "512x128"), (13, "256x256 16 waves")):
    rows.append(("1x1 plain, tile %s" % tname, (lambda t: (lambda: K.conv2d_igemm(x, wt1, bias=b, relu=True, tile=t)))(tid), F1))
for name, fn, fl in rows:
    us = timed(fn)
    print("%-40s %7.1f us  %6.0f TFLOP/s" % (name, us, fl / us / 1e6))
