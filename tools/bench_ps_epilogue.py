#!/usr/bin/env python3
"""The trunk's 3x3 conv on the patch-stationary kernel, alone on the chip: plain epilogue vs FILM_RES vs ADD_MASK, at the eval.sh
preset's size (1120 images x 14x14, 1024 channels) and at the headline's (280 images, 512 channels).  VNQA_LIB=<variant .so> selects
an A/B build (tools/build_variant.py unbatched -DVNQA_PS_EPI_UNBATCHED -DVNQA_EPI_UNBATCHED = one chunk at a time in the store loop).
GPU box:  PYTHONPATH=$PWD python tools/bench_ps_epilogue.py"""
import torch, os
from videonavqa_amd import kernels as K, _lib as L
for N, C in ((1120, 1024), (280, 512)):
    H = W = 14; dt = torch.bfloat16
    def padded():
        t = torch.zeros(N, H + 2, W + 2, C, dtype=dt, device="cuda"); t[:, 1:-1, 1:-1] = torch.randn(N, H, W, C, device="cuda").to(dt); return t
    x, res, dout = padded(), padded(), padded()
    wt3 = K.pack_conv_weight(torch.randn(C, C, 3, 3, device="cuda") / (C * 9) ** 0.5, dt)
    b = torch.randn(C, device="cuda") * 0.1
    film = torch.randn(N, 2 * C, device="cuda")
    def timed(fn, it=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
    F3 = 2.0 * N * H * W * C * C * 9
    for name, fn in (("PS plain", lambda: K.conv2d_igemm(x, wt3, bias=b, relu=False, tile=L.TILE_PS_224x256)),
                     ("PS FILM_RES", lambda: K.conv2d_igemm_film_res(x, wt3, b, film[:, :C], film[:, C:], C, res, tile=L.TILE_PS_224x256)),
                     ("PS ADD_MASK", lambda: K.conv2d_igemm_add_mask(x, wt3, dout, res, tile=L.TILE_PS_224x256))):
        us = timed(fn); print("N %4d C %4d %-16s %8.1f us %6.0f TFLOP/s" % (N, C, name, us, F3 / us / 1e6))
