#!/usr/bin/env python3
"""ON THE GPU BOX: the 3x3 512 -> 512 conv on 280 x 14 x 14 maps (conv31 / conv32 / conv_init at the headline) alone on the chip:
one fp16 product (patch-stationary kernel) and the split contraction as three fp16 products (plain and [hi | lo | hi] output).
ms per launch, TFLOP/s on the FLOPs the launch executes.  (The MX-fp8 form of the corrections: tools/experiments/mx/README.md.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from videonavqa_amd import _lib as L  # noqa: E402
L.set_half("f16")
from videonavqa_amd import kernels as K  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    n, h, w, c = 280, 14, 14, 512
    g = torch.Generator().manual_seed(0)
    v = torch.zeros(n, h + 2, w + 2, c)
    v[:, 1:-1, 1:-1] = torch.randn(n, h, w, c, generator=g).abs()
    v = v.cuda()
    hi = v.half()
    lo = (v - hi.float()).half()
    tri = torch.cat([hi, lo, hi], dim=-1).contiguous()
    w4 = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(w4, torch.float32)
    wt16 = K.pack_conv_weight(w4, torch.float16)
    bias = torch.zeros(c, device="cuda")
    out1 = K.empty_padded((n, h + 2, w + 2, c), torch.float16, "cuda")
    out3 = K.empty_padded((n, h + 2, w + 2, 3 * c), torch.float16, "cuda")
    gf = 2.0 * n * h * w * c * c * 9 / 1e9
    rows = [("one fp16 product (plain in, plain out)", lambda: K.conv2d_igemm(hi, wt16, bias=bias, relu=True, out=out1, tile=L.TILE_STEM_PS_224x256)),
            ("three fp16 products ([hi|lo|hi] in, plain out)", lambda: K.conv2d_igemm(tri, wt32, bias=bias, relu=True, out=out1, split_in=True, tile=L.TILE_STEM_PS_224x256)),
            ("three fp16 products ([hi|lo|hi] in and out)", lambda: K.conv2d_igemm(tri, wt32, bias=bias, relu=True, out=out3, split_in=True, dual_out=3, tile=L.TILE_STEM_PS_224x256))]
    for name, fn in rows:
        ms = timed(fn)
        print("%-52s %7.3f ms   %6.0f TFLOP/s executed" % (name, ms, gf * (3 if "three" in name else 1) / ms))


if __name__ == "__main__":
    main()
