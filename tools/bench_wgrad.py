#!/usr/bin/env python3
"""ON THE GPU BOX: the 16-bit weight-gradient kernel alone — the default 4-wave ring form against the first 8-wave form
(VNQA_WGRAD_EIGHT_WAVES) — on the trunk's shapes: us per call (kernel + slab reduce + bias column sums) and TFLOP/s."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videonavqa_amd import kernels as K, _lib as L   # noqa: E402

if "--bf16" not in sys.argv:
    L.set_half("f16")
dt = L.half_dtype()


def padded(n, h, w, c):
    t = torch.zeros(n, h + 2, w + 2, c, dtype=dt, device="cuda")
    t[:, 1:-1, 1:-1] = torch.randn(n, h, w, c, device="cuda").to(dt)
    return t


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


SHAPES = ((280, 14, 14, 512, 512, 9, 1), (280, 14, 14, 512, 512, 9, 3), (280, 14, 14, 512, 512, 1, 1),
          (1120, 14, 14, 1024, 1024, 9, 1), (280, 20, 26, 512, 512, 9, 1))
if "--headline-only" in sys.argv:
    SHAPES = SHAPES[:1]
for n, h, w, cin, cout, taps, segs in SHAPES:
    x, dy = padded(n, h, w, cin * segs), padded(n, h, w, cout)
    fl = 2.0 * n * h * w * cin * cout * taps
    for name, ew, cs in (("4-wave rows", False, False), ("4-wave ring", False, True), ("8-wave", True, False)):
        us = timed(lambda: K.conv2d_wgrad(x, dy, taps, x_segs=segs, eight_waves=ew, compact_stages=cs))
        print("%4d x %dx%d  %4d -> %4d  taps %d  x segments %d  %-12s %7.1f us  %6.0f TFLOP/s" % (n, h, w, cin, cout, taps, segs, name, us, fl / us / 1e6))
