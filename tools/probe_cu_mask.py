#!/usr/bin/env python3
"""Frozen stem alone on a CU-masked stream (vnqa_stream_create_reserved) for several reservations: ms per pass."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from videonavqa_amd import _lib as L
from videonavqa_amd.models.common import FrameLayout

a = argparse.Namespace(precision="bf16", model="film_attn_pt", blocks=1, channels=512, tail_channels=0, batch=8, frames=35, height=224, width=224)
stem = bench.build(a, torch.device("cuda"))[1]
clip = torch.randn(8, 3, 224, 224, 35, device="cuda")
lay = FrameLayout([35] * 8, 35, "cuda")
def run(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3): stem.forward_clip(clip, lay.img_of, lay.n_img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): stem.forward_clip(clip, lay.img_of, lay.n_img)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
print("plain stream            : %.2f ms" % run(torch.cuda.Stream()))
for r in (0, 32, 64):
    st = L.reserved_stream(r)
    stem.reserve_cus = r          # (per-call descriptor field of the persistent kernels: the library keeps no global setting)
    print("masked stream, reserve %2d (persistent grids %d): %.2f ms" % (r, 256 - r, run(st)))
stem.reserve_cus = 0
print("plain stream again      : %.2f ms" % run(torch.cuda.Stream()))

import ctypes
def masked(bits_clear):
    words = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    for i in bits_clear: words[i // 32] &= ~(1 << (i % 32)) & 0xFFFFFFFF
    out = ctypes.c_void_p()
    L.check(L.lib().vnqa_stream_create_masked(ctypes.cast(words, ctypes.c_void_p), 8, ctypes.byref(out)), "mask")
    return torch.cuda.ExternalStream(out.value)
# only the non-persistent kernels tell the mapping apart cleanly, but the whole stem is what matters
for name, clr in (("bit 0", [0]), ("bits 0-7", range(8)), ("bits 0-31", range(32)), ("bits 224-255", range(224, 256)), ("i%8==7", [i for i in range(256) if i % 8 == 7]),
                  ("i%16==15 (16)", [i for i in range(256) if i % 16 == 15]), ("i%32>=28 (32)", [i for i in range(256) if i % 32 >= 28]),
                  ("i%64>=56 (32)", [i for i in range(256) if i % 64 >= 56]), ("bits 0-127", range(128))):
    clr = list(clr)
    stem.reserve_cus = 0
    t_full = run(masked(clr))
    stem.reserve_cus = (len(clr) + 7) // 8 * 8
    t_fit = run(masked(clr))
    print("cleared %-16s (%3d CUs): %.2f ms with 256-WG persistent grids, %.2f ms with fitted grids" % (name, len(clr), t_full, t_fit))
stem.reserve_cus = 0
