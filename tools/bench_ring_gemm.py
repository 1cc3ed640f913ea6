#!/usr/bin/env python3
"""Micro-benchmark of the composed conv's border-correction GEMMs (280 frames, 56x56 maps, 128 -> 512 -> 512):
the ring conv11 GEMM [61600 x 1152] x [512 x 1152]^T and the four edge products, per-edge vs grouped."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videonavqa_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n, H, W, ci, cm, co = 280, 56, 56, 128, 512, 512
    R = 2 * (W + 2) + 2 * H
    dt = torch.bfloat16
    a1 = torch.randn(n * R, 9 * ci, device="cuda").to(dt)
    w1 = (torch.randn(cm, 9 * ci, device="cuda") / 34).to(dt)
    b1 = torch.randn(cm, device="cuda")
    us = timeit(lambda: K.gemm_nt(a1, w1, bias=b1, split_k=False))
    print("ring conv11 GEMM          : %7.1f us  %6.0f TFLOP/s" % (us, 2.0 * n * R * 9 * ci * cm / us / 1e6))
    y1 = torch.randn(n * R, cm, device="cuda").to(dt)
    we = (torch.randn(4, co, 3 * cm, device="cuda") / 39).to(dt)
    fl = 2.0 * n * (2 * W + 2 * H) * 3 * cm * co
    us = timeit(lambda: [K.ring_edge_gather(y1, n, H, W, e) for e in range(4)])
    print("4 edge gathers            : %7.1f us" % us)
    us = timeit(lambda: K.ring_edge_gather_all(y1, n, H, W))
    print("grouped gather            : %7.1f us" % us)
    ops = [K.ring_edge_gather(y1, n, H, W, e) for e in range(4)]
    us = timeit(lambda: [K.gemm_nt(ops[e], we[e], split_k=False) for e in range(4)])
    print("4 edge GEMMs              : %7.1f us  %6.0f TFLOP/s" % (us, fl / us / 1e6))
    oa = K.ring_edge_gather_all(y1, n, H, W)
    us = timeit(lambda: K.gemm_nt_grouped(oa, we))          # (the library picks the grouped GEMM's tile: no env hook since round 4)
    print("grouped GEMM              : %7.1f us  %6.0f TFLOP/s" % (us, fl / us / 1e6))
    parts = [K.gemm_nt(ops[e], we[e], split_k=False) for e in range(4)]
    us = timeit(lambda: K.ring_assemble(parts[0], parts[1], parts[2], parts[3], n, H, W))
    print("ring_assemble             : %7.1f us" % us)
    x = torch.randn(n, H + 4, W + 4, ci, device="cuda").to(dt)
    us = timeit(lambda: K.ring_im2col(x, H, W))
    print("ring_im2col               : %7.1f us" % us)


if __name__ == "__main__":
    main()
