#!/usr/bin/env python3
"""Same-process A/B of the weights-in-registers direct conv (vnqa_conv2d_wreg_fwd) against the kernels it replaces on the
VGG front's short-K layers: interleaved rounds, median / min ms and TFLOP/s per arm, plus a bit-level comparison."""
import argparse
import json
import statistics

import torch

from videonavqa_amd import kernels as K
from videonavqa_amd import _lib as L

LAYERS = [("conv1_2", 224, 64, 64, True), ("conv2_1", 112, 64, 128, False), ("conv2_2", 112, 128, 128, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=280)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="conv1_2,conv2_1,conv2_2")
    a = ap.parse_args()
    dt = torch.bfloat16
    N = a.frames
    for name, H, Cin, Cout, pool in LAYERS:
        if name not in a.layers.split(","):
            continue
        W = H
        x = torch.zeros(N, H + 2, W + 2, Cin, dtype=dt, device="cuda")
        x[:, 1:-1, 1:-1, :] = torch.randn(N, H, W, Cin, device="cuda").to(dt)
        wt = (torch.randn(Cout, 9, Cin, device="cuda") / (9 * Cin) ** 0.5).to(dt)
        b = torch.randn(Cout, device="cuda") * 0.1
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out_a = torch.zeros(N, Ho + 2, Wo + 2, Cout, dtype=dt, device="cuda")
        out_b = torch.zeros_like(out_a)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        arms = {"wreg": lambda: K.conv2d_wreg(x, wt, bias=b, relu=True, pool2=pool, out=out_a)}
        if Cin == 64:
            arms["c64"] = lambda: K.conv2d_c64(x, wt, bias=b, relu=True, pool2=pool, out=out_b)
        else:
            arms["igemm512x128"] = lambda: K.conv2d_igemm(x, wt, bias=b, relu=True, pool2=pool, out=out_b, tile=15)
        for f in arms.values():
            f()
        torch.cuda.synchronize()
        diff = float((out_a.float() - out_b.float()).abs().max() / (out_b.float().abs().max() + 1e-12))
        times = {k: [] for k in arms}
        for _ in range(a.rounds):
            for k, f in arms.items():
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(a.iters):
                    f()
                en.record()
                torch.cuda.synchronize()
                times[k].append(st.elapsed_time(en) / a.iters)
        for k, v in times.items():
            med = statistics.median(v)
            print(json.dumps(dict(layer=name, arm=k, ms_median=round(med, 4), ms_min=round(min(v), 4),
                                  tflops_median=round(flops / med / 1e9, 1), frames=N, rel_diff_vs_other=diff)), flush=True)
        del x, out_a, out_b


if __name__ == "__main__":
    main()
