#!/usr/bin/env python3
"""Per-layer microbench of the stem igemm convs on one MI355X (TFLOP/s per layer, tile sweep)."""
import argparse
import json
import time

import torch

from videonavqa_amd import kernels as K

LAYERS = [  # name, H, W, Cin, Cout, pool
    ("conv1_2", 224, 224, 64, 64, True), ("conv2_1", 112, 112, 64, 128, False),
    ("conv2_2", 112, 112, 128, 128, True), ("conv11", 56, 56, 128, 512, False),
    ("conv12", 56, 56, 512, 512, True), ("conv21", 28, 28, 512, 512, False),
    ("conv22", 28, 28, 512, 512, True), ("conv31", 14, 14, 512, 512, False),
    ("conv32", 14, 14, 512, 512, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=70)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--layers", default="")
    ap.add_argument("--relu-flags", type=int, default=1, help="diagnostic: value passed as the relu field")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    N = args.frames
    res = []
    layers = [l for l in LAYERS if not args.layers or l[0] in args.layers.split(",")]
    for name, H, W, Cin, Cout, pool in layers:
        x = torch.zeros(N, H + 2, W + 2, Cin, dtype=dt, device="cuda")
        x[:, 1:-1, 1:-1, :] = torch.randn(N, H, W, Cin, device="cuda").to(dt)
        wt = (torch.randn(Cout, 9, Cin, device="cuda") / (9 * Cin) ** 0.5).to(dt)
        b = torch.randn(Cout, device="cuda")
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.zeros(N, Ho + 2, Wo + 2, Cout, dtype=dt, device="cuda")
        flops = 2.0 * N * H * W * Cin * Cout * 9
        for tile in [int(t) for t in args.tiles.split(",")]:
            def run():
                if tile in (64, 65):
                    K.conv2d_c64(x, wt, bias=b, relu=True, pool2=pool, out=out, shape4=(tile == 65))
                else:
                    K.conv2d_igemm(x, wt, bias=b, relu=args.relu_flags, pool2=pool, out=out, tile=tile)
            try:
                run()
            except Exception as e:  # tile not available for this dtype / layer
                print(name, "tile", tile, "skipped:", e)
                continue
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(args.iters):
                run()
            en.record()
            torch.cuda.synchronize()
            ms = st.elapsed_time(en) / args.iters
            r = dict(layer=name, tile=tile, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1), frames=N)
            print(json.dumps(r), flush=True)
            res.append(r)
        del x, out
    tot = {}
    for r in res:
        tot.setdefault(r["layer"], []).append(r["ms"])
    best = sum(min(v) for v in tot.values())
    flops_all = sum(2.0 * N * H * W * Cin * Cout * 9 for _, H, W, Cin, Cout, _ in layers)
    print(json.dumps(dict(total_best_ms=round(best, 3), tflops=round(flops_all / best / 1e9, 1))))


if __name__ == "__main__":
    main()
