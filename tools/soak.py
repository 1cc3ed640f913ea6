#!/usr/bin/env python3
"""Stability soak: N training steps of the headline workload on fresh synthetic minibatches (ragged lengths, changing
packed-image counts, the 3-stage upload/stem/trunk pipeline), checking finite losses and a flat memory footprint."""
import argparse
import time

import torch

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--precision", default="fp16h", choices=["bf16", "fp16", "fp16h", "fp32"])
    ap.add_argument("--no-prefetch", action="store_true", help="run each step's stem inline instead of under the previous trunk")
    ap.add_argument("--every", type=int, default=25, help="print every N steps")
    ap.add_argument("--model", default="film_attn_pt", choices=["film_attn_pt", "film_gp_pt", "time_multi_hop", "mac"])
    a = ap.parse_args()
    args = argparse.Namespace(precision=a.precision, batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                              model=a.model, tail_channels=0)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if a.precision in ("fp16", "fp16h"):
        from videonavqa_amd import _lib as L
        L.set_half("f16")
    from videonavqa_amd.train import Trainer
    model, stem, _, _ = B.build(args, dev)
    tr = Trainer(model, stem, lr=1e-4)
    g = torch.Generator().manual_seed(7)

    def batch():
        clip = torch.rand(8, 3, 224, 224, 35, generator=g).pin_memory()
        v = torch.randint(3, 36, (8,), generator=g)
        ql = torch.randint(1, 57, (8,), generator=g)
        q = torch.randint(1, 134, (8, 56), generator=g) * (torch.arange(56)[None] < ql[:, None])
        y = torch.randint(0, 70, (8,), generator=g)
        return clip, q.to(dev), v, ql, y.to(dev)
    pool = [batch() for _ in range(6)]
    q = [tr.upload(pool[0][0]), tr.upload(pool[1][0])]
    peak0, t0, losses = None, time.time(), []
    for i in range(a.steps):
        cur, nxt = pool[i % 6], pool[(i + 1) % 6]
        c, n = q
        q[0], q[1] = n, tr.upload(pool[(i + 2) % 6][0])
        if a.no_prefetch:
            loss, _ = tr.step(c, cur[1], cur[2], cur[3], cur[4])
        else:
            loss, _ = tr.step(c, cur[1], cur[2], cur[3], cur[4], next_clip=n, next_v_lens_cpu=nxt[2])
        if i % a.every == a.every - 1:
            losses.append(float(loss))
            mem = torch.cuda.memory_allocated() / 2**30
            rsv = torch.cuda.memory_reserved() / 2**30
            if peak0 is None:
                peak0 = rsv
            ls = tr.loss_scaler
            print("step %4d  loss %.4f  allocated %.2f GiB  reserved %.2f GiB  %.1f clips/s%s" %
                  (i + 1, losses[-1], mem, rsv, 8 * (i + 1) / (time.time() - t0),
                   "" if ls is None else "  loss scale 2^%d, %d step(s) skipped, Adam step %d" % (
                       round(__import__("math").log2(ls.scale)), ls.skipped_steps, tr.fp.step_count)), flush=True)
            assert losses[-1] == losses[-1] and abs(losses[-1]) < 1e4, "non-finite loss"
    rsv = torch.cuda.memory_reserved() / 2**30
    assert rsv < peak0 * 1.5 + 1.0, "memory footprint grew: %.2f -> %.2f GiB" % (peak0, rsv)
    print("soak ok: %d steps, reserved %.2f GiB (first check %.2f)" % (a.steps, rsv, peak0))


if __name__ == "__main__":
    main()
