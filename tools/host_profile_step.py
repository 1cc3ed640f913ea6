#!/usr/bin/env python3
"""ON THE GPU BOX: the launch thread's cost of one OVERLAPPED training step (Trainer.step incl. the next minibatch's stem prefetch) on an
idle queue — synchronize, enqueue one step, stop the clock — and a cProfile of five such steps by cumulative / own time: where the
Python side of a step goes (ctypes calls into the C ABI, autograd bookkeeping, torch glue ops, allocator).

  python tools/host_profile_step.py [--model film_attn_pt|mac|...] [--precision fp16h] [--top 40]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="film_attn_pt")
    ap.add_argument("--precision", default="fp16h")
    ap.add_argument("--top", type=int, default=40)
    o = ap.parse_args()
    from videonavqa_amd import _lib as L
    from videonavqa_amd.train import Trainer
    L.set_half("bf16" if o.precision == "bf16" else "f16")
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(precision=o.precision, model=o.model, batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                              tail_channels=0, seed=0, clip_dtype="f32", h2d=False)
    model, stem, _, _ = bench.build(args, dev)
    tr = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum")
    batches = [bench.synth_batch(args, 0, dev, index=i) for i in range(2)]

    def step(i):
        b, bn = batches[i % 2], batches[(i + 1) % 2]
        return tr.step(*b, next_clip=bn[0], next_v_lens_cpu=bn[2])
    for i in range(6):
        step(i)
    times = []
    for i in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        times.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("host enqueue of one step on an idle queue: median %.2f ms (min %.2f, max %.2f)" %
          (sorted(times)[3] * 1e3, min(times) * 1e3, max(times) * 1e3))
    pr = cProfile.Profile()
    for i in range(5):
        torch.cuda.synchronize()
        pr.enable()
        step(i)
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    print("\n== by cumulative time (5 steps) ==")
    st.sort_stats("cumulative").print_stats(o.top)
    print("\n== by own time (5 steps) ==")
    st.sort_stats("tottime").print_stats(o.top)


if __name__ == "__main__":
    main()
