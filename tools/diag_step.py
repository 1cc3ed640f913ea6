#!/usr/bin/env python3
"""Staged timing of one full-size training step (prints per-phase wall time, flushes)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as Bn


def tick(msg, t0):
    torch.cuda.synchronize()
    print("[%7.2fs] %s" % (time.time() - t0, msg), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    t0 = time.time()
    dev = torch.device("cuda", 0)
    from videonavqa_amd.train import Trainer
    model, stem, vgg, od = Bn.build(args, dev)
    tick("built model+stem", t0)
    tr = Trainer(model, stem)
    clip, q, vl, ql, y = Bn.synth_batch(args, 0, dev)
    tick("synthetic batch on device", t0)
    for it in range(args.steps):
        native, v_sorted, perm = tr.extract_features(clip, vl)
        tick("step %d: stem done" % it, t0)
        model.train(); model.init_hidden()
        logits = model(native, q[perm.to(dev)], v_sorted, ql[perm])
        tick("step %d: forward done" % it, t0)
        loss = tr.loss_fn(logits, y[perm.to(dev)])
        loss.backward()
        tick("step %d: backward done" % it, t0)
        tr.fp.clip_adam_step(1e-4, 1.0)          # (clip + Adam + zero_grad AND the gradient sinks' reset)
        tick("step %d: adam done, loss %.4f" % (it, float(loss)), t0)


if __name__ == "__main__":
    main()
