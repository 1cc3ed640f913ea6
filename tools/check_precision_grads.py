#!/usr/bin/env python3
"""Per-parameter gradients of ONE Trainer.step (the production path: gradient sinks, fused loss, loss-scaled fp16
backward) in a 16-bit precision against the exact-f32 precision on identical weights and the same minibatch, captured
from the flat gradient buffer right before the fused clip+Adam kernel consumes it.  A parameter whose norm ratio is
not ~1 (e.g. 1024 = a missed loss-scale division) is a bug the logits-only parity checks cannot see.
    python tools/check_precision_grads.py --precision fp16 [--steps 3]        (one 16-bit format per process)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B                                   # noqa: E402
from videonavqa_amd import kernels as K             # noqa: E402
from videonavqa_amd.train import Trainer            # noqa: E402


def run(args, prec, batches, steps):
    a = argparse.Namespace(**vars(args))
    a.precision = prec
    dev = torch.device("cuda", 0)
    model, stem, _, _ = B.build(a, dev)
    tr = Trainer(model, stem, lr=1e-4)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    grads, losses = [], []
    orig = K.clip_adam_step

    def spy(p, g, *rest, **kw):
        grads.append(g.clone())
        return orig(p, g, *rest, **kw)

    K.clip_adam_step = spy
    try:
        for i in range(steps):
            clip, q, v_lens, q_lens, y = batches[i % len(batches)]
            loss, _ = tr.step(clip, q, v_lens, q_lens, y)
            losses.append(float(loss))
    finally:
        K.clip_adam_step = orig
    return names, tr.fp, grads, losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16", choices=["bf16", "fp16"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--model", default="film_attn_pt")
    ap.add_argument("--soak-data", action="store_true", help="tools/soak.py's six ragged minibatches (questions up to 56 tokens)")
    o = ap.parse_args()
    args = argparse.Namespace(precision=o.precision, batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                              model=o.model)
    from videonavqa_amd import _lib as L
    L.set_half("f16" if o.precision == "fp16" else "bf16")
    dev = torch.device("cuda", 0)
    batches = B.parity_batches(args, dev)
    if o.soak_data:
        g = torch.Generator().manual_seed(7)
        batches = []
        for _ in range(6):
            clip = torch.rand(8, 3, 224, 224, 35, generator=g)
            v = torch.randint(3, 36, (8,), generator=g)
            ql = torch.randint(1, 57, (8,), generator=g)
            q = torch.randint(1, 134, (8, 56), generator=g) * (torch.arange(56)[None] < ql[:, None])
            y = torch.randint(0, 70, (8,), generator=g)
            batches.append((clip.to(dev), q.to(dev), v, ql, y.to(dev)))
    names, fp_ref, g_ref, l_ref = run(args, "fp32", batches, o.steps)
    _, fp_low, g_low, l_low = run(args, o.precision, batches, o.steps)
    out = {"losses_fp32": l_ref, "losses_" + o.precision: l_low, "steps": []}
    for s in range(o.steps):
        rows, off = {}, 0
        for n, p in zip(names, fp_ref.params):
            k = p.numel()
            a, b = g_ref[s][off:off + k], g_low[s][off:off + k]
            na, nb = float(a.norm()), float(b.norm())
            rows[n] = {"norm_fp32": na, "norm_ratio": (nb / na) if na > 0 else None,
                       "rel_l2": float((a - b).norm() / (na + 1e-30))}
            off += k
        out["steps"].append(rows)
    print(json.dumps(out, indent=1))
    bad = [(s, n, r["norm_ratio"]) for s, rows in enumerate(out["steps"]) for n, r in rows.items()
           if r["norm_ratio"] is not None and r["norm_fp32"] > 1e-7 and not 0.7 < r["norm_ratio"] < 1.4]
    print("SUSPICIOUS:" if bad else "all norm ratios within [0.7, 1.4]", bad, file=sys.stderr)


if __name__ == "__main__":
    main()
