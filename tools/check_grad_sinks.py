#!/usr/bin/env python3
"""GPU diagnostic: one training step with the gradient sinks on (VNQA_DIRECT_GRADS=1: kernels write parameter gradients
straight into the flat buffer) and off (autograd AccumulateGrad) from identical weights — prints the per-parameter
difference of the flat gradient (expected: exactly 0 everywhere)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_trainer import _setup
from videonavqa_amd.train import Trainer
import videonavqa_amd.train as T
res = {}
for flag in ("0", "1"):
    os.environ["VNQA_DIRECT_GRADS"] = flag
    model, stem, batches = _setup(seed=10)
    tr = Trainer(model, stem, lr=1e-3)
    cap = {}
    orig = T.K.clip_adam_step
    def spy(p, g, m, v, partial, step, lr, clip=1.0, **kw):
        cap["g"] = g.clone()
        return orig(p, g, m, v, partial, step, lr, clip, **kw)
    T.K.clip_adam_step = spy
    tr.step(*batches[0])
    T.K.clip_adam_step = orig
    res[flag] = (cap["g"], [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad])
g0, names = res["0"]; g1, _ = res["1"]
off = 0
for n, k in names:
    a, b = g0[off:off+k], g1[off:off+k]
    err = float((a-b).abs().max()) / (float(a.abs().max()) + 1e-12)
    print("%-32s %8d  rel err %.3e  |g| %.3e" % (n, k, err, float(a.abs().max())))
    off += k
