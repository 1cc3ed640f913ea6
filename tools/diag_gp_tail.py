#!/usr/bin/env python3
"""ON THE GPU BOX — diagnostic: how much of the global-pooling head's 16-bit logits error (BASELINE config 3) is the LAST storage rounding —
relu(c1x1_tail(x)) stored in 16 bits right before the max over frames (film_global_pooling_pt_stem.py:228-236) — as opposed to everything
upstream?  Runs precision fp16h and, on the SAME 16-bit trunk output x, the tail in fp32 (torch); both against precision fp32's logits.
    python tools/diag_gp_tail.py [--model film_gp_pt|time_multi_hop] [--frames 35] [--seed 0] [--batches 3]"""
import argparse
import copy
import importlib.util
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench as Bn
    from videonavqa_amd import _lib as L
    from videonavqa_amd.train import Trainer
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="film_gp_pt")
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batches", type=int, default=3)
    a = ap.parse_args()
    L.set_half("f16")
    spec = importlib.util.spec_from_file_location("error_budget", os.path.join(ROOT, "tools", "error_budget.py"))
    eb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eb)
    args = argparse.Namespace(precision="fp32", batch=8, frames=a.frames, height=224, width=224, blocks=1, channels=512,
                              model=a.model, tail_channels=0, seed=a.seed)
    dev = torch.device("cuda", 0)
    data = eb.batches(args, dev, a.batches)
    ref = eb.run(args, "fp32", dev, data)
    b = copy.copy(args)
    b.precision = "fp16h"
    model, stem, _, _ = Bn.build(b, dev)
    tr = Trainer(model, stem)
    model.train()
    alt = {}
    real_tail = model._gp_tail

    def tail_both(x, lay, h, w):
        out = real_tail(x, lay, h, w)
        tail = model.c1x1_tail.out_channels
        c = model.c1x1_tail.weight.shape[1]
        t = F.relu(torch.einsum("nhwc,oc->nhwo", x[..., :c].float(), model.c1x1_tail.weight.float().view(tail, c)) + model.c1x1_tail.bias.float())
        dense = torch.zeros((lay.n_frames, lay.B) + tuple(t.shape[1:]), device=x.device)
        dense.index_put_((lay.frame_of.long(), lay.sample_of.long()), t)
        pooled = dense[:, :, 1:-1, 1:-1].amax(0).permute(0, 3, 1, 2).reshape(lay.B, -1)          # NCHW-flattened like the reference
        alt["logits"] = pooled @ model.out_linear.weight.float().t() + model.out_linear.bias.float()
        return out
    model._gp_tail = tail_both
    for i, (clip, q, v_lens, q_lens) in enumerate(data):
        with torch.no_grad():
            native, v_sorted, perm = tr.extract_features(clip.to(dev), v_lens)
            model.init_hidden()
            out = model(native, q.to(dev)[perm.to(dev)], v_sorted, q_lens[perm]).float().cpu()
        r = ref[i]
        e0 = float((out - r).abs().max() / r.abs().max())
        e1 = float((alt["logits"].cpu() - r).abs().max() / r.abs().max())
        print("minibatch %d: fp16h %.3e   same trunk output, tail conv + max + out_linear in fp32 %.3e" % (i, e0, e1), flush=True)


if __name__ == "__main__":
    main()
