#!/usr/bin/env python3
"""How many of the pooling heads' arg-max frame flips (16-bit run vs exact-f32 run, bench.py `pooling_head`) come from the
LAST rounding alone — the storage rounding of relu(c1x1_tail) right before the max over frames (film_global_pooling_pt_stem.py:
228-236) — as opposed to the error accumulated upstream of it?  Runs the exact-f32 precision on one benchmark minibatch,
captures the tail maps, rounds them to bf16 / fp16 and recomputes each (sample, feature)'s arg-max frame.
    python tools/diag_gp_flip.py [--model film_gp_pt|time_multi_hop] [--frames 35]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench as Bn
    from videonavqa_amd import ops
    from videonavqa_amd.train import Trainer
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="film_gp_pt")
    ap.add_argument("--frames", type=int, default=35)
    a = ap.parse_args()
    args = argparse.Namespace(precision="fp32", batch=8, frames=a.frames, height=224, width=224, blocks=1, channels=512,
                              model=a.model, tail_channels=0)
    dev = torch.device("cuda", 0)
    model, stem, _, _ = Bn.build(args, dev)
    tr = Trainer(model, stem)
    clip, q, v_lens, q_lens, y = Bn.parity_batches(args, dev)[1]        # a ragged minibatch
    captured = {}
    real = ops.frame_max

    def spy(t, lay, tail, gs=1.0, route=None):
        captured["t"], captured["lay"], captured["tail"] = t.detach().clone(), lay, tail
        return real(t, lay, tail, gs, route)
    ops.frame_max = spy
    native, v_sorted, perm = tr.extract_features(clip, v_lens)
    model.train()
    model.init_hidden()
    with torch.no_grad():
        model(native, q[perm.to(dev)], v_sorted, q_lens[perm])
    t, lay, tail = captured["t"].float(), captured["lay"], captured["tail"]
    n_img, hp, wp, tp = t.shape
    dense = torch.full((lay.n_frames, lay.B, hp, wp, tp), -1.0, device=dev)
    dense.index_put_((lay.frame_of, lay.sample_of), t)
    dense = dense[:, :, 1:-1, 1:-1, :tail]

    def argmax_of(d):
        m, am = d.max(0)
        return torch.where(m > 0, am, torch.full_like(am, -1))
    ref = argmax_of(dense)
    live = ref >= 0
    out = {"model": a.model, "frames": a.frames, "live_features": int(live.sum())}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        am = argmax_of(dense.to(dt).float())
        out["flip_frac_from_last_rounding_" + name] = round(float(((am != ref) & live).sum()) / float(live.sum()), 5)
    # how close the top two frames are: relative gap quantiles
    top2 = dense.topk(2, 0)[0]
    gap = ((top2[0] - top2[1]) / top2[0].clamp_min(1e-20))[live]
    out["top2_rel_gap_quantiles_1_5_25_50"] = [round(float(torch.quantile(gap, p)), 5) for p in (0.01, 0.05, 0.25, 0.5)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
