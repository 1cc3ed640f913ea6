#!/usr/bin/env python3
"""ON THE GPU BOX — experiment, not product: how much of the tolerance mode's remaining logits error is ACTIVATION rounding in
the fp16 stem, and would noise-shaped rounding (error diffusion inside small spatial blocks: the block's SUM keeps one rounding
error instead of the blocks' worth) reduce it?  The stem is simulated in torch fp32 (exact weights) with a rounding hook after every
tensor the fp16 stem stores; its features go through the library's exact-f32 FiLM-attn model (train-mode forward).  Reference =
the same simulated stem without rounding, so conv-implementation differences cancel.

  python tools/experiments/act_rounding_sim.py [--batches 6]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

BN_EPS = 1e-5


def rtn(t):
    return t.half().float()


def diffuse(t, bs):
    """fp16 rounding with the error carried along a serpentine path inside every bs x bs spatial block (per image and channel)."""
    N, C, H, W = t.shape
    if H % bs or W % bs:
        return rtn(t)
    x = t.view(N, C, H // bs, bs, W // bs, bs).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // bs, W // bs, bs * bs).clone()
    order = []
    for r in range(bs):
        cols = range(bs) if r % 2 == 0 else range(bs - 1, -1, -1)
        order += [r * bs + c for c in cols]
    carry = torch.zeros_like(x[..., 0])
    out = torch.empty_like(x)
    for k in order:
        v = x[..., k] + carry
        q = v.half().float()
        # post-ReLU tensors stay non-negative; a carried error must not turn an exact zero (ReLU-dead or pooled-away) into a value
        q = torch.where(x[..., k] == 0, torch.zeros_like(q), q)
        carry = torch.where(x[..., k] == 0, carry, v - q)
        out[..., k] = q
    return out.view(N, C, H // bs, W // bs, bs, bs).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H, W)


@torch.no_grad()
def sim_stem(vgg, od, frames, rnd, chunk=40):
    f = vgg.features
    conv = lambda t, c: F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
    bn = lambda t, b: F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, BN_EPS)
    outs = []
    for i in range(0, frames.shape[0], chunk):
        x = rnd(frames[i:i + chunk])
        a = rnd(F.relu(conv(x, f["0"])))
        a = rnd(F.max_pool2d(F.relu(conv(a, f["2"])), 2))
        a = rnd(F.relu(conv(a, f["5"])))
        a = rnd(bn(rnd(F.max_pool2d(F.relu(conv(a, f["7"])), 2)), od.bn_input))
        a = conv(a, od.conv11)                                   # (composed pair: the intermediate is never stored)
        a = rnd(F.max_pool2d(F.relu(bn(conv(a, od.conv12), od.bn1)), 2))
        a = rnd(conv(a, od.conv21))
        a = rnd(F.max_pool2d(F.relu(bn(conv(a, od.conv22), od.bn2)), 2))
        a = rnd(conv(a, od.conv31))
        outs.append(F.relu(bn(conv(a, od.conv32), od.bn3)))       # conv32: fp32 output in the tolerance mode
    return torch.cat(outs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=6)
    o = ap.parse_args()
    from videonavqa_amd.models.common import FrameLayout, NativeFeatures
    from videonavqa_amd.train import Trainer
    args = argparse.Namespace(precision="fp32", model="film_attn_pt", batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                              tail_channels=0, seed=0)
    dev = torch.device("cuda", 0)
    model, stem, vgg, od = bench.build(args, dev)
    tr = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum")
    model.train()
    spec = importlib_budget()
    data = spec.batches(args, dev, o.batches)
    variants = [("none", lambda t: t), ("round-to-nearest", rtn), ("diffuse 2x2", lambda t: diffuse(t, 2)), ("diffuse 4x4", lambda t: diffuse(t, 4))]
    res = {k: [] for k, _ in variants}
    with torch.no_grad():
        for clip, q, v_lens, q_lens in data:
            clip = clip.to(dev)
            B, _, H, W, T = clip.shape
            v_sorted, perm = torch.sort(v_lens, dim=0, descending=True, stable=True)
            lay = FrameLayout(v_sorted, T, dev, perm=perm)
            img_of = lay.img_of.long().cpu()
            frames = torch.empty(lay.n_img, 3, H, W, device=dev)
            for bt in range(B * T):
                if int(img_of[bt]) >= 0:
                    frames[int(img_of[bt])] = clip[bt // T, :, :, :, bt % T]
            for name, rnd in variants:
                feat = sim_stem(vgg, od, frames, rnd)                             # [n_img, 512, 14, 14]
                data_nhwc = torch.zeros(lay.n_img, 16, 16, 512, device=dev)
                data_nhwc[:, 1:-1, 1:-1, :] = feat.permute(0, 2, 3, 1)
                native = NativeFeatures(data_nhwc, lay, 512, H // 16, W // 16)
                model.init_hidden()
                out = model(native, q.to(dev)[perm.to(dev)], v_sorted, q_lens[perm])
                res[name].append(out.float().cpu())
    ref = res["none"]
    for name, _ in variants[1:]:
        rel = [float((g - r).abs().max() / r.abs().max()) * 1e3 for g, r in zip(res[name], ref)]
        print("%-18s max %.3f rms %.3f   %s" % (name, max(rel), (sum(x * x for x in rel) / len(rel)) ** 0.5, " ".join("%.2f" % x for x in rel)), flush=True)


def importlib_budget():
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("x3_error_budget", os.path.join(root, "x3_error_budget.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == "__main__":
    main()
