#!/usr/bin/env python3
"""ON THE GPU BOX — diagnostic: the 16-bit stem's error LAYER BY LAYER against a torch fp32 chain on the same frames and weights, next
to what storage rounding alone predicts (the same chain with every stored tensor rounded to fp16 — tools/experiments/
precision_budget.py's model of the stem).  A layer whose measured error exceeds the prediction carries an error the rounding model
does not know: a kernel defect or an unmodelled rounding (round 6: the 160 x 208 geometry read 1.45e-3 where the model said 0.83e-3).

  python tools/stem_layer_errors.py [--height 160 --width 208] [--seed 3] [--batch-index 2] [--precision fp16|fp16h] [--data noise]"""
import argparse
import importlib.util
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@torch.no_grad()
def chain(vgg, od, frames, rnd):
    """The stem as torch fp32 ops; rnd(t) is applied wherever the 16-bit stem STORES a tensor.  Returns {layer: NCHW tensor}."""
    f = vgg.features
    conv = lambda t, c: F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
    bn = lambda t, b: F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, 1e-5)
    out = {}
    x = rnd(frames)
    a = rnd(F.relu(conv(x, f["0"])))
    a = out["conv1 (fused conv1_1 + conv1_2, pooled)"] = rnd(F.max_pool2d(F.relu(conv(a, f["2"])), 2))
    a = out["conv2_1"] = rnd(F.relu(conv(a, f["5"])))
    a = out["conv2_2 (pooled, bn_input)"] = rnd(bn(F.max_pool2d(F.relu(conv(a, f["7"])), 2), od.bn_input))
    a = out["conv11.conv12 (composed, pooled)"] = rnd(F.max_pool2d(F.relu(bn(conv(conv(a, od.conv11), od.conv12), od.bn1)), 2))
    a = out["conv21"] = rnd(conv(a, od.conv21))
    a = out["conv22 (pooled)"] = rnd(F.max_pool2d(F.relu(bn(conv(a, od.conv22), od.bn2)), 2))
    a = out["conv31"] = rnd(conv(a, od.conv31))
    out["conv32 (features)"] = rnd(F.relu(bn(conv(a, od.conv32), od.bn3)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=160)
    ap.add_argument("--width", type=int, default=208)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--batch-index", type=int, default=2)
    ap.add_argument("--data", default="noise")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp16h"])
    ap.add_argument("--calibration", default="auto", help="auto (second-order rounded weights) | none (round to nearest)")
    o = ap.parse_args()
    from videonavqa_amd import _lib as L
    L.set_half("f16")
    from videonavqa_amd.stem import FrozenStem
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    eb = load(os.path.join(ROOT, "tools", "error_budget.py"), "error_budget")
    pb = load(os.path.join(ROOT, "tools", "experiments", "precision_budget.py"), "precision_budget")
    args = argparse.Namespace(precision=o.precision, model="film_attn_pt", batch=8, frames=35, height=o.height, width=o.width, blocks=1,
                              channels=512, tail_channels=0, seed=o.seed)
    clip, q, v_lens, q_lens = eb.batches(args, dev, o.batch_index + 1, o.data)[o.batch_index]
    _, _, vgg, od = bench.build(args, dev)
    stem = FrozenStem(vgg, od, o.precision, calibration=None if o.calibration == "none" else "auto", split_features=False)
    frames, cts, v_sorted, perm = pb.pack_frames(clip.to(dev), v_lens)
    from videonavqa_amd.models.common import FrameLayout
    lay = FrameLayout(v_sorted, 35, dev, perm=perm)
    stem._tap = {}
    feat = stem.forward_clip(clip.to(dev), lay.img_of, lay.n_img).clone()
    H, W = o.height, o.width

    def nchw(t, c, halo=1, segs=1):
        t = t[:lay.n_img].float()
        v = t[..., :c] + (t[..., c:2 * c] if segs > 1 else 0)
        return v[:, halo:v.shape[1] - halo, halo:v.shape[2] - halo].permute(0, 3, 1, 2)
    segs = lambda t, c: 2 if t.shape[-1] >= 2 * c else 1
    # (mean-shifted storage: the stored tensor is v - mu_c; add the shift back)
    sh = lambda key, c: (stem.shift[key][:c].view(1, -1, 1, 1) if key in getattr(stem, "shift", {}) else 0.0)
    got = {"conv1 (fused conv1_1 + conv1_2, pooled)": nchw(stem._bufs[("vgg", 0, H // 2, W // 2)], 64) + sh("c12", 64),
           "conv2_1": nchw(stem._tap[("vgg", 1)], 128) + sh("c21", 128),
           "conv2_2 (pooled, bn_input)": nchw(stem._tap[("vgg", 2)], 128, halo=2 if stem.composed is not None else 1) + sh("c22", 128)}
    t = stem._tap[("od", "c")]
    got["conv11.conv12 (composed, pooled)"] = nchw(t, 512, segs=segs(t, 512)) + sh("comp", 512)
    for i, name, key in ((2, "conv21", "od21"), (3, "conv22 (pooled)", "od22"), (4, "conv31", "od31")):
        t = stem._tap[("od", i)]
        got[name] = nchw(t, 512, segs=segs(t, 512)) + sh(key, 512)
    got["conv32 (features)"] = nchw(feat, 512)
    exact = chain(vgg, od, frames, lambda t: t)
    model = chain(vgg, od, frames, lambda t: t.half().float())          # (un-shifted fp16 storage: what rounds 3-5 stored)
    print("%dx%d, weight seed %d, minibatch %d (%d images), precision %s, calibration %s" % (H, W, o.seed, o.batch_index, lay.n_img, o.precision, o.calibration))
    print("%-44s %12s %12s %8s   %s" % ("layer output (rel. L2 error vs torch fp32)", "library", "rounding model", "ratio", "max abs err / max |v|: library, model"))
    for k in exact:
        e, g, m = exact[k], got[k], model[k]
        lib = float((g - e).norm() / e.norm())
        mod = float((m - e).norm() / e.norm())
        print("%-44s %12.3e %12.3e %8.2f   %.3e %.3e" % (k, lib, mod, lib / mod, float((g - e).abs().max() / e.abs().max()),
                                                       float((m - e).abs().max() / e.abs().max())), flush=True)
        # where: border ring vs interior
        b = torch.zeros_like(e[0, 0], dtype=torch.bool)
        b[0, :] = b[-1, :] = b[:, 0] = b[:, -1] = True
        eb_, ei_ = (g - e)[:, :, b], (g - e)[:, :, ~b]
        mb_, mi_ = (m - e)[:, :, b], (m - e)[:, :, ~b]
        print("    border pixels rms err: library %.3e model %.3e | interior: library %.3e model %.3e" %
              (float(eb_.pow(2).mean().sqrt()), float(mb_.pow(2).mean().sqrt()), float(ei_.pow(2).mean().sqrt()), float(mi_.pow(2).mean().sqrt())))


if __name__ == "__main__":
    main()
