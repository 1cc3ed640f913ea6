#!/usr/bin/env python3
"""ON THE GPU BOX: the logits error of a precision (default 'fp16h') against precision 'fp32' (identical weights, train-mode forward at
the headline size) on N seeded minibatches (default 12: one full-length, the rest ragged): the maximum over the minibatches (the
tests' criterion), their RMS and mean, and how many answers differ.

  python tools/error_budget.py [--batches 12] [--seed S] [--data noise|smooth|blocks|textured] [--precision fp16h|fp16|bf16]
(tools/experiments/precision_budget.py has the per-rounding-point budget behind the mode's design)"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

KEYS = {"COH": "VNQA_COHERENT_ROUND"}


def batches(args, device, n, data="noise"):
    """data: 'noise' = i.i.d. uniform pixels (the benchmark's synthetic clips; half of the stem's default calibration frames);
    'smooth' = 14 x 14 noise per frame bilinearly upsampled 16 x, plus a per-clip brightness and a slow drift over the frames — large
    flat regions, other channel means (the other half of the calibration frames is of this kind, without the drift);
    'blocks' = NOT a calibration distribution: piecewise-constant images (bench.blocks_clip)."""
    import torch.nn.functional as F
    out = []
    for i in range(n):
        g = torch.Generator(device="cpu").manual_seed(777 + i)
        B, T = args.batch, args.frames
        if data == "smooth":
            low = torch.rand(B * T, 3, args.height // 16, args.width // 16, generator=g)
            up = F.interpolate(low, size=(args.height, args.width), mode="bilinear", align_corners=False).view(B, T, 3, args.height, args.width)
            gain = 0.3 + 0.7 * torch.rand(B, 1, 1, 1, 1, generator=g)
            drift = torch.linspace(0, 0.2, T).view(1, T, 1, 1, 1) * torch.rand(B, 1, 1, 1, 1, generator=g)
            clip = (up * gain + drift).clamp_(0, 1).permute(0, 2, 3, 4, 1).contiguous()
        elif data == "blocks":       # a THIRD kind, like neither half of the calibration frames: piecewise-constant images
            clip = bench.blocks_clip(B, T, args.height, args.width, g)
        elif data == "textured":     # a FOURTH: flat regions x static texture x slow illumination ramp (bench.textured_clip)
            clip = bench.textured_clip(B, T, args.height, args.width, g)
        else:
            clip = torch.rand(B, 3, args.height, args.width, T, generator=g)
        q_lens = torch.randint(5, 26, (B,), generator=g)
        q = torch.randint(1, 134, (B, 56), generator=g)
        q = q * (torch.arange(56).unsqueeze(0) < q_lens.unsqueeze(1)).long()
        v_lens = torch.full((B,), T, dtype=torch.long) if i == 0 else torch.randint(3, T + 1, (B,), generator=g)
        if i > 0:
            v_lens[0] = T
            clip = clip * (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()
        out.append((clip, q, v_lens, q_lens))
    return out


def run(args, prec, device, data):
    from videonavqa_amd.train import Trainer
    a = copy.copy(args)
    a.precision = prec
    model, stem, _, _ = bench.build(a, device)
    tr = Trainer(model, stem, lr=1e-4, clip=1.0, loss_reduction="sum")
    model.train()
    res = []
    with torch.no_grad():
        for clip, q, v_lens, q_lens in data:
            native, v_sorted, perm = tr.extract_features(clip.to(device), v_lens)
            model.init_hidden()
            out = model(native, q.to(device)[perm.to(device)], v_sorted, q_lens[perm])
            res.append(out.float().cpu())
    del tr, model, stem
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--data", default="noise", choices=["noise", "smooth", "blocks", "textured"], help="the minibatches' pixel statistics (the default "
                    "calibration frames are half uniform noise, half smooth; 'blocks' is like neither)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the random weights (0 = the benchmark's)")
    ap.add_argument("--precision", default="fp16h", help="the precision under test (fp16h, fp16, bf16)")
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224, help="--height 160 --width 208: the reference's own frames (eval/utils.py:24-25)")
    ap.add_argument("--model", default="film_attn_pt", choices=["film_attn_pt", "film_gp_pt", "time_multi_hop"])
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("settings", nargs="*", default=["COH=1"], help="one pass per argument; each word KEY=VALUE of it sets an environment variable (COH = VNQA_COHERENT_ROUND) or, "
                    "as module.ATTRIBUTE=int, a module attribute of the package for that pass")
    o = ap.parse_args()
    args = argparse.Namespace(precision="fp32", model=o.model, batch=8, frames=o.frames, height=o.height, width=o.width, blocks=1, channels=512,
                              tail_channels=0, seed=o.seed)
    from videonavqa_amd import _lib as L
    L.set_half("bf16" if o.precision == "bf16" else "f16")
    device = torch.device("cuda", 0)
    data = batches(args, device, o.batches, o.data)
    ref = run(args, "fp32", device, data)
    print("%-44s %8s %8s %8s  %s   per batch (x 1e-3)" % ("setting", "max", "rms", "mean", "flips"), flush=True)
    for s in o.settings:
        saved = {}
        attrs = {}
        for word in s.split():
            k, v = word.split("=")
            if "." in k:          # module.ATTRIBUTE=int of the package (the A/B switches that are module attributes, e.g. ops.HEAD_SPLIT_OUT=0)
                import importlib
                mod, name = k.rsplit(".", 1)
                m = importlib.import_module("videonavqa_amd." + mod)
                attrs[(m, name)] = getattr(m, name)
                setattr(m, name, type(getattr(m, name))(int(v)))
                continue
            k = KEYS.get(k, k)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        got = run(args, o.precision, device, data)
        for (m, name), v in attrs.items():
            setattr(m, name, v)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
        rel = [float((g - r).abs().max() / r.abs().max()) * 1e3 for g, r in zip(got, ref)]
        flips = sum(int((g.argmax(1) != r.argmax(1)).sum()) for g, r in zip(got, ref))
        rms = (sum(x * x for x in rel) / len(rel)) ** 0.5
        print("%-44s %8.3f %8.3f %8.3f  %d/%d   %s" % (s, max(rel), rms, sum(rel) / len(rel), flips, 8 * len(rel),
                                                    " ".join("%.2f" % x for x in rel)), flush=True)


if __name__ == "__main__":
    main()
