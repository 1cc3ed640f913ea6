"""Where does the 16-bit mode's (--low bf16 | fp16) logits error come from?  Hybrid runs at the headline size on identical weights:
  A fp32 stem + fp32 trunk (reference)      B bf16 stem + bf16 trunk (the benchmark precision)
  C fp32 stem -> bf16 trunk                 D bf16 stem -> fp32 trunk
usage (GPU box): python tools/parity_localize.py [--model film_attn_pt] > gpurun_out/parity_localize.json"""
import argparse
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as Bn                                                    # noqa: E402
from videonavqa_amd.models.common import NativeFeatures               # noqa: E402
from videonavqa_amd.train import Trainer                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="film_attn_pt")
    ap.add_argument("--frames", type=int, default=35)
    ap.add_argument("--low", default="bf16", choices=["bf16", "fp16", "fp16h"], help="the 16-bit precision under test")
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--seed", type=int, default=0, help="weight seed (bench.build)")
    ap.add_argument("--batches", type=int, default=0, help="> 0: tools/error_budget.py's seeded minibatches 0..N-1 instead of the three parity batches")
    a = ap.parse_args()
    from videonavqa_amd import _lib as L
    L.set_half("f16" if a.low in ("fp16", "fp16h") else "bf16")      # one 16-bit format per process, fixed before the fp32 build
    args = argparse.Namespace(precision=a.low, batch=8, frames=a.frames, height=a.height, width=a.width, blocks=1, channels=512,
                              model=a.model, seed=a.seed, tail_channels=0)
    dev = torch.device("cuda", 0)
    if a.batches > 0:
        import importlib.util
        spec = importlib.util.spec_from_file_location("error_budget", os.path.join(os.path.dirname(os.path.abspath(__file__)), "error_budget.py"))
        eb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(eb)
        batches = [(c.to(dev), q.to(dev), vl, ql, None) for c, q, vl, ql in eb.batches(args, dev, a.batches)]
    else:
        batches = Bn.parity_batches(args, dev)
    tr = {}
    for prec in ("fp32", a.low):
        b = copy.copy(args)
        b.precision = prec
        model, stem, _, _ = Bn.build(b, dev)
        tr[prec] = Trainer(model, stem)
        model.train()

    def logits(stem_prec, trunk_prec, batch):
        clip, q, v_lens, q_lens, y = batch
        native, v_sorted, perm = tr[stem_prec].extract_features(clip, v_lens)
        cdt = torch.float32 if trunk_prec == "fp32" else (torch.float16 if a.low in ("fp16", "fp16h") else torch.bfloat16)
        data, shift = native.data, native.shift
        if trunk_prec != stem_prec:      # another precision's trunk reads PLAIN features (split: hi + lo; mean-shifted: value + shift)
            data, shift = tr[stem_prec].stem.plain_features(data), None
        native = NativeFeatures(data.to(cdt).contiguous(), native.layout, native.channels, native.h, native.w, shift=shift)
        m = tr[trunk_prec].model
        m.init_hidden()
        with torch.no_grad():
            return m(native, q[perm.to(dev)], v_sorted, q_lens[perm]).float(), native.data.float()

    out = {}
    for bi, batch in enumerate(batches):
        ref, fref = logits("fp32", "fp32", batch)
        row = {}
        for name, (sp, tp) in dict(B=(a.low, a.low), C=("fp32", a.low), D=(a.low, "fp32")).items():
            lg, f = logits(sp, tp, batch)
            f = f[..., :512]
            row[name] = round(float((lg - ref).abs().max() / ref.abs().max()), 6)
            if name == "D":
                row["stem_feat_rel_l2"] = round(float((f - fref).norm() / fref.norm()), 6)
                row["stem_feat_max_rel"] = round(float((f - fref).abs().max() / fref.abs().max()), 6)
        out["batch%d" % bi] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
