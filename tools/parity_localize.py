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
    ap.add_argument("--low", default="bf16", choices=["bf16", "fp16"], help="the 16-bit storage format under test")
    a = ap.parse_args()
    from videonavqa_amd import _lib as L
    L.set_half("f16" if a.low == "fp16" else "bf16")      # one 16-bit format per process, fixed before the fp32 build
    args = argparse.Namespace(precision=a.low, batch=8, frames=a.frames, height=224, width=224, blocks=1, channels=512,
                              model=a.model)
    dev = torch.device("cuda", 0)
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
        cdt = torch.float32 if trunk_prec == "fp32" else (torch.float16 if a.low == "fp16" else torch.bfloat16)
        native = NativeFeatures(native.data.to(cdt), native.layout, native.channels, native.h, native.w)
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
            row[name] = round(float((lg - ref).abs().max() / ref.abs().max()), 6)
            if name == "D":
                row["stem_feat_rel_l2"] = round(float((f - fref).norm() / fref.norm()), 6)
                row["stem_feat_max_rel"] = round(float((f - fref).abs().max() / fref.abs().max()), 6)
        out["batch%d" % bi] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
