#!/usr/bin/env python3
"""ON THE GPU BOX — experiment: can the rounding of a TRAINABLE layer's weights (conv_init: its input, the frozen stem's features, has
a FIXED second moment H) be done per step?  The full second-order recursion costs 9 ms per 4608 x 512 layer; this prices a rank-k
form: H ~ V V^T + diag(d) (top-k eigenpairs + the residual diagonal), columns rounded sequentially to the neighbour that minimises
|a + e V_j|^2 + d_j e^2 with a = V^T dw so far (greedy discrepancy minimisation in k dimensions: K k work per row instead of K^2).
Prints the spectrum of H_feat and E|p . dw|^2 = dw^T H dw of round-to-nearest / mean-coherent / rank-k / full second order, with H
from noise frames and evaluated on H from noise AND from smooth frames."""
import argparse
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
import bench  # noqa: E402


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


G = load("gptq_stem_weights", os.path.join(HERE, "..", "gptq_stem_weights.py"))


@torch.no_grad()
def rank_k_round(w, V, d, order):
    """w [co, K] float64, V [K, k] (eigvec * sqrt(eigval)), d [K] residual diagonal; returns fp16-grid values [co, K]."""
    co, K = w.shape
    lo = w.float().half()
    lo_f = lo.double()
    # the other neighbour of each value on the fp16 grid
    up = lo_f < w
    bits = lo.view(torch.int16).to(torch.int32)
    step = torch.where((lo_f > 0) == up, torch.ones_like(bits), -torch.ones_like(bits))
    step = torch.where(lo_f == 0, torch.where(up, torch.ones_like(bits), torch.full_like(bits, -32767)), step)
    other = torch.where(lo_f == 0, step, bits + step).to(torch.int16).view(torch.float16).double()
    e0, e1 = lo_f - w, other - w
    q = lo_f.clone()
    a = torch.zeros(co, V.shape[1], dtype=torch.float64, device=w.device)
    for j in order.tolist():
        vj = V[j]                                        # [k]
        c0 = ((a + e0[:, j:j + 1] * vj) ** 2).sum(1) + d[j] * e0[:, j] ** 2
        c1 = ((a + e1[:, j:j + 1] * vj) ** 2).sum(1) + d[j] * e1[:, j] ** 2
        pick = c1 < c0
        e = torch.where(pick, e1[:, j], e0[:, j])
        q[:, j] = torch.where(pick, other[:, j], lo_f[:, j])
        a += e.unsqueeze(1) * vj
    return q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ranks", type=int, nargs="*", default=[4, 16, 32, 64])
    ap.add_argument("--batches", type=int, default=6)
    o = ap.parse_args()
    from videonavqa_amd import _lib as L
    from videonavqa_amd.stem import coherent_round, second_order_round
    L.set_half("f16")
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(precision="fp32", model="film_attn_pt", batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                              tail_channels=0, seed=o.seed)
    model, stem, vgg, od = bench.build(args, dev)
    w = model.conv_init.weight.detach().float()                       # [512, 512, 3, 3]
    H = {}
    for kind in ("noise", "smooth"):
        frames = G.calib_frames(kind, 40, seed=4242 if kind == "noise" else 99).to(dev)
        ins = G.layer_inputs(vgg, od, frames)
        a1 = ins["sw_od32"]
        from videonavqa_amd.stem import BN_EPS
        import torch.nn.functional as F
        feat = F.relu(F.batch_norm(F.conv2d(a1, od.conv32.weight.float(), od.conv32.bias.float(), padding=1), od.bn3.running_mean.float(),
                                   od.bn3.running_var.float(), od.bn3.weight.float(), od.bn3.bias.float(), False, 0.0, BN_EPS))
        H[kind] = G.second_moment(feat, 3, 1)
        H[kind + "_mean"] = feat.mean((0, 2, 3))
        del ins
    Hn = H["noise"]
    ev, evec = torch.linalg.eigh(Hn)
    ev, evec = ev.flip(0), evec.flip(1)
    tr = float(ev.sum())
    print("H_feat (noise frames): K = %d, trace %.4g; energy in the top k eigenvalues: %s" %
          (Hn.shape[0], tr, "  ".join("%d: %.4f" % (k, float(ev[:k].sum()) / tr) for k in (1, 4, 16, 64, 256, 1024))))
    W = w.reshape(512, -1).double()
    err = lambda q, Hm: float((((q - W) @ Hm) * (q - W)).sum())
    rows = [("round to nearest", W.float().half().double())]
    rows.append(("coherent (mean)", coherent_round(w, H["noise_mean"], torch.float16).reshape(512, -1).double()))
    for k in o.ranks:
        V = evec[:, :k] * ev[:k].clamp_min(0).sqrt()
        dres = (torch.diagonal(Hn) - (V ** 2).sum(1)).clamp_min(0)
        order = torch.argsort(torch.diagonal(Hn), descending=True)
        rows.append(("rank-%d greedy" % k, rank_k_round(W, V, dres, order)))
    rows.append(("full second order", second_order_round(w, Hn, torch.float16).reshape(512, -1).double()))
    print("%-22s %14s %14s" % ("conv_init weights", "H noise", "H smooth"))
    for name, q in rows:
        print("%-22s %14.4e %14.4e" % (name, err(q, H["noise"]), err(q, H["smooth"])))
    # ... and what that is in LOGITS (precision_budget's restatement, everything else exact): squared error x 1e-6 of the conv_init
    # weight rounding alone, on noise and on smooth minibatches
    PB = G.PB
    bm = PB.budget_mod()
    Wd = {k: v.detach().float() for k, v in model.state_dict().items()}
    Wd.update({k: v.detach().float() for k, v in model.extra_state_tensors().items()})
    exact = PB.Setting("exact")
    for data_kind in ("noise", "smooth"):
        data = bm.batches(args, dev, o.batches, data_kind)
        acc = {name: [] for name, _ in rows}
        for clip, q, v_lens, q_lens in data:
            frames, cts, v_sorted, perm = PB.pack_frames(clip.to(dev), v_lens)
            film = PB.question_film(Wd, q.to(dev)[perm.to(dev)], q_lens[perm], cts)
            feat = PB.sim_stem(vgg, od, frames, exact)
            wq = PB.trunk_weights(Wd, exact, None)
            ref = PB.sim_trunk(Wd, wq, feat, cts, film, 8, 35, exact)
            for name, qw in rows:
                wq2 = dict(wq)
                wq2["w_init"] = qw.float().view_as(w)
                out = PB.sim_trunk(Wd, wq2, feat, cts, film, 8, 35, exact)
                acc[name].append(float((out - ref).abs().max() / ref.abs().max()) * 1e3)
        print("-- logits, %s minibatches (squared x 1e-6 [max x 1e-3])" % data_kind)
        for name, _ in rows:
            e = acc[name]
            print("   %-22s %.4f [%.3f]" % (name, sum(x * x for x in e) / len(e), max(e)))


if __name__ == "__main__":
    main()
