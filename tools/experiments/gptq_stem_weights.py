#!/usr/bin/env python3
"""ON THE GPU BOX — experiment: second-order rounding of the FROZEN stem's 16-bit weights.

stem.coherent_round picks, per output channel, the few roundings that cancel the weight-rounding error against the MEAN input
(first moment).  This experiment rounds against the whole second-moment matrix H = E[p p^T] of the layer's input patches p
(K = c_in * taps) — the GPTQ / OBQ sequential rounding: column by column in order of decreasing H_jj, the rounding error of
column j is pushed onto the not-yet-rounded columns along H^-1, so that E[(p . dw)^2] = dw^T H dw is minimised greedily — and
prices it with tools/experiments/precision_budget.py's restatement: squared logits error of the weight roundings alone, per
layer group, calibrated on one data kind and evaluated on noise AND on smooth clips.

  python tools/experiments/gptq_stem_weights.py [--seeds 0 1] [--batches 6] [--calib noise|smooth|mix] [--calib-frames 40]
"""
import argparse
import importlib.util
import os
import sys
import time

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


PB = load("precision_budget", os.path.join(HERE, "precision_budget.py"))
BN_EPS = 1e-5


@torch.no_grad()
def layer_inputs(vgg, od, frames):
    """The exact-f32 input of every stem layer (key -> [N, C, H, W]) and the conv padding it is read with."""
    f = vgg.features
    bn = lambda t, b: F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, BN_EPS)
    conv = lambda t, c: F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
    x = frames
    d = {"sw_c11": x}
    a = F.relu(conv(x, f["0"])); d["sw_c12"] = a
    a = F.max_pool2d(F.relu(conv(a, f["2"])), 2); d["sw_c21"] = a
    a = F.relu(conv(a, f["5"])); d["sw_c22"] = a
    a = bn(F.max_pool2d(F.relu(conv(a, f["7"])), 2), od.bn_input); d["sw_comp"] = a
    a = F.max_pool2d(F.relu(bn(conv(conv(a, od.conv11), od.conv12), od.bn1)), 2); d["sw_od21"] = a
    a1 = conv(a, od.conv21); d["sw_od22"] = a1
    a = F.max_pool2d(F.relu(bn(conv(a1, od.conv22), od.bn2)), 2); d["sw_od31"] = a
    a1 = conv(a, od.conv31); d["sw_od32"] = a1
    return d


@torch.no_grad()
def second_moment(x, k, pad, max_rows=400000):
    """H = sum p p^T over the k x k patches of x [N, C, H, W] (zero padding `pad`), float64 [K, K]; positions subsampled to max_rows."""
    N, C, Hh, Ww = x.shape
    K = C * k * k
    H = torch.zeros(K, K, dtype=torch.float64, device=x.device)
    rows = 0
    per = Hh * Ww
    stride = max(1, (N * per + max_rows - 1) // max_rows)
    g = torch.Generator(device="cpu").manual_seed(5)
    for n in range(N):
        p = F.unfold(x[n:n + 1], k, padding=pad)[0].t()            # [positions, K]  (channel-major, taps minor: w.reshape(co, -1)'s order)
        if stride > 1:
            idx = torch.randperm(per, generator=g)[:per // stride].to(x.device)
            p = p[idx]
        p = p.double()
        H += p.t() @ p
        rows += p.shape[0]
    return H / rows


@torch.no_grad()
def gptq_round(w, H, damp=0.01, block=128):
    """w [co, ci, kh, kw] fp32 -> values exactly representable in fp16, rounded sequentially against H (float64 [K, K])."""
    co = w.shape[0]
    W = w.detach().reshape(co, -1).double().clone()
    K = W.shape[1]
    H = H.clone()
    dead = torch.diag(H) == 0
    H[dead, dead] = 1.0
    W[:, dead] = W[:, dead]
    perm = torch.argsort(torch.diag(H), descending=True)
    W = W[:, perm]
    H = H[perm][:, perm]
    H += torch.eye(K, dtype=H.dtype, device=H.device) * damp * torch.mean(torch.diag(H))
    L = torch.linalg.cholesky(H)
    Hinv = torch.cholesky_inverse(L)
    U = torch.linalg.cholesky(Hinv, upper=True)
    Q = torch.zeros_like(W)
    for b0 in range(0, K, block):
        b1 = min(b0 + block, K)
        Wb = W[:, b0:b1].clone()
        Eb = torch.zeros_like(Wb)
        Ub = U[b0:b1, b0:b1]
        for j in range(b1 - b0):
            wj = Wb[:, j]
            q = wj.float().half().double()
            Q[:, b0 + j] = q
            e = (wj - q) / Ub[j, j]
            Wb[:, j:] -= e.unsqueeze(1) * Ub[j, j:].unsqueeze(0)
            Eb[:, j] = e
        W[:, b1:] -= Eb @ U[b0:b1, b1:]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(K, device=perm.device)
    return Q[:, inv].float().view_as(w)


def folded_weights(vgg, od):
    """key -> (exact fp32 weight the 16-bit stem rounds, kernel size, padding): BatchNorm scales folded, the pair composed."""
    from videonavqa_amd.stem import _fold_bn
    f = vgg.features
    dev = f["0"].weight.device
    s1, _ = _fold_bn(od.bn1)
    s2, _ = _fold_bn(od.bn2)
    s3, _ = _fold_bn(od.bn3)
    w1, w2 = od.conv11.weight.double().cpu(), od.conv12.weight.double().cpu() * s1.double().cpu().view(-1, 1, 1, 1)
    wc = F.conv2d(w1.permute(1, 0, 2, 3), w2.flip(2, 3), padding=2).permute(1, 0, 2, 3).float().to(dev).contiguous()
    return {"sw_c11": (f["0"].weight.float(), 3, 1), "sw_c12": (f["2"].weight.float(), 3, 1), "sw_c21": (f["5"].weight.float(), 3, 1),
            "sw_c22": (f["7"].weight.float(), 3, 1), "sw_comp": (wc, 5, 2), "sw_od21": (od.conv21.weight.float(), 3, 1),
            "sw_od22": (od.conv22.weight.float() * s2.view(-1, 1, 1, 1), 3, 1), "sw_od31": (od.conv31.weight.float(), 3, 1),
            "sw_od32": (od.conv32.weight.float() * s3.view(-1, 1, 1, 1), 3, 1)}


def calib_frames(kind, n, seed=4242):
    g = torch.Generator().manual_seed(seed)
    noise = torch.rand(n, 3, 224, 224, generator=g)
    low = torch.rand(n, 3, 14, 14, generator=g)
    smooth = (F.interpolate(low, size=(224, 224), mode="bilinear", align_corners=False) * (0.3 + 0.7 * torch.rand(n, 1, 1, 1, generator=g))).clamp_(0, 1)
    if kind == "noise":
        return noise
    if kind == "smooth":
        return smooth
    return torch.cat([noise[:n // 2], smooth[:n - n // 2]])


def blocks_batches(bm, args, dev, n):
    """A THIRD kind of clip, used for evaluation only: piecewise-constant images — 24 random axis-aligned rectangles of random colour
    over a random background per frame, drifting slowly over the frames — on the 'noise' batches' questions and lengths."""
    out = []
    for i, (clip, q, v_lens, q_lens) in enumerate(bm.batches(args, dev, n, "noise")):
        g = torch.Generator().manual_seed(9000 + i)
        B, _, H, W, T = clip.shape
        img = torch.rand(B, 3, 1, 1, 1, generator=g).expand(B, 3, H, W, T).clone()
        for _ in range(24):
            y0, x0 = torch.randint(0, H - 8, (B,), generator=g), torch.randint(0, W - 8, (B,), generator=g)
            hh, ww = torch.randint(8, H // 2, (B,), generator=g), torch.randint(8, W // 2, (B,), generator=g)
            col = torch.rand(B, 3, generator=g)
            dx = torch.randint(-2, 3, (B,), generator=g)
            for b in range(B):
                for t in range(T):
                    xs = int(min(max(x0[b] + dx[b] * t, 0), W - 8))
                    img[b, :, int(y0[b]):int(y0[b] + hh[b]), xs:xs + int(ww[b]), t] = col[b].view(3, 1, 1)
        mask = (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()
        out.append((img * mask, q, v_lens, q_lens))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--calib", default="noise", choices=["noise", "smooth", "mix"])
    ap.add_argument("--calib-frames", type=int, default=40)
    ap.add_argument("--layers", nargs="*", default=["sw_od31", "sw_od32", "sw_od21", "sw_od22", "sw_comp", "sw_c22", "sw_c21", "sw_c12", "sw_c11"])
    o = ap.parse_args()
    from videonavqa_amd import _lib as L
    from videonavqa_amd.stem import calibration_means
    L.set_half("f16")
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    bm = PB.budget_mod()
    groups = [("conv31", ["sw_od31"]), ("conv32", ["sw_od32"]), ("conv21 + conv22 + composed", ["sw_od21", "sw_od22", "sw_comp"]),
              ("conv1_1 .. conv2_2", ["sw_c11", "sw_c12", "sw_c21", "sw_c22"]), ("all nine", list(PB.STEM_W))]
    res = {}
    t0 = time.time()
    for seed in o.seeds:
        args = argparse.Namespace(precision="fp32", model="film_attn_pt", batch=8, frames=35, height=224, width=224, blocks=1, channels=512,
                                  tail_channels=0, seed=seed)
        model, stem, vgg, od = bench.build(args, dev)
        W = {k: v.detach().float() for k, v in model.state_dict().items()}
        W.update({k: v.detach().float() for k, v in model.extra_state_tensors().items()})
        del model, stem
        exact = PB.Setting("exact")
        fw = folded_weights(vgg, od)
        d_coh = PB.stem_weight_deltas(vgg, od, calibration_means(vgg, od))
        ins = layer_inputs(vgg, od, calib_frames(o.calib, o.calib_frames).to(dev))
        d_gptq, d_rtn = {}, {}
        for key in PB.STEM_W:
            w, k, pad = fw[key]
            d_rtn[key] = w.half().float() - w
            if key in o.layers:
                H = second_moment(ins[key], k, pad)
                q = gptq_round(w, H)
                d_gptq[key] = q - w
                # layer-level check on the calibration patches: dw^T H dw summed over output channels
                e = lambda d: float(((d.reshape(d.shape[0], -1).double() @ H) * d.reshape(d.shape[0], -1).double()).sum())
                print("seed %d %-8s K %5d  E|p.dw|^2: rtn %.3e  coherent %.3e  gptq %.3e   (%.0f s)" %
                      (seed, key, H.shape[0], e(d_rtn[key]), e(d_coh[key]), e(d_gptq[key]), time.time() - t0), flush=True)
                del H
            else:
                d_gptq[key] = d_coh[key]
        del ins
        torch.cuda.empty_cache()
        for data_kind in ("noise", "smooth", "blocks"):
            data = blocks_batches(bm, args, dev, o.batches) if data_kind == "blocks" else bm.batches(args, dev, o.batches, data_kind)
            for bi, (clip, q, v_lens, q_lens) in enumerate(data):
                frames, cts, v_sorted, perm = PB.pack_frames(clip.to(dev), v_lens)
                film = PB.question_film(W, q.to(dev)[perm.to(dev)], q_lens[perm], cts)
                wq = PB.trunk_weights(W, exact, None)
                ref = PB.sim_trunk(W, wq, PB.sim_stem(vgg, od, frames, exact), cts, film, 8, 35, exact)
                for gname, keys in groups:
                    st = PB.Setting(gname, acts=keys)
                    for method, wd in (("rtn", d_rtn), ("coherent", d_coh), ("gptq", d_gptq)):
                        out = PB.sim_trunk(W, wq, PB.sim_stem(vgg, od, frames, st, wd=wd), cts, film, 8, 35, exact)
                        err = float((out - ref).abs().max() / ref.abs().max()) * 1e3
                        res.setdefault((data_kind, gname, method), []).append(err)
            print("seed %d %s data done (%.0f s)" % (seed, data_kind, time.time() - t0), flush=True)
    print("\ncalibrated on %d %s frames; squared logits error x 1e-6 (mean over %d minibatches x %d weight seeds) [max x 1e-3]" %
          (o.calib_frames, o.calib, o.batches, len(o.seeds)))
    for data_kind in ("noise", "smooth", "blocks"):
        print("-- evaluated on %s clips" % data_kind)
        for gname, _ in groups:
            row = []
            for method in ("rtn", "coherent", "gptq"):
                e = res[(data_kind, gname, method)]
                row.append("%s %.4f [%.3f]" % (method, sum(x * x for x in e) / len(e), max(e)))
            print("   %-30s %s" % (gname, "   ".join(row)))


if __name__ == "__main__":
    main()
