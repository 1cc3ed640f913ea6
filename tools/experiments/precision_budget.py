#!/usr/bin/env python3
"""ON THE GPU BOX — experiment, not product: the per-rounding-point budget of the 16-bit logits error at the headline size.

The whole path (frozen stem + FiLM-attn train-mode forward) is restated in plain torch fp32 ON THE DEVICE with a hook at every
tensor a 16-bit precision of the library stores or feeds to an MFMA (activations) and at every weight tensor it rounds.  One
exact pass per minibatch, then one pass per SETTING (a set of active rounding points); reported per setting: squared-rms / rms / max
of max|d logit| / max|logit| over the minibatches, pooled over weight seeds.  Rounding errors are small and add in variance, so
single-point settings give the budget and combined settings check it.

The exact pass is cross-checked against the library's exact-f32 precision (same weights), and the all-points setting against the
library's own fp16 precision, so the restatement is known to model the product.

  python tools/experiments/precision_budget.py [--seeds 0 1 2 3] [--batches 12] [--data noise|smooth|blocks|textured] [--only-combos]
      [--height 160 --width 208] [--per-batch] [--no-product]
      [--round6]   the round-6 settings: what is left of fp16h's error point by point on another kind of clip, candidate lo formats
                   (e4m3 / MX-fp4), dithered rounding, and MEAN-SHIFTED storage ('shift_own' / 'shift_cal') — the measurement behind
                   stem.FrozenStem._setup_mean_shift (profiles/r06_precision_budget_*.txt)
NOTE: the restatement models UN-shifted fp16 storage (rounds 3-5); its cross-check against the library's precisions ("library precision
... vs restatement") therefore shows the library BELOW the all-points setting since round 6.
"""
import argparse
import importlib.util
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

BN_EPS = 1e-5
NEG_MASK = float(-(1 << 31))

STEM_ACTS = ["a_clip", "a_c11", "a_c12", "a_c21", "a_c22", "a_comp", "a_od21", "a_od22", "a_od31", "a_feat"]
TRUNK_ACTS = ["a_init", "a_bn", "a_res", "a_z", "a_out"]
TRUNK_W = ["w_init", "w_1x1", "w_3x3", "w_fc"]


def budget_mod():
    spec = importlib.util.spec_from_file_location("error_budget", os.path.join(ROOT, "tools", "error_budget.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class Setting(object):
    """acts: set of activation points rounded to fp16; wmode: {weight name: 'rtn' | 'coh'} (absent = exact)."""

    def __init__(self, name, acts=(), wmode=None, amode=None):
        """amode: {activation point: 'h8' | 'h4' | 'dither'} — what the point stores instead of plain fp16 ('h8': fp16 hi + an e4m3 lo
        with one power-of-two scale per tensor; 'h4': fp16 hi + an MX-fp4 (e2m1, one E8M0 scale per 32 channels) lo; 'dither': fp16
        with a position-dependent rounding offset)."""
        self.name, self.acts, self.wmode, self.amode = name, set(acts), dict(wmode or {}), dict(amode or {})

    def R(self, key, t):
        if key not in self.acts:
            return t
        m = self.amode.get(key)
        hi = t.half().float()
        if m is None:
            return hi
        lo = t - hi
        if m == "h8":
            sc = 2.0 ** torch.floor(torch.log2(256.0 / lo.abs().max().clamp_min(1e-30)))
            return hi + (lo * sc).to(torch.float8_e4m3fn).float() / sc
        if m == "h4":
            return hi + mxfp4(lo)
        if m == "shift_own" or m == "shift_cal":
            # store v - mu_c (a per-channel constant; the consumer's bias absorbs conv(W, mu)): the rounding error scales with |v - mu_c|
            # instead of |v| — 'own': the tensor's own channel means, 'cal': the calibration frames' (what a frozen stem can know)
            if key == "a_clip":
                mu = torch.full((1, t.shape[1], 1, 1), 0.5, device=t.device)
            elif m == "shift_cal":
                mu = SHIFT_MEANS[key].to(t.device).view(1, -1, 1, 1)
            else:
                mu = t.mean(dim=(0, 2, 3), keepdim=True)
            mu = mu.half().float()
            return (t - mu).half().float() + mu
        if m == "dither":
            u = torch.rand_like(t) - 0.5
            ulp = 2.0 ** (torch.floor(torch.log2(t.abs().clamp_min(6.2e-5))) - 10)
            return (t + u * ulp).half().float()
        raise ValueError(m)


def mxfp4(lo):
    """e2m1 values {0, .5, 1, 1.5, 2, 3, 4, 6} x 2^s, one shared s per block of 32 consecutive CHANNELS (dim 1) of a pixel"""
    n, c, h, w = lo.shape
    cb = 32 if c % 32 == 0 else c
    b = lo.view(n, c // cb, cb, h, w)
    amax = b.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    s = 2.0 ** (torch.floor(torch.log2(amax)) - 2)            # the block maximum lands in [4, 8) -> grid top 6
    v = (b / s).clamp(-6, 6)
    a = v.abs()
    q = torch.where(a < 2, torch.round(a * 2) / 2, torch.where(a < 4, torch.round(a), torch.round(a / 2) * 2))
    return (torch.sign(v) * q * s).view(n, c, h, w)


SHIFT_MEANS = {}      # activation point -> per-channel mean on the calibration frames (filled per weight seed in main)

STEM_W = ["sw_c11", "sw_c12", "sw_c21", "sw_c22", "sw_comp", "sw_od21", "sw_od22", "sw_od31", "sw_od32"]


def stem_weight_deltas(vgg, od, calib):
    """Per stem layer: (rounded - exact) of the weights the fp16 stem multiplies with — BatchNorm scale folded first, rounded
    coherently against the calibration means exactly as FrozenStem does; the composed pair as its 5x5 weight.  The simulation
    adds conv(x, delta) to the exact layer output (first-order exact; the composed pair's border is second order)."""
    from videonavqa_amd.stem import coherent_round, _fold_bn
    f = vgg.features
    d = {}
    cr = lambda w, m: coherent_round(w, calib[m], torch.float16) - w
    d["sw_c11"] = cr(f["0"].weight.float(), "first")
    d["sw_c12"] = cr(f["2"].weight.float(), "vgg0")
    d["sw_c21"] = cr(f["5"].weight.float(), "vgg1")
    d["sw_c22"] = cr(f["7"].weight.float(), "vgg2")
    s1, _ = _fold_bn(od.bn1)
    w1, w2 = od.conv11.weight.double().cpu(), od.conv12.weight.double().cpu() * s1.double().cpu().view(-1, 1, 1, 1)
    wc = F.conv2d(w1.permute(1, 0, 2, 3), w2.flip(2, 3), padding=2).permute(1, 0, 2, 3).float().to(f["0"].weight.device).contiguous()
    d["sw_comp"] = cr(wc, "od0")                       # acts on the composed output BEFORE relu / pool, already bn1-scaled
    d["sw_od21"] = cr(od.conv21.weight.float(), "od2")
    s2, _ = _fold_bn(od.bn2)
    d["sw_od22"] = cr(od.conv22.weight.float() * s2.view(-1, 1, 1, 1), "od3")      # bn2-scaled
    d["sw_od31"] = cr(od.conv31.weight.float(), "od4")
    s3, _ = _fold_bn(od.bn3)
    d["sw_od32"] = cr(od.conv32.weight.float() * s3.view(-1, 1, 1, 1), "od5")
    return d


@torch.no_grad()
def sim_stem(vgg, od, frames, st, chunk=40, wd=None):
    f = vgg.features
    wd = wd or {}

    def conv(t, c, key=None, pad=1):
        y = F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
        if key in st.acts and key in wd:
            y = y + F.conv2d(t, wd[key], None, padding=pad)
        return y

    def bnx(t, b, x_in=None, key=None, pad=1):
        """eval BatchNorm; a folded layer's weight perturbation (already scaled by gamma * rstd) is added after it"""
        y = F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, BN_EPS)
        if key in st.acts and key in wd:
            y = y + F.conv2d(x_in, wd[key], None, padding=pad)
        return y
    bn = lambda t, b: F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, BN_EPS)
    outs = []
    n = frames.shape[0]
    if n % chunk:       # fixed chunk shapes (ragged minibatches would otherwise make MIOpen search a new shape per batch)
        frames = torch.cat([frames, frames.new_zeros((chunk - n % chunk,) + tuple(frames.shape[1:]))])
    for i in range(0, frames.shape[0], chunk):
        x = st.R("a_clip", frames[i:i + chunk])
        a = st.R("a_c11", F.relu(conv(x, f["0"], "sw_c11")))
        a = st.R("a_c12", F.max_pool2d(F.relu(conv(a, f["2"], "sw_c12")), 2))
        a = st.R("a_c21", F.relu(conv(a, f["5"], "sw_c21")))
        a = st.R("a_c22", bn(F.max_pool2d(F.relu(conv(a, f["7"], "sw_c22")), 2), od.bn_input))     # (the producer applies bn_input, rounds once)
        a0 = a
        a = conv(a, od.conv11)                                                            # composed pair: never stored
        a = st.R("a_comp", F.max_pool2d(F.relu(bnx(conv(a, od.conv12), od.bn1, a0, "sw_comp", 2)), 2))
        a1 = st.R("a_od21", conv(a, od.conv21, "sw_od21"))
        a = st.R("a_od22", F.max_pool2d(F.relu(bnx(conv(a1, od.conv22), od.bn2, a1, "sw_od22")), 2))
        a1 = st.R("a_od31", conv(a, od.conv31, "sw_od31"))
        outs.append(st.R("a_feat", F.relu(bnx(conv(a1, od.conv32), od.bn3, a1, "sw_od32"))))
    return torch.cat(outs)[:n]


def lstm_cell(xg, h, c, w_hh, b_hh):
    g = xg + h @ w_hh.t() + b_hh
    H = h.shape[1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


@torch.no_grad()
def question_film(W, q, q_lens, cts):
    """FiLM values per processed frame: the question LSTM re-run per frame with carried state (film_attn_pt_stem.py:144-181)."""
    B = q.shape[0]
    Hq = W["film_layer.1.weight"].shape[1]
    emb = F.embedding(q, W["embed.weight"])
    xg = emb @ W["film_layer.0.weight_ih_l0"].t() + W["film_layer.0.bias_ih_l0"]
    h = emb.new_zeros(B, Hq)
    c = emb.new_zeros(B, Hq)
    Lmax = int(q_lens.max())
    ql = q_lens.to(q.device)
    out = []
    for ct in cts:
        last = emb.new_zeros(B, Hq)
        for t in range(Lmax):
            h2, c2 = lstm_cell(xg[:, t], h, c, W["film_layer.0.weight_hh_l0"], W["film_layer.0.bias_hh_l0"])
            m = (ql > t).float().unsqueeze(1)
            h = m * h2 + (1 - m) * h
            c = m * c2 + (1 - m) * c
            last = torch.where((ql == t + 1).unsqueeze(1), h2, last)
        out.append(F.relu(last[:ct] @ W["film_layer.1.weight"].t() + W["film_layer.1.bias"]))
    return out


@torch.no_grad()
def sim_trunk(W, Wq, feat, cts, film, B, T, st, taps=None):
    """feat [n_img, 512, h, w] frame-major; Wq = the (possibly rounded) conv_init / 1x1 / 3x3 / fc weights of this setting."""
    C = W["conv_init.weight"].shape[0]
    at = W["fc_attn_1.weight"].shape[1]
    all_features = []
    off = 0
    masks = feat.new_zeros(B, T, 1)
    valid = feat.new_zeros(B, T, 1)
    for i, ct in enumerate(cts):
        x = feat[off:off + ct]
        off += ct
        r = st.R("a_init", F.relu(F.conv2d(x, Wq["w_init"], W["conv_init.bias"], padding=1)))
        mean = r.mean(dim=(0, 2, 3))
        var = r.var(dim=(0, 2, 3), unbiased=False)
        xb = (r - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + BN_EPS)
        xb = st.R("a_bn", xb * W["bn_init.weight"].view(1, -1, 1, 1) + W["bn_init.bias"].view(1, -1, 1, 1))
        res = st.R("a_res", F.relu(F.conv2d(xb, Wq["w_1x1"], W["conv1x1_layers.0.bias"])))
        z = st.R("a_z", F.conv2d(res, Wq["w_3x3"], W["film_pipeline.0.bias"], padding=1))
        fv = film[i]
        g, b = fv[:, :C].view(-1, C, 1, 1), fv[:, C:2 * C].view(-1, C, 1, 1)
        out = st.R("a_out", F.relu(g * z + b) + res)
        if taps is not None:       # input means for the coherent rounding of the trunk's weights (exact pass)
            for k, t in (("w_init", x), ("w_3x3", res)):
                taps.setdefault(k, []).append((t.double().sum((0, 2, 3)), t.shape[0] * t.shape[2] * t.shape[3]))
            taps.setdefault("w_fc", []).append((out.reshape(ct, -1).double().sum(0), ct))
        fe = out.reshape(ct, -1) @ Wq["w_fc"].t() + W["fc_embed_attn.bias"]
        all_features.append(F.pad(fe, (0, 0, 0, B - ct)))
        masks[ct:, i, 0] = NEG_MASK
        valid[:ct, i, 0] = 1.0
    all_features = torch.stack(all_features, 0).permute(1, 0, 2)
    all_features = F.pad(all_features, (0, 0, 0, T - all_features.shape[1]))
    masks[:, len(cts):, 0] = 0.0
    features = (all_features @ W["fc_attn_1.weight"].t() + W["fc_attn_1.bias"]) * valid
    h = feat.new_zeros(B, at)
    cell = feat.new_zeros(B, at)
    hs = []
    for i in range(T):
        v_i = (h @ W["fc_hidden_attn.weight"].t() + W["fc_hidden_attn.bias"]).view(B, 1, 1)
        coefs = torch.softmax(v_i + features + masks, dim=1)
        ctxt = torch.bmm(coefs.permute(0, 2, 1), all_features).view(B, -1)
        g = ctxt @ W["lstm_attn.weight_ih"].t() + W["lstm_attn.bias_ih"]
        h, cell = lstm_cell(g, h, cell, W["lstm_attn.weight_hh"], W["lstm_attn.bias_hh"])
        hs.append(h)
    hs = torch.stack(hs, 1).reshape(B, -1)
    return hs @ W["out_linear.weight"].t() + W["out_linear.bias"]


def trunk_weights(W, st, means):
    from videonavqa_amd.stem import coherent_round
    src = {"w_init": W["conv_init.weight"], "w_1x1": W["conv1x1_layers.0.weight"], "w_3x3": W["film_pipeline.0.weight"],
           "w_fc": W["fc_embed_attn.weight"]}
    out = {}
    for k, w in src.items():
        mode = st.wmode.get(k)
        if mode is None:
            out[k] = w
        elif mode == "rtn" or (mode == "coh" and k == "w_1x1"):     # (BatchNorm output: zero mean at initialisation, nothing to cancel)
            out[k] = w.half().float()
        else:
            w4 = w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
            out[k] = coherent_round(w4, means[k], torch.float16).view_as(w)
    return out


def pack_frames(clip, v_lens):
    """Frame-major image list of the valid (sample, frame) pairs, samples in descending-length (stable) order."""
    B, _, H, W, T = clip.shape
    v_sorted, perm = torch.sort(v_lens, dim=0, descending=True, stable=True)
    cts = []
    for i in range(T):
        ct = int((v_sorted >= i + 1).sum())
        if ct == 0:
            break
        cts.append(ct)
    idx_b = torch.cat([perm[:ct] for ct in cts])
    idx_t = torch.cat([torch.full((ct,), i, dtype=torch.long) for i, ct in enumerate(cts)])
    frames = clip[idx_b.to(clip.device), :, :, :, idx_t.to(clip.device)]
    return frames, cts, v_sorted, perm


def settings_list(only_combos=False):
    S = []
    if not only_combos:
        for p in STEM_ACTS + TRUNK_ACTS:
            S.append(Setting(p, [p]))
        for w in TRUNK_W:
            S.append(Setting(w + ":rtn", wmode={w: "rtn"}))
        for w in ("w_init", "w_3x3", "w_fc"):
            S.append(Setting(w + ":coh", wmode={w: "coh"}))
    if not only_combos:
        for p in STEM_W:
            S.append(Setting(p + " (coherent, as the fp16 stem)", [p]))
    S += [Setting("stem weights (all 9, coherent)", STEM_W),
          Setting("stem weights - od31,od32", [w for w in STEM_W if w not in ("sw_od31", "sw_od32")]),
          Setting("stem weights - od22,od31,od32", [w for w in STEM_W if w not in ("sw_od22", "sw_od31", "sw_od32")])]
    fp16h = [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22")] + ["a_init", "a_bn", "a_res", "a_out"]
    S += [Setting("fp16h as built (acts + 3x3 rtn + stem weights)", fp16h + STEM_W, {"w_3x3": "rtn"}),
          Setting("fp16h + split weights conv31,conv32", fp16h + [w for w in STEM_W if w not in ("sw_od31", "sw_od32")], {"w_3x3": "rtn"}),
          Setting("fp16h + split weights conv22,conv31,conv32", fp16h + [w for w in STEM_W if w not in ("sw_od22", "sw_od31", "sw_od32")], {"w_3x3": "rtn"}),
          Setting("fp16h + split conv31,conv32 + a_init pair", [a for a in fp16h if a != "a_init"] + [w for w in STEM_W if w not in ("sw_od31", "sw_od32")], {"w_3x3": "rtn"}),
          Setting("fp16h + split conv31,conv32 + a_init,a_od21 pair", [a for a in fp16h if a not in ("a_init", "a_od21")] + [w for w in STEM_W if w not in ("sw_od31", "sw_od32")], {"w_3x3": "rtn"})]
    allw = {w: "rtn" for w in TRUNK_W}
    cohw = {w: "coh" for w in TRUNK_W}
    S += [Setting("stem acts (all 10)", STEM_ACTS),
          Setting("stem acts - clip", [a for a in STEM_ACTS if a != "a_clip"]),
          Setting("stem acts - feat", [a for a in STEM_ACTS if a != "a_feat"]),
          Setting("stem acts - feat,od31", [a for a in STEM_ACTS if a not in ("a_feat", "a_od31")]),
          Setting("stem acts - feat,od31,od22", [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22")]),
          Setting("stem acts - feat,od31,od22,clip", [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22", "a_clip")]),
          Setting("stem acts - feat,od31,od22,od21", [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22", "a_od21")]),
          Setting("trunk acts (all 5)", TRUNK_ACTS),
          Setting("trunk acts - z", [a for a in TRUNK_ACTS if a != "a_z"]),
          Setting("trunk w rtn (all 4)", wmode=allw),
          Setting("trunk w coh (all 4)", wmode=cohw),
          Setting("ALL acts + trunk w rtn  (~ precision fp16, exact stem weights)", STEM_ACTS + TRUNK_ACTS, allw),
          Setting("ALL acts + trunk w coh", STEM_ACTS + TRUNK_ACTS, cohw),
          Setting("ALL acts - z + trunk w coh", [a for a in STEM_ACTS + TRUNK_ACTS if a != "a_z"], cohw),
          # candidate modes: what stays rounded
          Setting("cand A: stem - feat,od31 ; trunk a_res only ; w coh(init,3x3)",
                  [a for a in STEM_ACTS if a not in ("a_feat", "a_od31")] + ["a_res"], {"w_init": "coh", "w_3x3": "coh"}),
          Setting("cand B: stem - feat,od31,od22 ; trunk a_res ; w coh(init,3x3)",
                  [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22")] + ["a_res"], {"w_init": "coh", "w_3x3": "coh"}),
          Setting("cand C: stem - feat,od31,od22 ; trunk exact",
                  [a for a in STEM_ACTS if a not in ("a_feat", "a_od31", "a_od22")]),
          Setting("cand D: stem all ; trunk a_res ; w coh(init,3x3)", STEM_ACTS + ["a_res"], {"w_init": "coh", "w_3x3": "coh"}),
          Setting("cand E: stem - od31 ; trunk a_res,a_out ; w coh(init,3x3,fc)",
                  [a for a in STEM_ACTS if a != "a_od31"] + ["a_res", "a_out"], {"w_init": "coh", "w_3x3": "coh", "w_fc": "coh"})]
    return S


def settings_round6():
    """Round 6: what is left of the fp16h precision's error, point by point, on clips of another kind (--data blocks ...), and what the
    candidate fixes leave."""
    left = ["a_clip", "a_c11", "a_c12", "a_c21", "a_c22", "a_comp", "a_od21"]
    trunk = ["a_init", "a_bn", "a_res", "a_out"]
    S = [Setting(p, [p]) for p in left + ["a_od22", "a_od31", "a_feat"] + trunk]
    base = left + trunk
    S.append(Setting("fp16h acts (7 stem + 4 trunk)", base))
    for p in left:
        S.append(Setting("fp16h acts - " + p, [a for a in base if a != p]))
    S.append(Setting("fp16h acts - clip,od21", [a for a in base if a not in ("a_clip", "a_od21")]))
    S.append(Setting("fp16h acts - clip,od21,comp", [a for a in base if a not in ("a_clip", "a_od21", "a_comp")]))
    S.append(Setting("fp16h acts - clip,c12,od21", [a for a in base if a not in ("a_clip", "a_c12", "a_od21")]))
    S.append(Setting("fp16h acts - clip,c12,c21,od21", [a for a in base if a not in ("a_clip", "a_c12", "a_c21", "a_od21")]))
    S.append(Setting("fp16h acts - clip,c12,c21,comp,od21", [a for a in base if a not in ("a_clip", "a_c12", "a_c21", "a_comp", "a_od21")]))
    S.append(Setting("fp16h acts - all stem (trunk only)", trunk))
    S.append(Setting("fp16h acts, od21 as hi + e4m3 lo", base, amode={"a_od21": "h8"}))
    S.append(Setting("fp16h acts, od21 as hi + mxfp4 lo", base, amode={"a_od21": "h4"}))
    S.append(Setting("fp16h acts, od21,comp as hi + mxfp4 lo", base, amode={"a_od21": "h4", "a_comp": "h4"}))
    S.append(Setting("fp16h acts, all 7 stem dithered", base, amode={p: "dither" for p in left}))
    S.append(Setting("stem acts (all 10)", STEM_ACTS))
    six = ["a_clip", "a_c12", "a_c21", "a_c22", "a_comp", "a_od21"]
    for mode in ("shift_own", "shift_cal"):
        S.append(Setting("fp16h acts, clip..od21 except c11 stored mean-shifted (%s)" % mode, base, amode={p: mode for p in six}))
        S.append(Setting("fp16h acts, all 7 stem stored mean-shifted (%s)" % mode, base, amode={p: mode for p in left}))
        for p in six[:1] + ["a_c11"] + six[1:]:
            S.append(Setting("%s mean-shifted (%s)" % (p, mode), [p], amode={p: mode}))
    S.append(Setting("w_3x3:rtn", wmode={"w_3x3": "rtn"}))
    S.append(Setting("w_init:rtn", wmode={"w_init": "rtn"}))
    S.append(Setting("w_1x1:rtn", wmode={"w_1x1": "rtn"}))
    S.append(Setting("w_fc:rtn", wmode={"w_fc": "rtn"}))
    S.append(Setting("stem weights (all 9, coherent)", STEM_W))
    S.append(Setting("early stem acts only (clip,c11,c12,c21,c22)", ["a_clip", "a_c11", "a_c12", "a_c21", "a_c22"]))
    S.append(Setting("trunk acts only (init,bn,res,out)", trunk))
    return S


def product_logits(args, prec, device, data):
    bm = budget_mod()
    return bm.run(args, prec, device, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="*", default=[0, 1, 2, 3])
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--data", default="noise", choices=["noise", "smooth", "blocks", "textured"])
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=224, help="--height 160 --width 208: the reference's own frames (10 x 13 maps)")
    ap.add_argument("--per-batch", action="store_true", help="print every setting's error per minibatch (last seed) as well")
    ap.add_argument("--round6", action="store_true", help="the round-6 settings (what is left of fp16h's error, and the candidate fixes)")
    ap.add_argument("--only-combos", action="store_true")
    ap.add_argument("--no-product", action="store_true", help="skip the cross-checks against the library's fp32 / fp16 precisions")
    o = ap.parse_args()
    from videonavqa_amd import _lib as L
    L.set_half("f16")
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    bm = budget_mod()
    S = settings_round6() if o.round6 else settings_list(o.only_combos)
    acc = {s.name: [] for s in S}
    per_seed, last_errs = {}, {}
    t0 = time.time()
    for seed in o.seeds:
        args = argparse.Namespace(precision="fp32", model="film_attn_pt", batch=8, frames=35, height=o.height, width=o.width, blocks=1, channels=512,
                                  tail_channels=0, seed=seed)
        data = bm.batches(args, dev, o.batches, o.data)
        model, stem, vgg, od = bench.build(args, dev)
        W = {k: v.detach().float() for k, v in model.state_dict().items()}
        W.update({k: v.detach().float() for k, v in model.extra_state_tensors().items()})
        exact = Setting("exact")
        refs, packed, means = [], [], None
        for bi, (clip, q, v_lens, q_lens) in enumerate(data):
            frames, cts, v_sorted, perm = pack_frames(clip.to(dev), v_lens)
            film = question_film(W, q.to(dev)[perm.to(dev)], q_lens[perm], cts)
            feat = sim_stem(vgg, od, frames, exact)
            taps = {} if bi == 0 else None
            ref = sim_trunk(W, trunk_weights(W, exact, None), feat, cts, film, 8, 35, exact, taps)
            if taps is not None:    # trunk input means from minibatch 0 (a deployment: running means of the previous steps)
                means = {k: (sum(s for s, _ in v) / sum(n for _, n in v)).float() for k, v in taps.items()}
            refs.append(ref.cpu())
            packed.append((frames.cpu(), cts, film, perm, v_sorted, feat.cpu()))
        if not o.no_product:
            got = product_logits(args, "fp32", dev, data)
            e = [float((g - r).abs().max() / r.abs().max()) for g, r in zip(got, refs)]
            print("seed %d: restatement (exact) vs library precision 'fp32': max %.2e" % (seed, max(e)), flush=True)
            for prec in ("fp16", "fp16h"):
                got = product_logits(args, prec, dev, data)
                e = [float((g - r).abs().max() / r.abs().max()) * 1e3 for g, r in zip(got, refs)]
                print("seed %d: library precision '%s' vs restatement (exact): max %.3f rms %.3f   %s" %
                      (seed, prec, max(e), (sum(x * x for x in e) / len(e)) ** 0.5, " ".join("%.2f" % x for x in e)), flush=True)
        from videonavqa_amd.stem import calibration_means
        cal = calibration_means(vgg, od, n_frames=40)
        for key, ck in (("a_c11", "vgg0"), ("a_c12", "vgg1"), ("a_c21", "vgg2"), ("a_c22", "od0"), ("a_comp", "od2"), ("a_od21", "od3")):
            SHIFT_MEANS[key] = cal[ck].float()
        wdel = stem_weight_deltas(vgg, od, cal)
        del model, stem
        torch.cuda.empty_cache()
        stem_cache = {}
        for s in S:
            wq = trunk_weights(W, s, means)
            errs = []
            sk = tuple(sorted(a for a in s.acts if a in STEM_ACTS or a in STEM_W)) + tuple(sorted(s.amode.items()))
            for bi, (frames, cts, film, perm, v_sorted, feat0) in enumerate(packed):
                if not sk:
                    feat = feat0.to(dev)
                else:
                    if (sk, bi) not in stem_cache:
                        stem_cache[(sk, bi)] = sim_stem(vgg, od, frames.to(dev), s, wd=wdel).cpu()
                    feat = stem_cache[(sk, bi)].to(dev)
                out = sim_trunk(W, wq, feat, cts, film, 8, 35, s).cpu()
                errs.append(float((out - refs[bi]).abs().max() / refs[bi].abs().max()) * 1e3)
            acc[s.name] += errs
            last_errs[s.name] = errs
            per_seed.setdefault(s.name, []).append(max(errs))
            if len(stem_cache) > 36:
                stem_cache.clear()
        print("seed %d done (%.0f s)" % (seed, time.time() - t0), flush=True)
    print("\n%-78s %7s %7s %7s   max per seed" % ("setting (x 1e-3; sq = mean squared, x 1e-6)", "sq", "rms", "max"))
    for s in S:
        e = acc[s.name]
        sq = sum(x * x for x in e) / len(e)
        print("%-78s %7.4f %7.3f %7.3f   %s%s" % (s.name, sq, sq ** 0.5, max(e), " ".join("%.2f" % x for x in per_seed[s.name]),
                                                  ("   | " + " ".join("%.2f" % x for x in last_errs[s.name])) if o.per_batch else ""), flush=True)


if __name__ == "__main__":
    main()
