"""GPU, BASELINE.json full size (8 clips x 35 frames x 224x224, bf16): size-independent properties.

* the fused stem on all 280 frames (activation buffers > 1.8 GB: exercises the 64-bit address arithmetic)
  against a plain PyTorch fp32 conv chain evaluated on a few sampled frames;
* linearity in the batch: features of a frame do not depend on which other frames share the launch;
* a few full-size training steps: finite, loss decreases on a fixed batch, gradients were zeroed.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


def _build(seed=0):
    from videonavqa_amd.models import ObjDetectCNN
    from videonavqa_amd.stem import FrozenStem, VGGFront
    torch.manual_seed(seed)
    vgg = VGGFront(LOW)
    od = ObjDetectCNN(27, 512, 1024, 0, True, True, precision=LOW)
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
            conv.bias.normal_(0, 0.02)
        for m in od.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1.0)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.8, 1.2)
    vgg, od = vgg.cuda().eval(), od.cuda().eval()
    return vgg, od, FrozenStem(vgg, od, LOW)


def _torch_stem(frames, vgg, od):
    """plain PyTorch fp32 reference of the frozen stem on [n,3,H,W] frames"""
    f = vgg.features
    x = F.relu(F.conv2d(frames, f["0"].weight, f["0"].bias, padding=1))
    x = F.max_pool2d(F.relu(F.conv2d(x, f["2"].weight, f["2"].bias, padding=1)), 2)
    x = F.relu(F.conv2d(x, f["5"].weight, f["5"].bias, padding=1))
    x = F.max_pool2d(F.relu(F.conv2d(x, f["7"].weight, f["7"].bias, padding=1)), 2)

    def bn(x, m):
        return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, 1e-5)

    x = bn(x, od.bn_input)
    x = F.max_pool2d(F.relu(bn(od.conv12(od.conv11(x)), od.bn1)), 2)
    x = F.max_pool2d(F.relu(bn(od.conv22(od.conv21(x)), od.bn2)), 2)
    return F.relu(bn(od.conv32(od.conv31(x)), od.bn3))


@pytest.mark.parametrize("B", [8, 40])      # 40 clips = 1400 frames: conv2_1's output alone is 4.66 GB (> 2^32 bytes)
def test_full_size_stem_vs_torch_reference_and_batch_independence(B):
    from videonavqa_amd import kernels as K
    from videonavqa_amd.models.common import FrameLayout
    vgg, od, stem = _build()
    T, H, W = 35, 224, 224
    g = torch.Generator().manual_seed(1)
    clip = torch.rand(B, 3, H, W, T, generator=g).cuda()
    lay = FrameLayout([T] * B, T, "cuda")
    feats = stem.forward_clip(clip, lay.img_of, lay.n_img)
    assert feats.shape == (B * T, 16, 16, 512)
    plain = stem.plain_features(feats)          # (mean-shifted storage: the stored tensor holds feature - mean_c, its halo -mean_c)
    assert float(plain[:, 0].abs().max()) == 0 and float(plain[:, :, -1].abs().max()) == 0
    picks = [(0, 0), (3, 17), (B - 1, T - 1)]                  # first image, middle, LAST image (highest addresses)
    with torch.no_grad():
        for b, t in picks:
            n = t * B + b
            ref = _torch_stem(clip[b:b + 1, :, :, :, t], vgg, od)
            got = plain[n:n + 1, 1:-1, 1:-1].permute(0, 3, 1, 2)
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err < 6e-2, (b, t, err)                     # ten stacked bf16 layers vs fp32
    # the same frames through a 3-frame launch give the same bf16 features bit for bit
    small = torch.stack([clip[b, :, :, :, t] for b, t in picks], 0).unsqueeze(-1).contiguous()   # [3,3,H,W,1]
    lay3 = FrameLayout([1, 1, 1], 1, "cuda")
    kept = [feats[t * B + b].clone() for b, t in picks]
    f3 = stem.forward_clip(small, lay3.img_of, lay3.n_img, slot=1)
    for i in range(3):
        assert torch.equal(f3[i], kept[i])


def _bench_args(**kw):
    import argparse
    d = dict(precision=LOW, batch=8, frames=35, height=224, width=224, blocks=1, channels=512, model="film_attn_pt",
             tail_channels=0)
    d.update(kw)
    return argparse.Namespace(**d)


# BASELINE.json configs 4 (the metric's), 3, 5 and the three eval.sh presets (/root/reference/eval.sh:8-41)
FULL_SIZE_CONFIGS = {
    "config4_film_attn": dict(),
    "config3_film_gp": dict(model="film_gp_pt"),
    "config5_time_multi_hop_T70": dict(model="time_multi_hop", frames=70),
    "evalsh_film_attn_5x1024_bs32": dict(blocks=5, channels=1024, batch=32),
    "evalsh_film_gp_4x1024_bs32": dict(model="film_gp_pt", blocks=4, channels=1024, batch=32, tail_channels=32),   # eval.sh:30-31
    "evalsh_time_multi_hop_3x1024_bs16": dict(model="time_multi_hop", blocks=3, channels=1024, batch=16, tail_channels=64),   # eval.sh:10-13,24
}


@pytest.mark.parametrize("name", list(FULL_SIZE_CONFIGS))
def test_full_size_training_steps_are_sane(name):
    import bench as Bn
    from videonavqa_amd.train import Trainer
    args = _bench_args(**FULL_SIZE_CONFIGS[name])
    dev = torch.device("cuda", 0)
    model, stem, vgg, od = Bn.build(args, dev)
    tr = Trainer(model, stem, lr=1e-4)
    batch = Bn.synth_batch(args, 0, dev)
    losses = []
    for _ in range(6):
        loss, logits = tr.step(*batch, next_clip=batch[0], next_v_lens_cpu=batch[2])
        losses.append(float(loss))
        assert logits.shape == (args.batch, 70) and bool(torch.isfinite(logits).all())
    assert all(l == l and l < 1e4 * args.batch for l in losses)
    assert losses[-1] < losses[0], losses                        # fixed batch: the loss goes down
    assert float(tr.fp.grad.abs().max()) == 0.0                  # fused zero_grad
    assert bool(torch.isfinite(tr.fp.flat).all())


# Stated tolerance of the bf16 benchmark precision against the exact-f32 parity precision AT FULL SIZE (north star:
# logits within 1e-3 relative with the answer-class argmax bit-exact; the fp32 mode itself is pinned <= 1e-3 to the
# reference goldens in tests/test_gpu_models.py).  Error = max |logit_bf16 - logit_fp32| / max |logit_fp32| over a
# minibatch, worst of three minibatches (full-length and ragged), train-mode forward.
# The attention models average frame features (errors of independent frames partly cancel): stated 1.2e-2 — measured 0.0091
# worst of three minibatches at the headline config (round 3; 0.0095 in round 2: the figure moves by a few % whenever a kernel
# change reorders roundings, so the stated tolerance keeps a 30 % margin instead of round 2's 5 %).  The global-max-pooling
# heads (film_gp_pt, time_multi_hop) pick ONE frame per feature, so a near-tie between frames that flips under bf16
# rounding moves that feature by the whole difference: stated looser, 3e-2.
# Round 4: with the frozen stem's weights rounded coherently (stem.coherent_round, the default of every 16-bit precision) the headline
# config measures 7.1e-3 (bf16) / 0.81e-3 (fp16) instead of 9.1e-3 / 1.30e-3: stated 9.5e-3 and 1.1e-3.
# Round 5: every stated tolerance = the measured value + 30 % (profiles/r05_gpu_tests.txt's run: 7.1e-3, 1.64e-3,
# 1.64e-2, 1.26e-2, 1.63e-2, 0.90e-2 in this order; flat-gradient errors 0.019, 0.0068, 0.19, 0.24, 0.33, 0.13).
# Round 6 (mean-shifted storage of the stem's activations and features, one storage rounding after bn_input's affine — DESIGN.md section 4 —
# apply to EVERY 16-bit precision): measured, in the order of FULL_SIZE_CONFIGS, bf16 logits 4.2e-3, 8.8e-3, 7.3e-3, 1.3e-3, 9.5e-3, 6.2e-3
# (flat gradient 0.012, 0.146, 0.195, 0.0049, 0.249, 0.104); fp16 logits 0.53e-3, 1.58e-3, 0.76e-3, 0.20e-3, 1.21e-3, 0.72e-3 (gradient
# 0.0040, 0.051, 0.072, 0.0015, 0.089, 0.038).  Stated = measured + 35 % (these pin regressions; the COMPLIANCE claim — 1e-3 — belongs to
# precision 'fp16h' and is asserted at the tolerance itself in tests/test_gpu_fp16h.py).
FULL_SIZE_LOGIT_TOL = {
    "bf16": {"config4_film_attn": 5.7e-3, "config3_film_gp": 1.2e-2, "config5_time_multi_hop_T70": 9.9e-3,
             "evalsh_film_attn_5x1024_bs32": 1.75e-3, "evalsh_film_gp_4x1024_bs32": 1.3e-2, "evalsh_time_multi_hop_3x1024_bs16": 8.4e-3},
    "fp16": {"config4_film_attn": 7.2e-4, "config3_film_gp": 2.15e-3, "config5_time_multi_hop_T70": 1.05e-3,
             "evalsh_film_attn_5x1024_bs32": 2.7e-4, "evalsh_film_gp_4x1024_bs32": 1.65e-3, "evalsh_time_multi_hop_3x1024_bs16": 9.8e-4}}
FULL_SIZE_GRAD_TOL = {
    "bf16": {"config4_film_attn": 0.016, "config3_film_gp": 0.20, "config5_time_multi_hop_T70": 0.265,
             "evalsh_film_attn_5x1024_bs32": 0.0067, "evalsh_film_gp_4x1024_bs32": 0.34, "evalsh_time_multi_hop_3x1024_bs16": 0.14},
    "fp16": {"config4_film_attn": 0.0055, "config3_film_gp": 0.069, "config5_time_multi_hop_T70": 0.098,
             "evalsh_film_attn_5x1024_bs32": 0.0021, "evalsh_film_gp_4x1024_bs32": 0.121, "evalsh_time_multi_hop_3x1024_bs16": 0.051}}


@pytest.mark.parametrize("name", list(FULL_SIZE_CONFIGS))
def test_bf16_vs_fp32_mode_logits_argmax_at_full_size(name):
    import json
    import os
    import bench as Bn
    args = _bench_args(**FULL_SIZE_CONFIGS[name])
    res = Bn.precision_parity(args, torch.device("cuda", 0), speed_steps=2)
    out_dir = os.path.join(os.path.dirname(os.path.abspath(Bn.__file__)), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_%s%s.json" % (name, "" if LOW == "bf16" else "_" + LOW)), "w") as fh:
        json.dump(res, fh, indent=1)
    print(name, json.dumps(res))
    tol = FULL_SIZE_LOGIT_TOL[LOW][name]
    err = res[LOW + "_logits_rel_err"]
    assert err < tol, res
    assert res["loss_rel_err"] < max(tol, 1.5e-3), res
    # an argmax can only flip where the fp32 top-2 gap is below twice the logits error: every flipped sample of the
    # untrained net must be such a near-tie ...
    assert all(g < 2 * err for g in res["fp32_top2_gap_rel_of_flipped_at_init"]), res
    # ... and on weights that have fit their minibatch (decisive predictions) the answer classes are identical
    fit = res["after_fit"]
    assert fit[LOW + "_logits_rel_err"] < tol, res
    if fit["fp32_min_top2_gap_rel"] > 2 * tol:
        assert fit["argmax_equal"], res
    assert abs(fit[LOW + "_loss"] - fit["fp32_loss"]) < 5 * tol * max(1.0, fit["fp32_loss"]), res
    # flat-gradient error of one backward pass at initialisation.  Attention models: 5 % (measured 2.2 % /
    # 0.8 %).  Max-pooling heads: each pooled feature's whole gradient goes to ONE frame and ~10 % of the features pick another
    # frame under bf16 rounding (1.3 % under fp16, 0.9 % under fp16h): measured 0.13 - 0.33 bf16 (stated per config above), and
    # routing the backward by the fp32 run's arg-max frames must not make it worse.
    gtol = FULL_SIZE_GRAD_TOL[LOW][name]
    assert res["grad_rel_l2_err"] < gtol, res
    if res.get("pooling_head"):
        ph = res["pooling_head"]
        assert ph["grad_rel_l2_err_routed_by_fp32_argmax"] <= res["grad_rel_l2_err"] * 1.02, res
        assert ph["argmax_frame_flip_frac"] < (0.088 if LOW == "bf16" else 0.012), res      # (round 6: measured <= 0.065 bf16 / 0.0088 fp16)


def test_full_size_trunk_wgrad_vs_torch_and_additivity():
    """The trunk's weight-gradient launch at BASELINE size (280 images x 14x14, 512 -> 512, 3x3: 36 tiles x 7 pixel slices,
    XCD-remapped) against torch's fp32 conv backward on the same bf16-rounded operands, and additivity over a split of
    the images (dW of all = dW of the first 100 + dW of the rest)."""
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(2)
    N, H, W, C = 280, 14, 14, 512
    x = torch.randn(N, C, H, W, generator=g).cuda().to(LOW_DTYPE).float()
    dy = torch.randn(N, C, H, W, generator=g).cuda().to(LOW_DTYPE).float()
    w = torch.zeros(C, C, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    xn, dyn = K.nchw_to_nhwc(x, LOW_DTYPE, c_pad=C), K.nchw_to_nhwc(dy, LOW_DTYPE, c_pad=C)
    dwt, dbias = K.conv2d_wgrad(xn, dyn, 9)
    dw = K.unpack_conv_wgrad(dwt, C, C)
    scale = float(w.grad.abs().max())
    assert float((dw - w.grad).abs().max()) < 2e-4 * scale
    assert float((dbias - dy.sum((0, 2, 3))).abs().max()) < 2e-4 * float(dy.sum((0, 2, 3)).abs().max())
    a, _ = K.conv2d_wgrad(xn[:100].contiguous(), dyn[:100].contiguous(), 9)
    b, _ = K.conv2d_wgrad(xn[100:].contiguous(), dyn[100:].contiguous(), 9)
    assert float((a + b - dwt).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("ragged", [False, True])
def test_hip_fp32_vs_oracle_at_baseline_geometry(ragged):
    """the HIP path against the ORACLE ITSELF at BASELINE.json's geometry — 35 frames of 224 x 224 through the
    full-width stem (VGG-16[:10] + ObjDetectCNN(512)) and the default FiLM-attn model — on B = 2 clips (what the oracle
    finishes in seconds on the host cores).  precision='fp32' through the C ABI vs oracle.stem_forward + film_attn_forward:
    stem features <= 1e-3 of their max, logits <= 1e-3 of max |logit|, answer classes identical (north star's tolerance)."""
    import argparse
    import bench as Bn
    from oracle import vnqa_oracle as O
    from videonavqa_amd import kernels as K
    from videonavqa_amd.models import FiLMAttnPretrainedStem, ObjDetectCNN
    from videonavqa_amd.models.common import FrameLayout, NativeFeatures
    from videonavqa_amd.stem import FrozenStem, VGGFront
    B, T, H, Wd = 2, 35, 224, 224
    args = argparse.Namespace(height=H, width=Wd, frames=T, channels=512, blocks=1)
    W_vgg, W_od, W, (clip, q, v_lens, q_lens, y) = Bn.oracle_workload(args, B)
    if ragged:
        v_lens = torch.tensor([T, 11])
        clip = clip * (torch.arange(T).view(1, 1, 1, 1, T) < v_lens.view(B, 1, 1, 1, 1)).float()
    # ---- oracle (CPU, fp32) ----
    with torch.no_grad():
        feats_o = O.stem_forward(clip, W_vgg, W_od)                       # [B,512,14,14,T]
        v2, q2, vl2, ql2, y2, perm_o = O.sort_batch(feats_o, q, v_lens, q_lens, y)
        ref = O.film_attn_forward({k: v.clone() for k, v in W.items()}, v2, q2, vl2, ql2, training=True)
    # ---- product (HIP, exact-f32 precision) ----
    dev = torch.device("cuda", 0)
    vgg, od = VGGFront("fp32"), ObjDetectCNN(27, 512, 1024, 0, True, True, precision="fp32")
    vgg.load_state_dict(W_vgg)
    od.load_state_dict(W_od, strict=False)
    model = FiLMAttnPretrainedStem(B, 128, 70, max_num_frames=T, spatial_size=196, precision="fp32")
    vgg, od, model = vgg.to(dev).eval(), od.to(dev).eval(), model.to(dev).train()
    model.load_reference_tensors(W)
    stem = FrozenStem(vgg, od, "fp32")
    v_sorted, perm = torch.sort(v_lens, dim=0, descending=True, stable=True)
    lay = FrameLayout(v_sorted, T, dev, perm=perm)
    feats = stem.forward_clip(clip.to(dev), lay.img_of, lay.n_img)
    # stem features of the first and the last valid image against the oracle's
    for b, t in ((0, 0), (int(perm[0]), T - 1)):
        n = int(lay.img_of[b * T + t])
        got = K.nhwc_to_nchw(feats[n:n + 1].contiguous(), 512)[0].cpu()
        want = feats_o[b, :, :, :, t]
        assert float((got - want).abs().max() / want.abs().max()) < 1e-3
    model.init_hidden()
    with torch.no_grad():
        out = model(NativeFeatures(feats, lay, 512, H // 16, Wd // 16), q.to(dev)[perm.to(dev)], v_sorted, q_lens[perm])
    out_s = out.float().cpu()
    out, ref_s = torch.empty_like(out_s), ref
    out[perm] = out_s                                       # both back to the ORIGINAL sample order (the oracle's sort is
    ref = torch.empty_like(ref_s)                           # not stable: equal lengths may come out in another order)
    ref[perm_o] = ref_s
    err = float((out - ref).abs().max() / ref.abs().max())
    print("HIP fp32 vs oracle at 2 x 35 x 224 x 224 (ragged=%s): logits rel err %.3g" % (ragged, err))
    assert err < 1e-3, err                                  # north star: logits within 1e-3 rel of reference
    assert torch.equal(out.argmax(1), ref.argmax(1))        # answer-class argmax bit-exact
