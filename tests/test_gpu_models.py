"""GPU parity of the product (HIP) models against golden vectors captured from the reference
(tests/golden/*.npz, tools/capture_goldens.py) and against the CPU oracle.

Tolerances (north star: logits within 1e-3 relative of the reference, argmax bit-exact):
  precision='fp32' (exact-f32 MFMA path): logits 1e-3 * max|logits|, gradients 2e-3 * max|grad|
  precision='bf16' (bf16 storage, fp32 accumulate): logits 2.5e-2 * max|logits| (measured worst over all cases 1.26e-2,
                   tools/measure_smallnet_tol.py; round 2 stated 6e-2);
                   gradients of these 8-channel nets: relative L2 error < 0.3 per tensor and worst element <= 0.6 max|grad|
                   (measured worst 0.296 / 0.53, both on the bag-of-words max-pooling case: a frame arg-max that flips
                   under rounding moves a whole feature's gradient; BN backward cancels large terms).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import QV_CASES, build_product_model, load_golden, rel_err, weights_from, LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu

# Round 6: the 16-bit tolerances are measured + 30 % (tools/measure_smallnet_tol.py, worst over all cases of these
# 8-channel nets — where ONE 16-bit rounding is 1e-3 of the logits: the full-size 1e-3 claim is tests/test_gpu_fp16h.py's): logits bf16
# 1.26e-2 / fp16 5.0e-3; per-tensor gradient rel. L2 bf16 0.296 / fp16 0.125; worst element bf16 0.53 / fp16 0.16 of max |grad|.
LOGIT_TOL = {"fp32": 1e-3, "bf16": 1.65e-2, "fp16": 6.6e-3}
GRAD_L2_TOL = {"bf16": 0.385, "fp16": 0.165}
GRAD_ELEM_TOL = {"bf16": 0.69, "fp16": 0.21}


def _inputs(g):
    return tuple(torch.from_numpy(g[k]).cuda() for k in ("v", "q", "v_lens", "q_lens", "y"))


@pytest.mark.parametrize("fused_eval", [True, False])
@pytest.mark.parametrize("precision", ["fp32", LOW])
@pytest.mark.parametrize("case", QV_CASES)
def test_eval_logits_vs_reference_golden(case, precision, fused_eval, monkeypatch):
    """Eval-mode logits against the reference's own (tests/golden): on the round-4 inference path (forward-only fused trunk:
    eval BatchNorm folded into conv_init's epilogue, FILM_RES without the z output, packed tails — the default under
    no_grad) and on the op-by-op eval graph (VNQA_FUSED_EVAL=0), same tolerance."""
    monkeypatch.setenv("VNQA_FUSED_EVAL", "1" if fused_eval else "0")
    model, g = build_product_model(case, precision)
    v, q, vl, ql, y = _inputs(g)
    model.eval()
    assert model._use_fused_trunk() is False            # eval mode with autograd on: the op-by-op graph
    with torch.no_grad():
        assert model._use_fused_trunk() is fused_eval
        model.init_hidden()
        logits = model(v, q, vl, ql)
    got = logits.float().cpu().numpy()
    assert got.shape == g["eval_logits"].shape
    assert rel_err(got, g["eval_logits"]) < LOGIT_TOL[precision], rel_err(got, g["eval_logits"])
    if precision == "fp32":
        assert (got.argmax(1) == g["eval_logits"].argmax(1)).all()


@pytest.mark.parametrize("precision", ["fp32", LOW])
@pytest.mark.parametrize("case", QV_CASES)
def test_train_forward_backward_vs_reference_golden(case, precision):
    model, g = build_product_model(case, precision)
    v, q, vl, ql, y = _inputs(g)
    model.train()
    model.init_hidden()
    logits = model(v, q, vl, ql)
    loss = nn.CrossEntropyLoss(reduction="sum")(logits, y)
    loss.backward()
    got = logits.detach().float().cpu().numpy()
    tol = LOGIT_TOL[precision]
    assert rel_err(got, g["train_logits"]) < tol, rel_err(got, g["train_logits"])
    assert abs(float(loss.detach()) - float(g["train_loss"])) < 5 * tol * max(1.0, abs(float(g["train_loss"])))
    if precision == "fp32":
        assert (got.argmax(1) == g["train_logits"].argmax(1)).all()
    assert rel_err(model.bn_init.running_mean.cpu().numpy(), g["bn_running_mean_after"]) < tol
    assert rel_err(model.bn_init.running_var.cpu().numpy(), g["bn_running_var_after"]) < tol
    # carried LSTM state is kept in q_len-sorted order like the reference (ties: stable order)
    if "film_hidden_h_after" in g:          # (the bag-of-words encoder carries no state)
        assert rel_err(model.film_hidden[0][0].cpu().numpy(), g["film_hidden_h_after"][0]) < tol
    # fp32: every element within 2e-3 of the tensor's max |grad|.  bf16 (stated, looser): ReLU/FiLM
    # masks of near-zero activations may flip under bf16 rounding, so single elements can move a
    # lot on these tiny nets; bound the relative L2 error per tensor and the worst element.
    checked = 0
    for name, p in model.named_parameters():
        ref = g["grad/" + name]
        got_g = np.zeros_like(ref) if p.grad is None else p.grad.float().cpu().numpy()
        scale = np.abs(ref).max()
        err = np.abs(got_g - ref).max()
        if precision == "fp32":
            assert err <= 2e-3 * scale + 1e-6, (name, err, scale)
        else:
            l2 = np.linalg.norm(got_g - ref) / (np.linalg.norm(ref) + 1e-9)
            assert l2 < GRAD_L2_TOL[precision] or np.linalg.norm(ref) < 1e-5, (name, l2)
            assert err <= GRAD_ELEM_TOL[precision] * scale + 1e-6, (name, err, scale)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("case", ["film_attn_ragged", "film_gp_full", "tmh_ragged"])
def test_training_trajectory_vs_reference_golden(case):
    """3 steps: CE(sum) -> clip_grad_norm 1.0 -> Adam (eval/q_and_v_eval.py:124-139).  The goldens were
    captured on a CUDA-less box where the reference leaves film_layer out of the optimiser
    (SURVEY §0.6): freeze it here to replay that exact trajectory."""
    model, g = build_product_model(case, "fp32")
    v, q, vl, ql, y = _inputs(g)
    if hasattr(model, "film_layer"):
        for p in model.film_layer.parameters():
            p.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    loss_fn = nn.CrossEntropyLoss(reduction="sum")
    model.train()
    with torch.no_grad():     # the capture ran one train-mode forward first (BN running stats)
        model.init_hidden()
        model(v, q, vl, ql)
    opt = torch.optim.Adam(params, lr=float(g["traj_lr"]))
    losses = []
    for _ in range(len(g["traj_losses"])):
        model.init_hidden()
        loss = loss_fn(model(v, q, vl, ql), y)
        losses.append(float(loss))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        opt.zero_grad()
    assert np.allclose(losses, g["traj_losses"], rtol=2e-3, atol=1e-3), (losses, g["traj_losses"])
    model.eval()
    with torch.no_grad():
        model.init_hidden()
        logits = model(v, q, vl, ql).cpu().numpy()
    assert rel_err(logits, g["traj_final_eval_logits"]) < 5e-3


@pytest.mark.parametrize("precision", ["fp32", LOW])
def test_obj_detect_cnn_vs_reference_golden(precision):
    from videonavqa_amd.models import ObjDetectCNN
    g = load_golden("objdet_f16")
    m = ObjDetectCNN(nb_classes=5, num_filters=16, tail_hidden_dim=8, tail_dropout_p=0, logits=True,
                     pretrained_features=True, precision=precision)
    m.load_state_dict({k: v for k, v in weights_from(g, "w").items()})
    m = m.cuda().eval()
    y = m(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < (1e-4 if precision == "fp32" else 3e-2), rel_err(y, g["y"])


@pytest.mark.parametrize("precision", ["fp32", LOW])
def test_fused_stem_vs_oracle(precision):
    """clip [B,3,H,W,T] -> packed native features, against the oracle's per-frame stem loop
    (eval/q_and_v_eval.py:102-110) with seeded synthetic weights; ragged frame validity."""
    from oracle import vnqa_oracle as O
    from videonavqa_amd.models import ObjDetectCNN
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem, VGGFront
    from videonavqa_amd import kernels as K
    torch.manual_seed(7)
    B, T, H, W, NF = 3, 4, 32, 48, 16
    vgg = VGGFront(precision)
    od = ObjDetectCNN(5, NF, 8, 0, True, True, precision=precision)
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
            conv.bias.normal_(0, 0.05)
        for m in od.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.4)
                m.bias.normal_(0, 0.1)
    W_vgg = {k: v.clone() for k, v in vgg.state_dict().items()}
    W_od = {k: v.clone() for k, v in od.state_dict().items()}
    clip = torch.rand(B, 3, H, W, T)
    ref = O.stem_forward(clip, W_vgg, W_od)                       # [B,NF,h,w,T]
    vgg, od = vgg.cuda().eval(), od.cuda().eval()
    lay = FrameLayout([4, 3, 1], T, "cuda")
    stem = FrozenStem(vgg, od, precision)
    feats = stem.plain_features(stem.forward_clip(clip.cuda(), lay.img_of, lay.n_img))
    got = feats[:, 1:-1, 1:-1, :NF].permute(0, 3, 1, 2).cpu()
    tol = 1e-4 if precision == "fp32" else 4e-2
    for n in range(lay.n_img):
        t, b = int(lay.frame_of[n]), int(lay.sample_of[n])
        r = ref[b, :, :, :, t]
        assert float((got[n] - r).abs().max() / (r.abs().max() + 1e-9)) < tol, (n, t, b)
    # drop-in per-module path agrees with the oracle too
    f = vgg(clip[:, :, :, :, 0].cuda())
    rf = O.vgg_front(clip[:, :, :, :, 0], W_vgg)
    assert float((f.cpu() - rf).abs().max() / rf.abs().max()) < tol


def test_q_only_lstm_vs_reference_golden():
    """config-1 plumbing model (models/q_only_lstm.py) through the persistent LSTM kernel."""
    from videonavqa_amd.models import QOnlyLSTM
    g = load_golden("qonly_small")
    B, L = g["q"].shape
    m = QOnlyLSTM(B, 12, 16, 7, 20).cuda()
    m.load_state_dict(weights_from(g, "w"))
    m.hidden_1 = (torch.from_numpy(g["h0"]).cuda(), torch.from_numpy(g["c0"]).cuda())
    with torch.no_grad():
        logits = m(torch.from_numpy(g["q"]).cuda(), torch.from_numpy(g["q_lens"]))
    assert rel_err(logits.cpu().numpy(), g["logits"]) < 1e-4
