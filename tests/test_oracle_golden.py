"""CPU: the oracle restatement vs golden vectors captured from the reference
(tools/capture_goldens.py).  Pins oracle/vnqa_oracle.py before it is trusted as the checker."""
import numpy as np
import pytest
import torch

from oracle import vnqa_oracle as O
from helpers import QV_CASES, load_golden, model_of_case, rel_err, weights_from

FROZEN_CPU = ("conv1x1_layers.", "film_layer.")  # CPU flavour of the reference


def _inputs(g):
    return tuple(torch.from_numpy(g[k]) for k in ("v", "q", "v_lens", "q_lens", "y"))


@pytest.mark.parametrize("case", QV_CASES)
def test_eval_logits(case):
    g = load_golden(case)
    W = weights_from(g, "w0")
    v, q, vl, ql, y = _inputs(g)
    with torch.no_grad():
        logits = O.FORWARDS[model_of_case(case)](W, v, q, vl, ql, training=False)
    assert rel_err(logits.numpy(), g["eval_logits"]) < 2e-5
    assert (logits.argmax(1).numpy() == g["eval_logits"].argmax(1)).all()


@pytest.mark.parametrize("case", QV_CASES)
def test_train_forward_backward(case):
    g = load_golden(case)
    W = weights_from(g, "w0")
    v, q, vl, ql, y = _inputs(g)
    names = [k for k in W if W[k].is_floating_point() and "running" not in k]
    for k in names:
        W[k].requires_grad_(True)
    aux = {}
    logits = O.FORWARDS[model_of_case(case)](W, v, q, vl, ql, training=True, aux=aux)
    loss = O.cross_entropy_sum(logits, y)
    grads = torch.autograd.grad(loss, [W[k] for k in names], allow_unused=True)
    assert rel_err(logits.detach().numpy(), g["train_logits"]) < 2e-5
    assert abs(float(loss.detach()) - float(g["train_loss"])) < 1e-4 * max(1.0, abs(float(g["train_loss"])))
    assert rel_err(aux["bn"]["running_mean"].numpy(), g["bn_running_mean_after"]) < 1e-5
    assert rel_err(aux["bn"]["running_var"].numpy(), g["bn_running_var_after"]) < 1e-5
    # the reference keeps the carried state in q_len-sorted order (film_attn_pt_stem.py:150,160)
    perm = ql.sort(0, descending=True)[1]
    if "film_hidden_h_after" in g:          # the bag-of-words encoder carries no state
        assert rel_err(aux["hidden"][0].detach()[perm].numpy(), g["film_hidden_h_after"][0]) < 2e-5
        assert rel_err(aux["hidden"][1].detach()[perm].numpy(), g["film_hidden_c_after"][0]) < 2e-5
    checked = 0
    for k, gr in zip(names, grads):
        ref = g["grad/" + k]
        got = np.zeros_like(ref) if gr is None else gr.numpy()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 5e-4 * scale + 1e-6, k
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("case", QV_CASES)
def test_training_trajectory(case):
    """3 steps of CE(sum) -> clip 1.0 -> Adam restating eval/q_and_v_eval.py:124-139."""
    g = load_golden(case)
    model = model_of_case(case)
    W = weights_from(g, "w0")
    v, q, vl, ql, y = _inputs(g)
    # replay the capture sequence: one train-mode forward first (advances BN running stats)
    aux = {}
    with torch.no_grad():
        O.FORWARDS[model](W, v, q, vl, ql, training=True, aux=aux)
    for key in ("running_mean", "running_var", "num_batches_tracked"):
        W["bn_init." + key] = aux["bn"][key]
    adam = O.AdamState([k for k in W])
    losses = []
    for _ in range(len(g["traj_losses"])):
        loss, _, _ = O.train_step(model, W, v, q, vl, ql, y, adam, float(g["traj_lr"]),
                                  frozen_prefixes=FROZEN_CPU)
        losses.append(loss)
    assert np.allclose(losses, g["traj_losses"], rtol=2e-4, atol=1e-4), (losses, g["traj_losses"])
    with torch.no_grad():
        logits = O.FORWARDS[model](W, v, q, vl, ql, training=False)
    assert rel_err(logits.numpy(), g["traj_final_eval_logits"]) < 5e-4
    for k, ref in weights_from(g, "w_final").items():
        if k.endswith("num_batches_tracked"):
            assert int(W[k]) == int(ref)
            continue
        # Adam normalises each element's step to ~lr, so an element whose gradient is pure
        # rounding noise may move by O(lr) in either direction: bound the worst element by the
        # total possible travel and require the bulk of the tensor to agree tightly.
        d = np.abs(W[k].detach().numpy() - ref.numpy())
        travel = float(g["traj_lr"]) * len(g["traj_losses"])
        assert d.max() <= 0.5 * travel + 1e-7, k
        if d.size >= 32 and not k.startswith("fc_hidden_attn"):  # its gradient is ~0
            assert np.quantile(d, 0.9) <= 2e-2 * travel + 1e-7, k


def test_obj_detect_cnn():
    g = load_golden("objdet_f16")
    W = weights_from(g, "w")
    with torch.no_grad():
        y = O.obj_detect_cnn(torch.from_numpy(g["x"]), W)
    assert y.shape == g["y"].shape
    assert rel_err(y.numpy(), g["y"]) < 2e-5


def test_q_only_lstm():
    g = load_golden("qonly_small")
    W = weights_from(g, "w")
    with torch.no_grad():
        logits, _ = O.q_only_lstm_forward(W, torch.from_numpy(g["q"]), torch.from_numpy(g["q_lens"]),
                                          torch.from_numpy(g["h0"][0]), torch.from_numpy(g["c0"][0]))
    assert rel_err(logits.numpy(), g["logits"]) < 2e-5


def test_sort_batch_matches_reference_rule():
    v = torch.arange(4 * 2).float().view(4, 2)
    q = torch.arange(4).view(4, 1)
    vl = torch.tensor([3, 9, 5, 9])
    ql = torch.tensor([1, 2, 3, 4])
    ys = torch.tensor([10, 11, 12, 13])
    v2, q2, vl2, ql2, y2, perm = O.sort_batch(v, q, vl, ql, ys)
    assert vl2.tolist() == [9, 9, 5, 3]
    assert sorted(y2.tolist()[:2]) == [11, 13] and y2.tolist()[2:] == [12, 10]
    assert O.ct_batch_sizes(vl2, 10) == [4, 4, 4, 3, 3, 2, 2, 2, 2]


def test_video_only_cnn3d_features():
    """conv/pool/BN3d trunk of VideoOnlyCNN3D (models/v_only_cnn3d.py:59-72) vs the reference golden"""
    g = load_golden("cnn3d_small")
    W = {k: v.float() for k, v in weights_from(g, "w").items()}
    with torch.no_grad():
        f = O.video_only_cnn3d_features(W, torch.from_numpy(g["x"]), training=False)
    assert f.shape == g["conv_features"].shape
    assert rel_err(f.numpy(), g["conv_features"]) < 2e-5


# ---- MACNetwork (models/mac.py; `--model mac` of eval/q_and_v_eval.py) --------------------------
from helpers import MAC_CASES, mac_case


def _mac_kw(cfg):
    return dict(max_num_frames=cfg["max_num_frames"], self_attention=cfg["self_attention"],
                memory_gate=cfg["memory_gate"])


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_forward_and_gradients(case):
    g, cfg, W, (v, q, vl, ql, y), masks = mac_case(case)
    with torch.no_grad():
        ev = O.mac_forward(W, v, q, vl, ql, cfg["max_step"], **_mac_kw(cfg))
    assert rel_err(ev.numpy(), g["eval_logits"]) < 2e-5
    for tag, mk in (("train", None), ("drop", masks)):
        Wg = {k: t.clone().requires_grad_(True) for k, t in W.items()}
        logits = O.mac_forward(Wg, v, q, vl, ql, cfg["max_step"], masks=mk, **_mac_kw(cfg))
        loss = O.cross_entropy_sum(logits, y)
        names = list(Wg)
        grads = torch.autograd.grad(loss, [Wg[k] for k in names], allow_unused=True)
        assert rel_err(logits.detach().numpy(), g[tag + "_logits"]) < 2e-5
        assert abs(float(loss.detach()) - float(g[tag + "_loss"])) < 1e-4 * max(1.0, abs(float(g[tag + "_loss"])))
        checked = 0
        for k, gr in zip(names, grads):
            if tag + "_grad/" + k not in g:
                continue
            ref = g[tag + "_grad/" + k]
            got = np.zeros_like(ref) if gr is None else gr.numpy()
            assert np.abs(got - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-6, (tag, k)
            checked += 1
        assert checked >= 30


def test_mac_question_vector_stays_in_sorted_order():
    """mac.py:221 leaves `h` in question-length-sorted order while lstm_out is unsorted (:217-218)."""
    g, cfg, W, (v, q, vl, ql, y), _ = mac_case("mac_plain")
    assert list(ql) != sorted(ql, reverse=True)
    ctx, h = O.mac_question(W, q, ql)
    perm = ql.sort(0, descending=True)[1]
    ctx_s, h_s = O.mac_question(W, q[perm], ql[perm])      # already sorted: identity permutation
    assert torch.allclose(h, h_s, atol=1e-6) and torch.allclose(ctx[perm], ctx_s, atol=1e-6)


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_training_trajectory(case):
    """clamp hooks (eval/q_and_v_eval.py:348-351) -> clip 1.0 -> Adam, 3 steps."""
    g, cfg, W, (v, q, vl, ql, y), _ = mac_case(case)
    adam = O.AdamState(list(W))
    losses = []
    for _ in range(len(g["traj_losses"])):
        loss, _, _ = O.mac_train_step(W, v, q, vl, ql, y, adam, float(g["traj_lr"]), cfg["max_step"], **_mac_kw(cfg))
        losses.append(loss)
    assert np.allclose(losses, g["traj_losses"], rtol=2e-4, atol=1e-4)
    with torch.no_grad():
        logits = O.mac_forward(W, v, q, vl, ql, cfg["max_step"], **_mac_kw(cfg))
    assert rel_err(logits.numpy(), g["traj_final_eval_logits"]) < 5e-4
    travel = float(g["traj_lr"]) * len(g["traj_losses"])
    for k, ref in weights_from(g, "w_final").items():
        d = np.abs(W[k].numpy() - ref.numpy())
        assert d.max() <= 0.5 * travel + 1e-7, k
        if d.size >= 32:
            assert np.quantile(d, 0.9) <= 2e-2 * travel + 1e-7, k
