"""GPU parity: the wide packed LSTM kernels and MACNetwork (HIP path) vs the reference goldens / oracle."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import MAC_CASES, mac_case, rel_err, LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rocblas_step", "cabi_step", "cabi_chain"])
def mac_core_impl(request, monkeypatch):
    """Every MAC test runs with all three forms of the reasoning steps: the op-by-op node on torch / rocBLAS GEMMs, one C-ABI
    node per step (vnqa_mac_core_fwd / _bwd) and — the default — all steps as one node (vnqa_mac_chain_fwd / _bwd)."""
    from videonavqa_amd import ops
    import torch_partners
    monkeypatch.setattr(ops, "MAC_CHAIN", request.param == "cabi_chain")
    if request.param == "rocblas_step":       # the op-by-op torch / rocBLAS form of the step lives in tests/torch_partners.py
        monkeypatch.setattr(ops, "mac_core", torch_partners.mac_core_torch)
    return request.param


def _packed_reference(lstm, x_sorted, lens):
    """torch.nn.LSTM on a packed batch (CPU fp32) -> padded output [B,Lmax,dirs*H] with grads enabled."""
    packed = nn.utils.rnn.pack_padded_sequence(x_sorted, lens, batch_first=True)
    out, _ = lstm(packed)
    return nn.utils.rnn.pad_packed_sequence(out, batch_first=True)[0]


@pytest.mark.parametrize("B,E,H,lens", [(5, 12, 16, [9, 7, 7, 3, 1]), (11, 20, 48, [6, 6, 5, 5, 5, 4, 3, 3, 2, 1, 1]),
                                        (3, 300, 512, [13, 9, 4])])
def test_lstm_wide_matches_torch_packed_bidirectional(B, E, H, lens):
    """ops.lstm_wide (both directions) vs nn.LSTM(bidirectional) on pack_padded_sequence input:
    outputs and gradients w.r.t. input projection weights, recurrent weights and inputs."""
    from videonavqa_amd import ops
    torch.manual_seed(3)
    lstm = nn.LSTM(E, H, batch_first=True, bidirectional=True)
    Lmax = lens[0]
    x = torch.randn(B, Lmax, E)
    for b, l in enumerate(lens):
        x[b, l:] = 0
    x.requires_grad_(True)
    ref = _packed_reference(lstm, x, torch.tensor(lens))
    gout = torch.randn_like(ref)
    for b, l in enumerate(lens):
        gout[b, l:] = 0          # padded positions carry no gradient upstream either
    ref.backward(gout)

    dev = torch.device("cuda")
    xd = x.detach().to(dev).requires_grad_(True)
    bsz = ops.packed_batch_sizes(lens)
    outs, params = [], []
    for sfx, rev in (("", False), ("_reverse", True)):
        w_ih = getattr(lstm, "weight_ih_l0" + sfx).detach().to(dev).requires_grad_(True)
        w_hh = getattr(lstm, "weight_hh_l0" + sfx).detach().to(dev).requires_grad_(True)
        bias = (getattr(lstm, "bias_ih_l0" + sfx) + getattr(lstm, "bias_hh_l0" + sfx)).detach().to(dev).requires_grad_(True)
        xg = torch.nn.functional.linear(xd, w_ih, bias).transpose(0, 1).contiguous()
        outs.append(ops.lstm_wide(xg, w_hh, bsz, rev))
        params.append((sfx, w_ih, w_hh, bias))
    got = torch.cat(outs, 2).transpose(0, 1)
    got.backward(gout.to(dev))
    assert rel_err(got.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
    assert rel_err(xd.grad.cpu().numpy(), x.grad.numpy()) < 2e-4
    for sfx, w_ih, w_hh, bias in params:
        assert rel_err(w_ih.grad.cpu().numpy(), getattr(lstm, "weight_ih_l0" + sfx).grad.numpy()) < 2e-4, sfx
        assert rel_err(w_hh.grad.cpu().numpy(), getattr(lstm, "weight_hh_l0" + sfx).grad.numpy()) < 2e-4, sfx
        assert rel_err(bias.grad.cpu().numpy(), getattr(lstm, "bias_ih_l0" + sfx).grad.numpy()) < 2e-4, sfx


@pytest.mark.parametrize("B,H,lens", [(5, 16, [9, 7, 7, 3, 1]), (11, 48, [6, 6, 5, 5, 5, 4, 3, 3, 2, 1, 1]), (3, 512, [13, 9, 4]),
                                      (8, 1536, [7, 7, 6, 4, 4, 2, 1, 1])])
def test_lstm_wide_bidir_is_two_single_direction_calls(B, H, lens):
    """ops.lstm_wide_bidir (chain position i of both directions in one launch) vs two ops.lstm_wide nodes: outputs and every
    gradient bit-identical (same arithmetic, only the launches are shared)."""
    from videonavqa_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(5)
    T = lens[0]
    bsz = ops.packed_batch_sizes(lens)
    xg = [(torch.randn(T, B, 4 * H, device=dev) * 0.3).requires_grad_(True) for _ in range(4)]
    w = [(torch.randn(4 * H, H, device=dev) / H ** 0.5).requires_grad_(True) for _ in range(2)]
    w2 = [t.detach().clone().requires_grad_(True) for t in w]
    with torch.no_grad():
        xg[2].copy_(xg[0]); xg[3].copy_(xg[1])
    valid = torch.zeros(T, B, 1, device=dev)
    for b, l in enumerate(lens):
        valid[:l, b] = 1
    gf, gr = torch.randn(T, B, H, device=dev) * valid, torch.randn(T, B, H, device=dev) * valid
    a_f, a_r = ops.lstm_wide(xg[0], w[0], bsz, False), ops.lstm_wide(xg[1], w[1], bsz, True)
    ((a_f * gf).sum() + (a_r * gr).sum()).backward()
    b_f, b_r = ops.lstm_wide_bidir(xg[2], xg[3], w2[0], w2[1], bsz)
    ((b_f * gf).sum() + (b_r * gr).sum()).backward()
    assert torch.equal(a_f, b_f) and torch.equal(a_r, b_r)
    assert torch.equal(xg[0].grad, xg[2].grad) and torch.equal(xg[1].grad, xg[3].grad)
    assert torch.equal(w[0].grad, w2[0].grad) and torch.equal(w[1].grad, w2[1].grad)
    assert float(a_f.detach().abs().sum()) > 0 and float(xg[1].grad.abs().sum()) > 0


@pytest.mark.parametrize("n,h,w,cin,cout", [(5, 14, 14, 64, 256), (3, 10, 12, 64, 128), (2, 28, 28, 128, 512)])
def test_conv_elu_epilogue_matches_elu_after_conv(n, h, w, cin, cout):
    """ops.conv(relu=2): ELU in the conv epilogue (igemm and patch-stationary tiles) vs F.elu applied to the plain conv's
    output, forward and all gradients, and vs an fp32 torch reference of conv -> ELU."""
    from videonavqa_amd import ops
    from helpers import LOW_DTYPE
    dev = torch.device("cuda")
    torch.manual_seed(n + h + cout)
    x = torch.zeros(n, h + 2, w + 2, cin, device=dev)
    x[:, 1:-1, 1:-1] = torch.randn(n, h, w, cin, device=dev)
    wgt = (torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5 * 1.5)
    b = torch.randn(cout, device=dev) * 0.2
    g = torch.zeros(n, h + 2, w + 2, cout, device=dev)
    g[:, 1:-1, 1:-1] = torch.randn(n, h, w, cout, device=dev)
    outs = []
    for fused in (True, False):
        xx = x.to(LOW_DTYPE).requires_grad_(True)
        ww, bb = wgt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv(xx, ww, bb, relu=2) if fused else torch.nn.functional.elu(ops.conv(xx, ww, bb, relu=False))
        y.backward(g.to(LOW_DTYPE))
        outs.append((y.detach().float(), xx.grad.float(), ww.grad, bb.grad))
    xr = x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).to(LOW_DTYPE).float().requires_grad_(True)
    wr, br = wgt.to(LOW_DTYPE).float().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.elu(torch.nn.functional.conv2d(xr, wr, br, padding=1))
    yr.backward(g[:, 1:-1, 1:-1].permute(0, 3, 1, 2).to(LOW_DTYPE).float())
    y_f, dx_f, dw_f, db_f = outs[0]
    assert float(y_f[:, 0].abs().max()) == 0 and float(y_f[:, :, -1].abs().max()) == 0          # halo stays zero
    assert float((y_f < 0).float().mean()) > 0.2                                                  # the negative branch is exercised
    ref = yr.detach().permute(0, 2, 3, 1)
    assert rel_err(y_f[:, 1:-1, 1:-1].cpu().numpy(), ref.cpu().numpy()) < 6e-3
    assert rel_err(outs[1][0][:, 1:-1, 1:-1].cpu().numpy(), ref.cpu().numpy()) < 8e-3          # the two-rounding form it replaces
    assert rel_err(dx_f[:, 1:-1, 1:-1].cpu().numpy(), xr.grad.permute(0, 2, 3, 1).cpu().numpy()) < 1.5e-2
    assert rel_err(dw_f.cpu().numpy(), wr.grad.cpu().numpy()) < 1.5e-2
    assert rel_err(db_f.cpu().numpy(), br.grad.cpu().numpy()) < 1.5e-2
    for a, c in zip(outs[0][1:], outs[1][1:]):
        assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1.5e-2


def test_lstm_wide_rejects_bad_batch_sizes():
    from videonavqa_amd import kernels as K
    from videonavqa_amd._lib import VnqaError
    xg = torch.zeros(3, 2, 64, device="cuda")
    w = torch.zeros(64, 16, device="cuda")
    with pytest.raises(VnqaError):
        K.lstm_wide_fwd(xg, w, [1, 2, 1])      # increasing
    with pytest.raises(VnqaError):
        K.lstm_wide_fwd(xg, w, [3, 2, 1])      # larger than the batch


def _product(case, precision):
    import videonavqa_amd.models as M
    g, cfg, W, inputs, masks = mac_case(case)
    model = M.MACNetwork(precision=precision, **cfg).cuda()
    model.load_reference_tensors(W)
    dev = torch.device("cuda")
    v, q, vl, ql, y = inputs
    return model, g, (v.to(dev), q.to(dev), vl, ql, y.to(dev)), masks


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_eval_logits_fp32(case):
    model, g, (v, q, vl, ql, y), _ = _product(case, "fp32")
    model.eval()
    with torch.no_grad():
        out = model(v, q, vl, ql)
    assert rel_err(out.cpu().numpy(), g["eval_logits"]) < 1e-3
    assert rel_err(out.cpu().numpy(), g["eval_logits"]) < 5e-5   # exact-f32 path: far inside the north-star bound


@pytest.mark.parametrize("case", MAC_CASES)
@pytest.mark.parametrize("tag", ["train", "drop"])
def test_mac_train_forward_backward_fp32(case, tag):
    """Train-mode logits, loss and every parameter gradient vs the reference; `drop` injects the variational
    dropout masks the golden run used (one pair per frame, concatenated in packed-image order)."""
    model, g, (v, q, vl, ql, y), masks = _product(case, "fp32")
    model.train()
    if tag == "drop":
        model.dropout_masks = (torch.cat([m[0] for m in masks]).cuda(), torch.cat([m[1] for m in masks]).cuda())
    else:
        model.mac.dropout = 0.0
    logits = model(v, q, vl, ql)
    loss = nn.functional.cross_entropy(logits, y, reduction="sum")
    loss.backward()
    assert rel_err(logits.detach().cpu().numpy(), g[tag + "_logits"]) < 5e-5
    assert abs(float(loss.detach()) - float(g[tag + "_loss"])) < 1e-4 * max(1.0, abs(float(g[tag + "_loss"])))
    checked = 0
    for k, p in model.named_parameters():
        key = tag + "_grad/" + k
        if key not in g:
            continue
        ref = g[key]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 2e-6, (k, np.abs(got - ref).max(), np.abs(ref).max())
        checked += 1
    assert checked >= 30


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_chain_node_is_the_per_step_nodes(case, mac_core_impl, monkeypatch):
    """All steps as one autograd node (ops.MacChainFn) vs one node per step (ops.MacCoreFn): logits and every parameter gradient
    bit-identical — the same kernels in the same order, only the loop over steps moved into C++."""
    if mac_core_impl != "cabi_chain":
        pytest.skip("one comparison is enough")
    res = []
    for chain in ("1", "0"):
        monkeypatch.setattr(__import__("videonavqa_amd.ops", fromlist=["ops"]), "MAC_CHAIN", chain == "1")
        model, g, (v, q, vl, ql, y), masks = _product(case, LOW)
        model.train()
        model.dropout_masks = (torch.cat([m[0] for m in masks]).cuda(), torch.cat([m[1] for m in masks]).cuda())
        logits = model(v, q, vl, ql)
        nn.functional.cross_entropy(logits, y, reduction="sum").backward()
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) >= 30
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_bf16_close_to_reference(case):
    model, g, (v, q, vl, ql, y), _ = _product(case, LOW)
    model.eval()
    with torch.no_grad():
        out = model(v, q, vl, ql)
    assert rel_err(out.cpu().numpy(), g["eval_logits"]) < 3e-2


def test_mac_random_dropout_masks_are_variational():
    """Without injected masks, train mode draws ONE Bernoulli(1-p)/(1-p) mask pair per packed image and reuses it at
    every reasoning step (mac.py:137-153)."""
    model, g, (v, q, vl, ql, y), _ = _product("mac_plain", "fp32")
    model.train()
    torch.manual_seed(0)
    m = model._masks(64, v.device)
    vals = torch.unique(torch.cat([m[0].flatten(), m[1].flatten()])).cpu().numpy()
    assert all(min(abs(float(x)), abs(float(x) - 1 / 0.85)) < 1e-5 for x in vals)
    assert 0.7 < float((m[0] > 0).float().mean()) < 0.95
    a = model(v, q, vl, ql)
    b = model(v, q, vl, ql)
    assert not torch.allclose(a, b)          # fresh masks every forward
    model.eval()
    assert model._masks(4, v.device) is None


@pytest.mark.parametrize("case", MAC_CASES)
def test_mac_trainer_trajectory(case):
    """Trainer (flat buffers, gradient clamp [-1,1] -> fused clip + Adam) vs the reference's 3-step trajectory."""
    from videonavqa_amd.models.common import FrameLayout, NativeFeatures
    from videonavqa_amd import kernels as K
    from videonavqa_amd.train import Trainer
    model, g, (v, q, vl, ql, y), _ = _product(case, "fp32")
    model.mac.dropout = 0.0
    trainer = Trainer(model, stem=None, lr=float(g["traj_lr"]), clip=1.0, feature_channels=512)
    lay = FrameLayout(vl, v.shape[-1], v.device)
    native = NativeFeatures(K.feat_to_nhwc(v, lay.img_of, lay.n_img, torch.float32), lay, 512, v.shape[2], v.shape[3])
    trainer.extract_features = lambda clip, v_lens_cpu, slot=0: (native, vl, torch.arange(len(vl)))
    losses = []
    for _ in range(len(g["traj_losses"])):
        loss, _ = trainer.step(v, q, vl, ql, y)
        losses.append(float(loss))
    assert np.allclose(losses, g["traj_losses"], rtol=1e-3, atol=1e-3), (losses, g["traj_losses"])
    model.eval()
    with torch.no_grad():
        out = model(v, q, vl, ql)
    assert rel_err(out.cpu().numpy(), g["traj_final_eval_logits"]) < 2e-3
    travel = float(g["traj_lr"]) * len(g["traj_losses"])
    sd = model.state_dict()
    for k in sd:
        d = np.abs(sd[k].cpu().numpy() - g["w_final/" + k].astype(np.float32))
        assert d.max() <= 0.7 * travel + 1e-7, k
        if d.size >= 32:
            assert np.quantile(d, 0.9) <= 5e-2 * travel + 1e-7, k


def test_mac_full_size_batch_independence_and_training():
    """Default CLI size (dim 512, 12 steps, bs 8, 35 frames of 14x14x512, bf16).  MACNetwork has no batch
    statistics, so with question lengths already sorted (which makes the upstream `h` ordering quirk the identity)
    every sample's logits must not depend on what else is in the minibatch — a size-independent property that
    exercises the packed image list, the ragged packed LSTMs and the batched reasoning steps at full size.
    Then three optimisation steps through the Trainer must stay finite and reduce the loss on a fixed batch."""
    import videonavqa_amd.models as M
    from videonavqa_amd.train import Trainer
    from videonavqa_amd.models.common import FrameLayout, NativeFeatures
    from videonavqa_amd import kernels as K
    torch.manual_seed(5)
    dev = torch.device("cuda")
    B, T = 8, 35
    model = M.MACNetwork(n_vocab=134, dim=512, embed_hidden=128, classes=70, precision=LOW).to(dev)
    v = torch.rand(B, 512, 14, 14, T, device=dev)
    v_lens = torch.tensor([35, 35, 30, 22, 22, 9, 4, 3])
    q_lens = torch.tensor([25, 19, 19, 12, 9, 7, 6, 5])
    q = torch.randint(1, 134, (B, 56), device=dev) * (torch.arange(56, device=dev)[None] < q_lens.to(dev)[:, None])
    model.eval()
    with torch.no_grad():
        full = model(v, q, v_lens, q_lens).float()
        for b in (0, 3, 7):
            solo = model(v[b:b + 1], q[b:b + 1], v_lens[b:b + 1], q_lens[b:b + 1]).float()
            assert torch.isfinite(solo).all()
            assert rel_err(solo.cpu().numpy(), full[b:b + 1].cpu().numpy()) < 2e-2, b
    y = torch.randint(0, 70, (B,), device=dev)
    trainer = Trainer(model, stem=None, lr=1e-4, clip=1.0, feature_channels=512)
    lay = FrameLayout(v_lens, T, dev)
    native = NativeFeatures(K.feat_to_nhwc(v, lay.img_of, lay.n_img, LOW_DTYPE), lay, 512, 14, 14)
    trainer.extract_features = lambda clip, v_lens_cpu, slot=0: (native, v_lens, torch.arange(B))
    model.mac.dropout = 0.0
    losses = [float(trainer.step(v, q, v_lens, q_lens, y)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
