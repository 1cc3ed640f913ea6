"""GPU: 3-D convolution on the MFMA igemm / wgrad kernels (27-tap table over padded NDHWC) and the
VideoOnlyCNN3D drop-in (config 2 bring-up)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import load_golden, rel_err, weights_from, LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


def _q(t, dt):
    return t.to(dt).float()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("dt", [torch.float32, LOW_DTYPE])
@pytest.mark.parametrize("cfg", [(2, 4, 10, 12, 64, 64), (1, 6, 8, 8, 128, 192), (3, 3, 14, 6, 64, 128), (2, 4, 6, 10, 128, 128)])
def test_conv3d_fwd_dgrad_wgrad_vs_torch(dt, cfg):
    from videonavqa_amd import ops
    N, D, H, W, Cin, Cout = cfg
    g = torch.Generator(device="cpu").manual_seed(sum(cfg))
    x = torch.randn(N, Cin, D, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    dy = torch.randn(N, Cout, D, H, W, generator=g).cuda()
    xr = _q(x, dt).detach().clone().requires_grad_(True)
    wr = _q(w, dt).detach().clone().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    ref = F.relu(F.conv3d(xr, wr, br, padding=1))
    ref.backward(_q(dy, dt))
    xp = x.detach().clone().requires_grad_(True)
    wp = w.detach().clone().requires_grad_(True)
    bp = b.detach().clone().requires_grad_(True)
    y = ops.conv3d(ops.ncdhw_to_ndhwc_padded(xp, dt, c_pad=Cin), wp, bp, relu=True, need_dx=True)
    got = ops.ndhwc_padded_to_ncdhw(y, Cout)
    got.backward(dy)
    tol = 3e-5 if dt == torch.float32 else 1e-2
    assert _rel(got.detach(), ref.detach()) < tol
    assert float(y.detach()[:, 0].abs().max()) == 0 and float(y.detach()[:, :, :, -1].abs().max()) == 0     # zero halo kept
    gt = 1e-4 if dt == torch.float32 else 3e-2
    assert _rel(xp.grad, xr.grad) < gt, _rel(xp.grad, xr.grad)
    assert _rel(wp.grad, wr.grad) < gt, _rel(wp.grad, wr.grad)
    assert _rel(bp.grad, br.grad) < gt


@pytest.mark.parametrize("precision", ["fp32", LOW])
def test_video_only_cnn3d_features_vs_reference_golden(precision):
    from videonavqa_amd.models import VideoOnlyCNN3D
    g = load_golden("cnn3d_small")
    m = VideoOnlyCNN3D(5, fc6_in_features=128, precision=precision)
    m.load_state_dict({k: v.float() for k, v in weights_from(g, "w").items()}, strict=False)
    m = m.cuda().eval()
    with torch.no_grad():
        f = m.features(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert f.shape == g["conv_features"].shape
    assert rel_err(f, g["conv_features"]) < (2e-4 if precision == "fp32" else 5e-2), rel_err(f, g["conv_features"])


def test_video_only_cnn3d_train_step_vs_oracle():
    """full forward + backward (train-mode BN) against the oracle's autograd on the same weights"""
    from oracle import vnqa_oracle as O
    from videonavqa_amd.models import VideoOnlyCNN3D
    torch.manual_seed(3)
    B, D, H, W = 4, 16, 32, 32
    m = VideoOnlyCNN3D(7, fc6_in_features=128, precision="fp32")
    Wd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(B, 3, D, H, W)
    y = torch.randint(0, 7, (B,))
    names = [k for k, v in Wd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        Wd[k].requires_grad_(True)
    ref = O.video_only_cnn3d_forward(Wd, x, training=True)
    loss_ref = F.cross_entropy(ref, y, reduction="sum")
    gref = dict(zip(names, torch.autograd.grad(loss_ref, [Wd[k] for k in names])))
    m = m.cuda().train()
    out = m(x.cuda())
    loss = F.cross_entropy(out, y.cuda(), reduction="sum")
    loss.backward()
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-3
    for k, p in m.named_parameters():
        # conv biases in front of a train-mode BatchNorm have a mathematically zero gradient (pure rounding
        # noise on both sides): absolute floor next to the relative bound
        err = float((p.grad.cpu() - gref[k]).abs().max())
        assert err <= 5e-3 * float(gref[k].abs().max()) + 1e-4, (k, err)


# ---- csrc/cnn3d.hip: the fused config-2 path -----------------------------------------------------------------------------------
def test_c3d_conv1_fwd_bwd_vs_torch():
    """bn_input (batch statistics) -> conv1 -> ReLU -> MaxPool3d(1,2,2) from the fp32 clip in one kernel, and its one-GEMM backward
    (conv1 weight / bias gradients AND bn_input's gamma / beta gradients) against torch autograd of the same graph."""
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    N, D, H, W = 2, 5, 32, 48
    x = torch.rand(N, 3, D, H, W, generator=g).cuda()
    w = (torch.randn(64, 3, 3, 3, 3, generator=g) / 9.0).cuda().requires_grad_(True)
    b = (torch.randn(64, generator=g) * 0.1).cuda().requires_grad_(True)
    gam = (torch.rand(3, generator=g) + 0.5).cuda().requires_grad_(True)
    bet = (torch.randn(3, generator=g) * 0.3).cuda().requires_grad_(True)
    ref = F.max_pool3d(F.relu(F.conv3d(F.batch_norm(x, None, None, gam, bet, True, 0.1, 1e-5), w, b, padding=1)), (1, 2, 2))
    dp = torch.randn(ref.shape, generator=g).cuda()
    ref.backward(dp)
    mean, rstd = K.bn_finalize(K.c3d_stats_ncdhw(x), N * D * H * W, 1e-5)
    assert _rel(mean, x.mean((0, 2, 3, 4))) < 1e-5 and _rel(rstd, torch.rsqrt(x.var((0, 2, 3, 4), unbiased=False) + 1e-5)) < 1e-5
    p, idx, part = K.c3d_conv1_fwd(x, w.detach(), b.detach(), mean, rstd, gam.detach(), bet.detach(), LOW_DTYPE)
    got = p.float().permute(0, 4, 1, 2, 3)
    assert _rel(got, ref.detach()) < 2e-2, _rel(got, ref.detach())
    s = part.sum(0)
    assert _rel(s[:, 0], p.float().sum((0, 1, 2, 3))) < 1e-4 and _rel(s[:, 1], (p.float() ** 2).sum((0, 1, 2, 3))) < 1e-4
    dpl = dp.permute(0, 2, 3, 4, 1).contiguous().to(LOW_DTYPE)
    dw, db, dg, dbt = K.c3d_conv1_bwd(x, w.detach(), mean, rstd, gam.detach(), bet.detach(), dpl, idx)
    # exact check: torch's conv backward on the SAME operands the kernel sees — the 16-bit normalised input, and dY routed by
    # the forward kernel's own arg-max bytes (a near-tie resolved differently from the fp32 graph moves a whole dY entry)
    xh = ((x - mean.view(1, 3, 1, 1, 1)) * rstd.view(1, 3, 1, 1, 1)).to(LOW_DTYPE).float()
    xb = (xh * gam.detach().view(1, 3, 1, 1, 1) + bet.detach().view(1, 3, 1, 1, 1)).requires_grad_(True)
    dY = torch.zeros(N, D, H, W, 64, device="cuda")
    for sub in range(4):
        dY[:, :, (sub >> 1)::2, (sub & 1)::2] = torch.where(idx.long() == sub, dpl.float(), torch.zeros_like(dpl.float()))
    dYc = dY.permute(0, 4, 1, 2, 3).contiguous()
    wr = w.detach().clone().requires_grad_(True)
    F.conv3d(xb, wr, None, padding=1).backward(dYc)
    for name, a, r in (("dw", dw, wr.grad), ("db", db, dYc.sum((0, 2, 3, 4))), ("dgamma", dg, (xb.grad * xh).sum((0, 2, 3, 4))),
                       ("dbeta", dbt, xb.grad.sum((0, 2, 3, 4)))):
        assert _rel(a, r) < 1e-4, (name, _rel(a, r))
    # and against the fp32 graph's own gradients: sums of ~1e5 random-sign terms, so routing flips show up at the few-% level
    for name, a, r in (("dw", dw, w.grad), ("db", db, b.grad), ("dgamma", dg, gam.grad), ("dbeta", dbt, bet.grad)):
        assert _rel(a, r) < 0.12, (name, _rel(a, r))


@pytest.mark.parametrize("shape", [(2, 8, 12, 16, 64), (1, 4, 14, 14, 128), (3, 5, 9, 8, 64)])
def test_pool444_fwd_bwd_vs_torch(shape):
    from videonavqa_amd import kernels as K
    N, D, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.zeros(N, D + 2, H + 2, W + 2, C, dtype=LOW_DTYPE, device="cuda")
    y[:, 1:-1, 1:-1, 1:-1] = torch.relu(torch.randn(N, D, H, W, C, generator=g)).cuda().to(LOW_DTYPE)
    yr = y[:, 1:-1, 1:-1, 1:-1].float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    ref = F.max_pool3d(yr, 4)
    p, idx, part = K.pool444_fwd(y)
    assert torch.equal(p.float().permute(0, 4, 1, 2, 3), ref.detach())
    assert _rel(part.sum(0)[:, 0], p.float().sum((0, 1, 2, 3))) < 1e-5
    dp = torch.randn(p.shape, generator=g).cuda().to(LOW_DTYPE)
    ref.backward(dp.float().permute(0, 4, 1, 2, 3))
    dy = torch.full_like(y, 7.0)
    dy[:, 0] = 0; dy[:, -1] = 0; dy[:, :, 0] = 0; dy[:, :, -1] = 0; dy[:, :, :, 0] = 0; dy[:, :, :, -1] = 0
    K.pool444_bwd(dp, idx, dy)
    want = yr.grad * (yr.detach() > 0)                   # the pool's arg-max bytes carry the ReLU mask of the conv output
    assert torch.equal(dy[:, 1:-1, 1:-1, 1:-1].float().permute(0, 4, 1, 2, 3), want)
    assert float(dy[:, 0].abs().max()) == 0 and float(dy[:, :, :, -1].abs().max()) == 0


@pytest.mark.parametrize("kind", ["rows_f32", "padded_low", "nc_flat"])
def test_bn_rows_fwd_bwd_vs_torch(kind):
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    N, D, H, W, C = 3, 2, 5, 4, 64
    R = N * D * H * W
    gam = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    bet = torch.randn(C, generator=g).cuda().requires_grad_(True)
    xdt = torch.float32 if kind == "rows_f32" else LOW_DTYPE
    x = (torch.randn(R, C, generator=g) * 2 + 1).cuda().to(xdt)
    xr = x.float().requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm2, rv2 = rm.clone(), rv.clone()
    ref = F.batch_norm(xr, rm, rv, gam, bet, True, 0.1, 1e-5)
    mean, rstd = K.bn_finalize(K.c3d_stats_rows(x), R, 1e-5, 0.1, rm2, rv2)
    assert _rel(rm2, rm) < 1e-5 and _rel(rv2, rv) < 1e-5
    dy = torch.randn(R, C, generator=g).cuda()
    ref.backward(dy)
    if kind == "rows_f32":
        out = torch.empty(R, C, device="cuda")
        view, dyv, dyt, dxdt = K.view_dense(1, 1, 1, C), K.view_dense(1, 1, 1, C), dy, torch.float32
        back = lambda o: o
    elif kind == "padded_low":
        out = torch.zeros(N, D + 2, H + 2, W + 2, C, dtype=LOW_DTYPE, device="cuda")
        view = dyv = K.view_padded_ndhwc(D, H, W, C)
        dyt = torch.zeros_like(out)
        dyt[:, 1:-1, 1:-1, 1:-1] = dy.view(N, D, H, W, C).to(LOW_DTYPE)
        dxdt = LOW_DTYPE
        back = lambda o: o[:, 1:-1, 1:-1, 1:-1].reshape(R, C).float()
    else:
        out = torch.empty(N, C * D * H * W, device="cuda")
        view = dyv = K.view_nc_flat(D, H, W, C)
        dyt = dy.view(N, D, H, W, C).permute(0, 4, 1, 2, 3).reshape(N, -1).contiguous()
        dxdt = LOW_DTYPE
        back = lambda o: o.view(N, C, D, H, W).permute(0, 2, 3, 4, 1).reshape(R, C)
    K.bn_rows_apply(x, out, view, mean, rstd, gam.detach(), bet.detach())
    tol = 1e-5 if kind == "rows_f32" else 1e-2
    assert _rel(back(out), ref.detach()) < tol
    dx, dg, db = K.bn_rows_bwd(dyt, dyv, x, dxdt, mean, rstd, gam.detach())
    assert _rel(dx.float(), xr.grad) < (1e-4 if kind == "rows_f32" else 2e-2), _rel(dx.float(), xr.grad)
    assert _rel(dg, gam.grad) < (1e-4 if kind == "rows_f32" else 1e-2) and _rel(db, bet.grad) < (1e-4 if kind == "rows_f32" else 1e-2)


def test_video_only_cnn3d_fused_train_step_vs_oracle_low_precision(monkeypatch):
    """the fused 16-bit path (ops.Cnn3dFeaturesFn + classifier on vnqa_sgemm / BatchNorm kernels): logits and every gradient
    against (a) the oracle's fp32 autograd and (b) the generic 16-bit path (igemm / wgrad convs + stock BatchNorm3d / MaxPool3d,
    pinned to the reference golden) on the same weights.  Every kernel is pinned exactly above; this net — train-mode BatchNorm
    over 8 samples behind three ReLU / max-pool stages — amplifies 16-bit rounding (both 16-bit paths sit at cos 0.93-0.99 to
    the fp32 gradients), so (a) bounds direction and size and (b) shows the two 16-bit paths are the same computation
    (measured: cos to the oracle 0.883 - 1.000, cos between the two 16-bit paths 0.9715 - 1.000)."""
    from oracle import vnqa_oracle as O
    from videonavqa_amd.models import VideoOnlyCNN3D
    torch.manual_seed(3)
    B, D, H, W = 8, 16, 64, 64
    m = VideoOnlyCNN3D(7, fc6_in_features=128 * 2 * 2, precision=LOW)
    Wd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(B, 3, D, H, W)
    y = torch.randint(0, 7, (B,))
    names = [k for k, v in Wd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        Wd[k].requires_grad_(True)
    ref = O.video_only_cnn3d_forward(Wd, x, training=True)
    loss_ref = F.cross_entropy(ref, y, reduction="sum")
    gref = dict(zip(names, torch.autograd.grad(loss_ref, [Wd[k] for k in names])))
    m = m.cuda().train()
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    runs = {}
    for mode in ("fused", "generic"):
        m.force_generic = mode == "generic"
        m.load_state_dict(state0)
        m.zero_grad()
        assert m._fast_ok(x.cuda()) == (mode == "fused")
        out = m(x.cuda())
        F.cross_entropy(out, y.cuda(), reduction="sum").backward()
        runs[mode] = (out.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()},
                      {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "running" in k or "tracked" in k})
    cosr = lambda a, r: (float(torch.dot(a.flatten(), r.flatten()) / (a.norm() * r.norm() + 1e-12)), float(a.norm() / (r.norm() + 1e-12)))
    out_f, g_f, st_f = runs["fused"]
    out_g, g_g, st_g = runs["generic"]
    assert _rel(out_f, ref.detach()) < 0.12 and _rel(out_f, out_g) < 0.06, (_rel(out_f, ref.detach()), _rel(out_f, out_g))
    for k in g_f:
        assert bool(torch.isfinite(g_f[k]).all()), k
        if float(gref[k].norm()) < 1e-3:          # conv2 / conv3a bias: analytically ~0 in front of a train-mode BatchNorm
            continue
        c_o, r_o = cosr(g_f[k], gref[k])
        c_g, r_g = cosr(g_f[k], g_g[k])
        wide = k.startswith("bn_input")            # 3-element, near-cancelling sums
        print("%-16s vs oracle cos %.4f size %.3f   vs generic 16-bit path cos %.4f size %.3f" % (k, c_o, r_o, c_g, r_g))
        assert c_o > 0.8 and (0.6 if wide else 0.85) < r_o < (1.6 if wide else 1.2), (k, "vs oracle", c_o, r_o)
        assert c_g > 0.95 and (0.7 if wide else 0.9) < r_g < (1.4 if wide else 1.12), (k, "vs generic 16-bit path", c_g, r_g)
    for k in st_f:                                 # BatchNorm running statistics / num_batches_tracked advance alike
        assert _rel(st_f[k].float(), st_g[k].float()) < 2e-2, k
    m.eval()                                       # eval mode: the same kernels on the running statistics
    m.force_generic = False
    with torch.no_grad():
        ev = m(x.cuda())
    m.force_generic = True
    with torch.no_grad():
        ev_g = m(x.cuda())
    assert _rel(ev.cpu(), ev_g.cpu()) < 0.06


def test_video_only_cnn3d_config2_batch32_trains():
    """BASELINE config 2 at its own size: bs = 32, 16 x 3 x 112 x 112 clips, a few Adam steps on a fixed batch"""
    from videonavqa_amd.models import VideoOnlyCNN3D
    torch.manual_seed(0)
    m = VideoOnlyCNN3D(70, fc6_in_features=128 * 1 * 3 * 3, precision=LOW).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x = torch.rand(32, 3, 16, 112, 112, device="cuda")
    y = torch.randint(0, 70, (32,), device="cuda")
    losses = []
    for _ in range(6):
        loss = F.cross_entropy(m(x), y, reduction="sum")
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


def test_fused_cnn3d_pooled_buffers_are_returned_once_and_a_second_backward_is_refused():
    """the fused 3-D path keeps two pooled (~GB-scale at config 2) padded activations as saved tensors.  They return to
    the pool exactly once — after the backward, or when a grad-enabled forward's graph is dropped without a backward — and a
    second backward over a retained graph raises instead of reading recycled buffers."""
    import gc
    from videonavqa_amd import ops
    from videonavqa_amd.models import VideoOnlyCNN3D
    torch.manual_seed(0)
    m = VideoOnlyCNN3D(70, fc6_in_features=128 * 1 * 1 * 1, precision=LOW).cuda().train()
    x = torch.rand(2, 3, 16, 48, 48, device="cuda")
    if not m._fast_ok(x):
        pytest.skip("geometry not on the fused path")
    y = torch.randint(0, 70, (2,), device="cuda")
    count = lambda: sum(len(v) for v in ops._HALO_POOL.free.values())
    loss = F.cross_entropy(m(x), y, reduction="sum")
    loss.backward(retain_graph=True)
    after_bwd = count()
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()
    assert count() == after_bwd                       # nothing returned twice
    del loss
    out = m(x)                                        # grad-enabled forward, never followed by a backward
    held = count()
    del out
    gc.collect()
    assert count() >= held                            # its lease went back when the graph was freed (pool keeps at most 2 per shape)
