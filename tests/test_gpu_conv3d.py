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
@pytest.mark.parametrize("cfg", [(2, 4, 10, 12, 64, 64), (1, 6, 8, 8, 128, 192), (3, 3, 14, 6, 64, 128)])
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
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, :, -1].abs().max()) == 0     # zero halo kept
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
