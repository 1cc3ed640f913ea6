"""CPU: the algebra of MEAN-SHIFTED storage (stem.FrozenStem._setup_mean_shift, ops.FilmTrunkHeadFn) on plain torch
tensors — what makes the scheme exact rather than approximate:

  * a tensor stored as x' = x - mu_c with -mu_c in its halo IS the zero-padded x everywhere, so conv(W, x' incl. halo) + sum_taps(W) mu equals
    the zero-padded conv of x at EVERY output pixel, the image border included;
  * relu(a) - mu = max(a - mu, -mu), and the 2x2 max-pool commutes with the shift (the composed conv's per-channel ReLU floor);
  * the weight gradient of a conv over a mean-shifted input is the gradient over x' (halo included) plus mu (x) db for every tap;
  * ops.unshift_features / ops.shift_bias_correction, the two host helpers, do exactly that arithmetic.
The GPU tests (tests/test_gpu_fp16h.py) check the kernels; this file pins the identities they rely on."""
import torch
import torch.nn.functional as F


def _shifted_padded(x, mu, halo=1):
    """x [N, C, H, W] -> x' padded by `halo` with -mu_c in the halo (NCHW)."""
    xp = F.pad(x, (halo,) * 4)                       # zero padding of the TRUE tensor ...
    return xp - mu.view(1, -1, 1, 1)                 # ... minus mu everywhere: interior x - mu, halo -mu


def test_conv_over_a_mean_shifted_tensor_with_bias_correction_is_the_zero_padded_conv():
    from videonavqa_amd import ops
    g = torch.Generator().manual_seed(0)
    for k in (3, 5):
        x = torch.rand(2, 6, 9, 7, generator=g, dtype=torch.float64) * 3
        w = torch.randn(5, 6, k, k, generator=g, dtype=torch.float64)
        b = torch.randn(5, generator=g, dtype=torch.float64)
        mu = x.mean((0, 2, 3))
        ref = F.conv2d(x, w, b, padding=k // 2)
        got = F.conv2d(_shifted_padded(x, mu, k // 2), w, b)       # the kernel's view: no padding logic, the halo is data
        corr = ops.shift_bias_correction(w.float(), mu.float(), 8)[:5].double()
        assert float((got + corr.view(1, -1, 1, 1) - ref).abs().max()) < 1e-5 * float(ref.abs().max())
        # exact in float64 with the float64 correction, border pixels included
        corr64 = w.sum((2, 3)) @ mu
        assert float((got + corr64.view(1, -1, 1, 1) - ref).abs().max()) < 1e-12 * float(ref.abs().max())


def test_relu_floor_and_pool_commute_with_the_shift():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 4, 8, 8, generator=g, dtype=torch.float64)
    mu = torch.rand(4, generator=g, dtype=torch.float64).view(1, -1, 1, 1)
    want = F.max_pool2d(F.relu(a), 2) - mu
    got = F.max_pool2d(torch.maximum(a - mu, -mu), 2)               # VNQA_CONV_RELU_FLOOR: -mu in the bias and as the ReLU's floor
    assert torch.equal(want, got) or float((want - got).abs().max()) < 1e-15


def test_weight_gradient_over_a_mean_shifted_input_gets_mu_outer_db():
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 5, 6, 6, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(4, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(2, 4, 6, 6, generator=g, dtype=torch.float64)
    mu = x.mean((0, 2, 3))
    F.conv2d(x, w, b, padding=1).backward(dy)
    dw_true, db = w.grad.clone(), b.grad.clone()
    w2 = w.detach().clone().requires_grad_(True)
    F.conv2d(_shifted_padded(x, mu), w2, None).backward(dy)       # what the wgrad kernel contracts: dY with x' (halo -mu included)
    dw = w2.grad + db.view(-1, 1, 1, 1) * mu.view(1, -1, 1, 1)      # ... + mu (x) db for every tap (ops.FilmTrunkHeadFn._backward)
    assert float((dw - dw_true).abs().max()) < 1e-12 * float(dw_true.abs().max())


def test_unshift_features_restores_the_plain_tensor_with_a_zero_halo():
    from videonavqa_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 4, 5, 8, generator=g)                          # NHWC interior
    mu = torch.rand(8, generator=g)
    stored = torch.empty(2, 6, 7, 8)
    stored[:] = -mu
    stored[:, 1:-1, 1:-1] = x - mu
    plain = ops.unshift_features(stored, mu)
    assert float(plain[:, 0].abs().max()) == 0 and float(plain[:, :, -1].abs().max()) == 0
    assert float((plain[:, 1:-1, 1:-1] - x).abs().max()) < 1e-6
