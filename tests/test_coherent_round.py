"""CPU: coherent rounding of frozen 16-bit weights (videonavqa_amd.stem.coherent_round) — pure tensor arithmetic, no GPU:
every weight stays one of the two 16-bit neighbours of its fp32 value, and each output channel's rounding errors cancel against
the mean input activation."""
import pytest
import torch

from videonavqa_amd.stem import coherent_round


def _neighbours(w, dt):
    r = w.to(dt).float()
    step = torch.where(r < w, torch.full_like(w, float("inf")), torch.full_like(w, float("-inf")))
    other = torch.nextafter(r.to(dt), step.to(dt)).float()
    return r, other


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(128, 64, 3, 3), (64, 3, 3, 3), (32, 128, 5, 5)])
def test_coherent_round_properties(dt, shape):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(shape, generator=g) / (shape[1] * shape[2] * shape[3]) ** 0.5
    w[0, :, 0, 0] = 0.0                                     # exact zeros (channel padding) ...
    w[1] = w[1].to(dt).float()                              # ... and exactly representable rows stay as they are
    m = torch.rand(shape[1], generator=g) + 0.1
    q = coherent_round(w, m, dt)
    assert q.shape == w.shape and q.dtype == torch.float32
    assert (q.to(dt).float() == q).all()                    # representable
    near, other = _neighbours(w, dt)
    assert ((q == near) | (q == other)).all()               # one of the two neighbours: element-wise error below one ulp
    assert (q[0, :, 0, 0] == 0).all() and (q[1] == w[1]).all()
    mm = m.view(1, -1, 1, 1)
    r_rtn = ((near - w) * mm).sum((1, 2, 3))
    r_coh = ((q - w) * mm).sum((1, 2, 3))
    small = shape[1] * shape[2] * shape[3] < 100          # conv1_1's 27 weights per channel: few candidates, coarser cancellation
    assert float(r_coh.pow(2).mean().sqrt()) < (0.3 if small else 0.15) * float(r_rtn.pow(2).mean().sqrt())
    assert (r_coh.abs() <= r_rtn.abs() + 1e-12).all()       # never worse than round-to-nearest on any channel
    # the price: the few flipped weights (near a rounding midpoint) add < 2 % to the element-wise rms error
    assert float((q - w).pow(2).mean().sqrt()) < (1.25 if small else 1.02) * float((near - w).pow(2).mean().sqrt())
    assert float((q != near).float().mean()) < (0.2 if small else 0.08)


def test_coherent_round_output_error_on_positive_inputs():
    """What it buys: on inputs with a positive mean (post-ReLU activations) the MEAN over pixels of a conv's output error — the part
    later pooling cannot average away — drops by > 5x on fresh data; the total rms error does not grow."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(2)
    C = 32
    w = torch.randn(64, C, 3, 3, generator=g) / (C * 9) ** 0.5
    calib = torch.relu(torch.randn(4, C, 24, 24, generator=g) + 0.5)
    fresh = torch.relu(torch.randn(8, C, 24, 24, generator=g) + 0.5)
    q = coherent_round(w, calib.mean((0, 2, 3)), torch.float16)
    ref = F.conv2d(fresh.double(), w.double())
    e_rtn = F.conv2d(fresh.double(), w.half().double()) - ref
    e_coh = F.conv2d(fresh.double(), q.double()) - ref
    coh = lambda e: float(e.mean((0, 2, 3)).pow(2).mean().sqrt())
    assert coh(e_coh) < 0.2 * coh(e_rtn)
    assert float(e_coh.pow(2).mean().sqrt()) < float(e_rtn.pow(2).mean().sqrt())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_second_order_round_minimises_the_output_error_on_correlated_inputs(dt):
    """stem.second_order_round (GPTQ-style sequential rounding against the patch second moment H): values on the 16-bit grid, within
    a few grid steps of the exact weights, and dw^T H dw — the mean squared output error on inputs with that second moment — far below
    round-to-nearest's and below coherent_round's on positive, spatially correlated inputs (what a post-ReLU layer sees)."""
    import torch.nn.functional as F
    from videonavqa_amd.stem import coherent_round, patch_second_moment, second_order_round
    g = torch.Generator().manual_seed(3)
    base = torch.rand(12, 4, 9, 9, generator=g)
    mix = torch.rand(24, 4, generator=g)
    x = F.relu(F.interpolate(torch.einsum("oc,nchw->nohw", mix, base), scale_factor=2, mode="bilinear") - 0.4) + 0.05      # [12, 24, 18, 18]
    w = torch.randn(16, 24, 3, 3, generator=g) / 15
    H = patch_second_moment(x, 3)
    assert H.shape == (216, 216) and H.dtype == torch.float64 and torch.allclose(H, H.t())
    q = second_order_round(w, H, dt)
    assert q.shape == w.shape and torch.equal(q.to(dt).float(), q)
    ulp = (w.abs().max() * (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7))
    assert float((q - w).abs().max()) < 8 * float(ulp)
    err = lambda d: float(((d.reshape(16, -1).double() @ H) * d.reshape(16, -1).double()).sum())
    rtn, coh, so = err(w.to(dt).float() - w), err(coherent_round(w, x.mean((0, 2, 3)), dt) - w), err(q - w)
    assert so < 0.25 * rtn and so < 0.7 * coh, (so, coh, rtn)
    y = F.conv2d(x, w, padding=1)
    mse = lambda ww: float((F.conv2d(x, ww, padding=1) - y).pow(2).mean())
    assert mse(q) < 0.3 * mse(w.to(dt).float())


def test_default_calibration_frames_are_a_seeded_mixture():
    """stem.default_calibration_frames: deterministic for a seed, values in [0, 1], the first half i.i.d. noise (neighbouring pixels
    uncorrelated), the second half smooth (neighbouring pixels nearly equal) — the two kinds whose mixture the second-order rounding of
    the frozen stem's weights is calibrated on."""
    from videonavqa_amd.stem import CALIBRATION_FRAMES, default_calibration_frames
    a = default_calibration_frames(CALIBRATION_FRAMES, 64, 96)
    b = default_calibration_frames(CALIBRATION_FRAMES, 64, 96)
    assert a.shape == (CALIBRATION_FRAMES, 3, 64, 96) and torch.equal(a, b) and float(a.min()) >= 0 and float(a.max()) <= 1
    assert not torch.equal(a, default_calibration_frames(CALIBRATION_FRAMES, 64, 96, seed=1))
    h = CALIBRATION_FRAMES - CALIBRATION_FRAMES // 2
    rough = lambda t: float((t[..., 1:] - t[..., :-1]).abs().mean())
    assert rough(a[:h]) > 0.25 and rough(a[h:]) < 0.05
    assert default_calibration_frames(1, 32, 32).shape == (1, 3, 32, 32)
