"""GPU: precision='fp16h' — SPLIT activations [hi | lo (| hi)] (the patch-stationary conv's dual epilogue, VNQA_CONV_DUAL_OUT /
_HI2), the three-product conv on a split tensor (a plain conv over 3 C channels against [w_hi | w_hi | w_lo], plain and with the
BNSTATS epilogue, on the patch-stationary and the implicit-GEMM tiles), the weight gradient from a split tensor's first segment
(VNQA_WGRAD_X_PAIR / _X_TRIPLE), and the precision itself against the exact-f32 path, the reference goldens and north
star's 1e-3 at BASELINE.json's size on 4 weight seeds x 12 minibatches and on data that does not look like the calibration frames.

Runs in the fp16 build of the library (one 16-bit storage format per process): collected only when the process's format is f16
(VNQA_TEST_LOW_PRECISION=fp16: tests/test_gpu_fp16.py starts that pytest run)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import LOW, QV_CASES, build_product_model, rel_err

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(LOW != "fp16", reason="pair tensors are fp16 halves: run with VNQA_TEST_LOW_PRECISION=fp16 "
                                                       "(tests/test_gpu_fp16.py does)")]
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture
def no_mean_shift():
    """The stem WITHOUT mean-shifted storage (round 6's default for calibrated 16-bit stems): the round-5 forms — split tensors between
    conv22 / conv31 / conv32, [hi | lo | hi] features — stay supported (calibrations without channel means, VNQA_MEAN_SHIFT=0) and tested."""
    from videonavqa_amd import stem as S
    old = S.MEAN_SHIFT
    S.MEAN_SHIFT = 0
    yield
    S.MEAN_SHIFT = old


def _padded(n, h, w, c, seed, scale=1.0, positive=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(n, h + 2, w + 2, c)
    v = torch.randn(n, h, w, c, generator=g) * scale
    x[:, 1:1 + h, 1:1 + w] = v.abs() if positive else v
    return x.cuda()


def _torch_conv(x_pad, w_oihw, bias, relu, pool, post=None):
    x = x_pad[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double()
    y = F.conv2d(x, w_oihw.double(), bias.double(), padding=1)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2)
    if post is not None:
        y = y * post[0].double().view(1, -1, 1, 1) + post[1].double().view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("cfg", [dict(h=28, w=28, relu=True, pool=True), dict(h=14, w=14, relu=False, pool=False),
                                 dict(h=14, w=14, relu=True, pool=False, post=True), dict(h=20, w=26, relu=True, pool=True),
                                 dict(h=28, w=28, relu=True, pool=False, n=3, cout=256)])
def test_dual_output_pair_holds_the_fp32_result_in_two_halves(cfg):
    """VNQA_CONV_DUAL_OUT on the patch-stationary tiles (both tile widths, tiles straddling images, pooling, affine, an overlapping
    last column block): hi is bit for bit the plain launch's 16-bit output; hi + lo reproduces the fp32 result (bias, ReLU, pool,
    affine in fp32) to 2^-21 — against a float64 torch conv of the same fp16 operands: the accumulation error of K = 4608 fp32 terms."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout = cfg.get("n", 5), cfg["h"], cfg["w"], 512, cfg.get("cout", 512)
    x = _padded(n, h, w, cin, 1).half()
    g = torch.Generator().manual_seed(2)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda().half().float()
    wt = K.pack_conv_weight(w4, torch.float16)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    post = ((torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()) if cfg.get("post") else None
    kw = dict(bias=bias, relu=cfg["relu"], pool2=cfg["pool"], post_scale=post[0] if post else None, post_shift=post[1] if post else None,
              tile=L.TILE_STEM_PS_224x256)
    plain = K.conv2d_igemm(x, wt, **kw)
    pair = K.conv2d_igemm(x, wt, dual_out=True, **kw)
    assert pair.shape == plain.shape[:3] + (2 * cout,) and pair.dtype == torch.float16
    hi, lo = pair[..., :cout], pair[..., cout:]
    if post is None:
        assert torch.equal(hi, plain)
    else:       # (the plain epilogue applies the affine to the storage-rounded value and rounds again; the dual one works in fp32)
        assert float((hi.float() - plain.float()).abs().max()) <= 2.0 ** -9 * float(plain.float().abs().max())
    assert float(pair[:, 0].abs().max()) == 0 and float(pair[:, :, -1].abs().max()) == 0          # zero halo (fresh output)
    ref = _torch_conv(x.float(), w4, bias, cfg["relu"], cfg["pool"], post)
    got = (hi.double() + lo.double())[:, 1:-1, 1:-1]
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-6 * scale, float((got - ref).abs().max()) / scale
    assert float((hi.double()[:, 1:-1, 1:-1] - ref).abs().max()) > 1e-4 * scale            # (the hi half alone is an fp16 rounding away)
    assert float(lo.abs().max()) <= 2.0 ** -10 * float(hi.abs().max())


@pytest.mark.parametrize("cfg", [dict(h=28, w=28, relu=True, pool=True, segs=2), dict(h=14, w=14, relu=True, pool=False, post=True, segs=3),
                                 dict(h=20, w=26, relu=True, pool=True, segs=2), dict(h=14, w=14, relu=False, pool=False, segs=2, cin=1024)])
def test_dual_output_of_the_implicit_gemm_tile_equals_the_patch_stationary_one(cfg):
    """Round 6: VNQA_CONV_DUAL_OUT [| _HI2] on the 256x256 implicit-GEMM tile (TAG 5: fp32 epilogue in two cout passes) writes the
    SAME [hi | lo (| hi)] tensor as the patch-stationary kernel wherever both serve the geometry (hi bit for bit; lo is the rounding
    of an fp32 sum formed in another order, so hi + lo agree to the fp32 accumulation's 2^-21) — incl. the 2 C-channel split INPUT."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout, segs = 5, cfg["h"], cfg["w"], cfg.get("cin", 512), 512, cfg["segs"]
    x = _padded(n, h, w, cin, 21).half()
    g = torch.Generator().manual_seed(22)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda().half().float()
    wt = K.pack_conv_weight(w4, torch.float16)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    post = ((torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()) if cfg.get("post") else None
    kw = dict(bias=bias, relu=cfg["relu"], pool2=cfg["pool"], post_scale=post[0] if post else None, post_shift=post[1] if post else None,
              dual_out=segs)
    a = K.conv2d_igemm(x, wt, tile=L.TILE_STEM_PS_224x256, **kw)
    b = K.conv2d_igemm(x, wt, tile=L.TILE_STEM_256x256, **kw)
    assert a.shape == b.shape and b.shape[-1] == segs * cout
    va, vb = a[..., :cout].double() + a[..., cout:2 * cout].double(), b[..., :cout].double() + b[..., cout:2 * cout].double()
    scale = float(va.abs().max())
    assert float((va - vb).abs().max()) < 2e-6 * scale
    flips = float((a[..., :cout] != b[..., :cout]).float().mean())          # (an fp32 sum within 2^-21 of a rounding midpoint may fall either way)
    assert flips < 2e-3, flips
    if segs == 3:
        assert torch.equal(b[..., :cout], b[..., 2 * cout:])
    assert float(b[:, 0].abs().max()) == 0 and float(b[:, :, -1].abs().max()) == 0 and float(b[:, :, 0].abs().max()) == 0


@pytest.mark.parametrize("cfg", [dict(h=10, w=13, relu=True, pool=False, segs=3), dict(h=10, w=13, relu=False, pool=False, segs=2, post=True),
                                 dict(h=20, w=26, relu=True, pool=True, segs=2, n=7)])
def test_dual_output_of_the_implicit_gemm_tile_on_the_reference_geometry(cfg):
    """... and on the maps of the reference's own 160 x 208 frames (eval/utils.py:24-25: 20 x 26 -> 10 x 13), which the patch-stationary
    tiles do not serve: hi + lo is the fp32 result against a float64 torch conv of the same fp16 operands; hi is the plain launch."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout, segs = cfg.get("n", 9), cfg["h"], cfg["w"], 512, 512, cfg["segs"]
    x = _padded(n, h, w, cin, 31).half()
    g = torch.Generator().manual_seed(32)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda().half().float()
    wt = K.pack_conv_weight(w4, torch.float16)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    post = ((torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()) if cfg.get("post") else None
    kw = dict(bias=bias, relu=cfg["relu"], pool2=cfg["pool"], post_scale=post[0] if post else None, post_shift=post[1] if post else None,
              tile=L.TILE_STEM_256x256)
    plain = K.conv2d_igemm(x, wt, **kw)
    pair = K.conv2d_igemm(x, wt, dual_out=segs, **kw)
    hi, lo = pair[..., :cout], pair[..., cout:2 * cout]
    if post is None:
        assert torch.equal(hi, plain)
    ref = _torch_conv(x.float(), w4, bias, cfg["relu"], cfg["pool"], post)
    got = (hi.double() + lo.double())[:, 1:-1, 1:-1]
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-6 * scale, float((got - ref).abs().max()) / scale
    assert float(lo.abs().max()) <= 2.0 ** -10 * float(hi.abs().max())
    assert float(pair[:, 0].abs().max()) == 0 and float(pair[:, :, -1].abs().max()) == 0


def test_split_out_on_the_implicit_gemm_tile_equals_the_patch_stationary_one():
    """VNQA_EPI_SPLIT_OUT (conv_init of precision 'fp16h') on tile 256x256: the two tensors (hi, lo) of the 10 x 13 trunk."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, cin, cout = 6, 512, 128
    g = torch.Generator().manual_seed(42)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(w4, torch.float32)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    for h, w in ((14, 14), (10, 13)):
        tri = _split3(_padded(n, h, w, cin, 41, positive=True))
        hb, lb = K.conv2d_igemm_split_out(tri, wt32, bias, True, split_in=True, tile=L.TILE_256x256)
        if (h, w) == (14, 14):
            ha, la = K.conv2d_igemm_split_out(tri, wt32, bias, True, split_in=True)
            sa, sb = ha.double() + la.double(), hb.double() + lb.double()
            assert float((sa - sb).abs().max()) < 2e-6 * float(sa.abs().max())
        plain = K.conv2d_igemm(tri, wt32, bias=bias, relu=True, split_in=True)
        assert float((hb != plain).float().mean()) < 2e-3
        assert float(lb.abs().max()) <= 2.0 ** -10 * float(hb.abs().max())
        assert float(hb[:, 0].abs().max()) == 0 and float(lb[:, :, -1].abs().max()) == 0


def test_conv_over_a_pair_tensor_with_doubled_weights_contracts_the_unrounded_activation():
    """The stem's conv31 / conv32 in precision 'fp16h': a plain conv over the pair tensor's 2 C channels against [w | w] equals the
    conv of the fp32 activation (hi + lo) with w — the consumer no longer sees the producer's storage rounding."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, c = 4, 14, 14, 512
    v = _padded(n, h, w, c, 3, positive=True)                       # the fp32 activation
    hi = v.half()
    lo = (v - hi.float()).half()
    pair = torch.cat([hi, lo], dim=-1).contiguous()
    g = torch.Generator().manual_seed(4)
    w4 = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).cuda().half().float()
    wt = K.pack_conv_weight(w4, torch.float16)
    wt2 = torch.cat([wt, wt], dim=2).contiguous()
    bias = torch.zeros(c, device="cuda")
    got = K.conv2d_igemm(pair, wt2, bias=bias, tile=L.TILE_STEM_PS_224x256, dual_out=True)
    got = (got[..., :c].double() + got[..., c:].double())[:, 1:-1, 1:-1]
    ref = _torch_conv(v, w4, bias, False, False)
    one = _torch_conv(hi.float(), w4, bias, False, False)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-6 * scale
    assert float((one - ref).abs().max()) > 3e-5 * scale            # what the plain fp16 input would have cost


def test_triple_output_feeds_a_plain_conv_with_split_weights_three_products():
    """VNQA_CONV_DUAL_HI2: the producer lays out [hi | lo | hi]; its consumer — a PLAIN patch-stationary conv over 3 C channels against
    [w_hi | w_hi | w_lo] — contracts the unrounded activation with unrounded weights (the stem's conv31 / conv32 in precision
    'fp16h'): against float64 on the fp32 operands, <= 3e-6 (x_lo . w_lo dropped, fp32 accumulation of K = 13 824 terms)."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, c = 4, 14, 14, 512
    x0 = _padded(n, h, w, c, 11).half()
    g = torch.Generator().manual_seed(12)
    wa = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).cuda().half().float()
    wb = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).cuda()                       # fp32 weights, NOT fp16-representable
    bias = torch.zeros(c, device="cuda")
    tri = K.conv2d_igemm(x0, K.pack_conv_weight(wa, torch.float16), bias=bias, relu=True, tile=L.TILE_STEM_PS_224x256, dual_out=3)
    assert tri.shape[-1] == 3 * c and torch.equal(tri[..., :c], tri[..., 2 * c:])
    v = tri[..., :c].float() + tri[..., c:2 * c].float()                                      # the producer's fp32 result
    wt3 = K.split_weight3(K.pack_conv_weight(wb, torch.float32))
    got = K.conv2d_igemm(tri, wt3, bias=bias, tile=L.TILE_STEM_PS_224x256, dual_out=True)
    got = (got[..., :c].double() + got[..., c:].double())[:, 1:-1, 1:-1]
    ref = _torch_conv(v, wb, bias, False, False)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-6 * scale, float((got - ref).abs().max()) / scale
    rounded = _torch_conv(v.half().float(), wb.half().float(), bias, False, False)            # what one fp16 product would see
    assert float((rounded - ref).abs().max()) > 1e-4 * scale


def _split3(v):
    hi = v.half()
    return torch.cat([hi, (v - hi.float()).half(), hi], dim=-1).contiguous()


@pytest.mark.parametrize("taps,tile", [(9, "ps"), (9, "igemm"), (1, "igemm")])      # (the patch-stationary kernel is a 3x3 / 5x5 kernel)
def test_three_product_conv_on_a_split_tensor_matches_exact_f32(taps, tile):
    """split_in: [x_hi | x_lo | x_hi] against [w_hi | w_hi | w_lo] — a plain conv over 3 C channels — against the exact-f32 MFMA
    conv of the same library on the fp32 activation and fp32 weights: the fp16 OUTPUT rounding is all that is left."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout = 6, 14, 14, 512, 512
    v = _padded(n, h, w, cin, 5, positive=True)
    tri = _split3(v)
    g = torch.Generator().manual_seed(6)
    k = 3 if taps == 9 else 1
    w4 = (torch.randn(cout, cin, k, k, generator=g) / (cin * taps) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(w4, torch.float32)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    ref = K.conv2d_igemm(v, wt32, bias=bias, relu=True)                                      # exact-f32 MFMA path
    got = K.conv2d_igemm(tri, wt32, bias=bias, relu=True, split_in=True, tile=L.TILE_PS_224x256 if tile == "ps" else L.TILE_AUTO)
    assert got.dtype == torch.float16 and got.shape == ref.shape
    assert float((got.float() - ref).abs().max()) < 6e-4 * float(ref.abs().max())
    plain = K.conv2d_igemm(v.half(), K.pack_conv_weight(w4, torch.float16), bias=bias, relu=True)
    e3 = float((got.float() - ref).pow(2).mean().sqrt())
    e1 = float((plain.float() - ref).pow(2).mean().sqrt())
    r0 = float((ref.half().float() - ref).pow(2).mean().sqrt())                             # the output rounding alone
    assert e3 < 1.05 * r0 and e1 > 1.3 * r0, (e3, e1, r0)


def test_three_product_conv_with_bnstats_epilogue_matches_the_two_pass_statistics():
    from videonavqa_amd import kernels as K
    from videonavqa_amd.models.common import FrameLayout
    lay = FrameLayout([4, 4, 3, 2], 4, "cuda")
    n, h, w, cin, cout = lay.n_img, 14, 14, 512, 512
    v = _padded(n, h, w, cin, 7, positive=True)
    pair = _split3(v)
    g = torch.Generator().manual_seed(8)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(w4, torch.float32)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    fused = K.conv2d_igemm_bnstats(pair, wt32, bias, True, lay.frame_of_i32, lay.frame_off_i32, lay.n_frames, min(lay.cts), split_in=True)
    assert fused is not None
    y, mean, var = fused
    y2 = K.conv2d_igemm(pair, wt32, bias=bias, relu=True, split_in=True)
    assert torch.equal(y, y2)
    m2, v2 = K.frame_bn_stats(y2, lay.frame_off_i32, lay.n_frames)
    assert float((mean - m2).abs().max()) < 1e-5 * float(m2.abs().max())
    assert float((var - v2).abs().max()) < 1e-4 * float(v2.abs().max())


def test_split_out_conv_and_the_batchnorm_of_its_two_tensors_match_exact_f32():
    """VNQA_EPI_SPLIT_OUT + vnqa_frame_bn_stats_split / _apply_split (conv_init of precision 'fp16h'): hi is the plain launch's
    tensor bit for bit, hi + lo the fp32 result (to the three products' own 1e-5), and the BatchNorm of the pair is the exact-f32 path's up to its ONE
    output rounding — the plain path's result carries the conv output's rounding as well."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    from videonavqa_amd.models.common import FrameLayout
    lay = FrameLayout([4, 4, 3, 2], 4, "cuda")
    n, h, w, cin, cout = lay.n_img, 14, 14, 512, 128
    v = _padded(n, h, w, cin, 17, positive=True)
    tri = _split3(v)
    g = torch.Generator().manual_seed(18)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(w4, torch.float32)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    gamma = (1 + 0.1 * torch.randn(cout, generator=g)).cuda()
    beta = (0.1 * torch.randn(cout, generator=g)).cuda()
    ref = K.conv2d_igemm(v, wt32, bias=bias, relu=True)                                      # exact-f32 MFMA path
    hi, lo = K.conv2d_igemm_split_out(tri, wt32, bias, True, split_in=True)
    plain = K.conv2d_igemm(tri, wt32, bias=bias, relu=True, split_in=True, tile=L.TILE_PS_224x256)
    assert torch.equal(hi, plain)
    # (the three products leave x_lo . w_lo and the operands' own lo roundings: a few 2^-22 of the sums' magnitude)
    assert float((hi.float() + lo.float() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    assert float(hi[:, 0].abs().max()) == 0 and float(lo[:, 0].abs().max()) == 0 and float(lo[:, :, -1].abs().max()) == 0
    m_ref, v_ref = K.frame_bn_stats(ref, lay.frame_off_i32, lay.n_frames)
    mean, var = K.frame_bn_stats_split(hi, lo, lay.frame_off_i32, lay.n_frames)
    assert float((mean - m_ref).abs().max()) < 2e-6 * float(m_ref.abs().max())
    assert float((var - v_ref).abs().max()) < 1e-5 * float(v_ref.abs().max())
    rstd = torch.rsqrt(v_ref + 1e-5)
    y_ref = K.frame_bn_apply(ref, lay.frame_of_i32, m_ref, rstd, gamma, beta)
    y = K.frame_bn_apply_split(hi, lo, lay.frame_of_i32, m_ref, rstd, gamma, beta)
    y1 = K.frame_bn_apply(hi, lay.frame_of_i32, m_ref, rstd, gamma, beta)
    assert y.dtype == torch.float16 and float(y[:, 0].abs().max()) == 0
    r0 = float((y_ref.half().float() - y_ref).pow(2).mean().sqrt())                         # the output rounding alone
    e2 = float((y.float() - y_ref).pow(2).mean().sqrt())
    e1 = float((y1.float() - y_ref).pow(2).mean().sqrt())
    assert e2 < 1.02 * r0 and e1 > 1.2 * r0, (e2, e1, r0)


@pytest.mark.parametrize("segs", [2, 3])
def test_wgrad_from_a_split_tensor_contracts_its_first_segment(segs):
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout = 5, 14, 14, 512, 512
    v = _padded(n, h, w, cin, 9, positive=True)
    hi = v.half().contiguous()
    parts = [hi, (v - hi.float()).half()] + ([hi] if segs == 3 else [])
    x = torch.cat(parts, dim=-1).contiguous()
    dy = _padded(n, h, w, cout, 10, scale=0.01).half()
    a, da = K.conv2d_wgrad(hi, dy, 9)
    b, db = K.conv2d_wgrad(x, dy, 9, x_segs=segs)
    assert a.shape == b.shape == (cout, 9, cin)
    assert torch.equal(a, b) and torch.equal(da, db)


def _padded_halo(n, h, w, c, seed, halo=1, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(n, h + 2 * halo, w + 2 * halo, c)
    x[:, halo:halo + h, halo:halo + w] = torch.randn(n, h, w, c, generator=g) * scale
    return x.cuda()


@pytest.mark.parametrize("cfg", [dict(cin=64, cout=128, taps=9, relu=True, pool=False, tile=15),
                                 dict(cin=128, cout=128, taps=9, relu=True, pool=True, tile=15, post=True, y_halo=2),
                                 dict(cin=512, cout=512, taps=9, relu=False, pool=False, tile=0),
                                 dict(cin=512, cout=512, taps=1, relu=True, pool=False, tile=0),
                                 dict(cin=128, cout=512, taps=25, relu=True, pool=True, tile=1, border=True)])
def test_split_weights_two_product_conv_removes_the_weight_rounding(cfg):
    """split_weights: the conv with a plain fp16 activation read twice along K against [w_hi | w_lo] (igemm TAG 4,
    VNQA_CONV_X_WRAP2).  Against the exact fp32 conv of the SAME fp16-valued activation with the fp32 weights: only the output's own
    fp16 rounding is left (<= 2^-11 relative to the value), whereas the plain fp16 conv also carries the weight rounding."""
    from videonavqa_amd import kernels as K
    cin, cout, taps = cfg["cin"], cfg["cout"], cfg["taps"]
    halo = 2 if taps == 25 else 1
    n, h, w = 3, 12, 16
    x16 = _padded_halo(n, h, w, cin, 21, halo).half()
    g = torch.Generator().manual_seed(22)
    k = {9: 3, 1: 1, 25: 5}[taps]
    wf = (torch.randn(cout, cin, k, k, generator=g) / (cin * taps) ** 0.5).cuda()
    wt32 = K.pack_conv_weight(wf, torch.float32)
    bias = torch.randn(cout, generator=g).cuda() * 0.1
    post = (torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda() * 0.1) if cfg.get("post") else (None, None)
    ring32 = torch.randn(n, 2 * w + 2 * (h - 2), cout, generator=g).cuda() * 0.05 if cfg.get("border") else None
    yh = cfg.get("y_halo", 1)
    kw = dict(bias=bias, relu=cfg["relu"], pool2=cfg["pool"], post_scale=post[0], post_shift=post[1], x_halo=halo, y_halo=yh)
    ref = K.conv2d_igemm(x16.float(), wt32, border_sub=ring32, **kw)                         # exact fp32 on the same values
    ring16 = None if ring32 is None else ring32.half()
    if ring16 is not None:
        ref = K.conv2d_igemm(x16.float(), wt32, border_sub=ring16.float(), **kw)
    plain = K.conv2d_igemm(x16, K.pack_conv_weight(wf, torch.float16), border_sub=ring16, **kw)
    got = K.conv2d_igemm(x16, wt32, border_sub=ring16, tile=cfg["tile"], split_weights=True, **kw)
    assert got.dtype == torch.float16 and got.shape == plain.shape
    scale = float(ref.abs().max())
    e_w2, e_plain = float((got.float() - ref).abs().max()) / scale, float((plain.float() - ref).abs().max()) / scale
    assert e_w2 < 6e-4, (e_w2, e_plain)                                    # the output's own rounding (2^-11) is what is left
    # ... and in the L2 norm the two-product result is never farther from the fp32 conv than the plain fp16 conv (whose error also
    # holds the weight rounding; for a single layer the output rounding dominates both, the gain shows over a stack of layers)
    assert float((got.float() - ref).norm()) <= 1.0 * float((plain.float() - ref).norm()), (e_w2, e_plain)
    assert float(got[:, 0].float().abs().max()) == 0                       # zero halo kept


def test_split_weights_gemm_nt_matches_exact_f32_up_to_output_rounding():
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(23)
    a = torch.randn(280, 4096, generator=g).cuda().half()
    b = (torch.randn(128, 4096, generator=g) / 64.0).cuda()
    bias = torch.randn(128, generator=g).cuda()
    ref = a.float() @ b.t() + bias
    got = K.gemm_nt(a, b, bias=bias, split_weights=True)
    plain = K.gemm_nt(a, b.half(), bias=bias)
    assert got.dtype == torch.float16
    assert float((got.float() - ref).abs().max()) < 6e-4 * float(ref.abs().max())
    assert float((got.float() - ref).norm()) <= float((plain.float() - ref).norm())


def _random_stem(prec):
    import torch.nn as nn
    from videonavqa_amd.models import ObjDetectCNN
    from videonavqa_amd.stem import VGGFront
    torch.manual_seed(0)
    vgg, od = VGGFront(prec), ObjDetectCNN(5, 512, 8, 0, True, True, precision=prec)
    with torch.no_grad():
        for conv in vgg.features.values():
            nn.init.kaiming_uniform_(conv.weight, a=1.0)
        for m in od.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1.0)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.8, 1.2)
    return vgg.cuda().eval(), od.cuda().eval()


def test_fp16h_stem_split_features_are_closer_to_exact_than_the_fp16_stem(no_mean_shift):
    """The frozen stem at 224 x 224 (the geometry whose 28 x 28 / 14 x 14 maps the pair path serves): pair features (hi + lo) against
    the exact-f32 stem are closer than the fp16 stem's, the hi half is a valid fp16 feature tensor, and with split_features=False
    (consumers that read no pairs) the output is a plain tensor of the usual shape.  At 160 x 208 (10 x 13 maps: not served by the
    patch-stationary kernel) the split tensors come from the implicit-GEMM tile's dual epilogue."""
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem
    torch.set_grad_enabled(False)
    try:
        clip = torch.rand(2, 3, 224, 224, 2, generator=torch.Generator().manual_seed(5)).cuda()
        lay = FrameLayout([2, 1], 2, "cuda")
        out = {}
        for name, prec, kw in (("ref", "fp32", {}), ("h", "fp16h", {}), ("p", "fp16", {}), ("hp", "fp16h", dict(split_features=False))):
            vgg, od = _random_stem(prec)
            stem = FrozenStem(vgg, od, prec, **kw)
            out[name] = stem.forward_clip(clip, lay.img_of, lay.n_img).clone()
        assert out["h"].shape[-1] == 1536 and out["p"].shape[-1] == 512 and out["hp"].shape == out["p"].shape
        assert torch.equal(out["h"][..., :512], out["h"][..., 1024:])
        ref = out["ref"][..., :512].double()
        pair = out["h"][..., :512].double() + out["h"][..., 512:1024].double()
        e_h = float((pair - ref).pow(2).mean().sqrt())
        e_p = float((out["p"].double() - ref).pow(2).mean().sqrt())
        e_hp = float((out["hp"].double() - ref).pow(2).mean().sqrt())
        assert e_h < 0.8 * e_p and e_hp < 0.95 * e_p, (e_h, e_hp, e_p)
        assert float((out["h"][..., :512].double() - ref).abs().max()) < 4e-3 * float(ref.abs().max())
        # the reference's own geometry (eval/utils.py:24-25: 160 x 208 frames -> 20 x 26 and 10 x 13 maps): conv22 on the patch-stationary
        # tile, conv31 / conv32 on the implicit-GEMM tile's dual epilogue (round 6; rounds 4-5 fell back to plain fp16 here)
        clip = torch.rand(2, 3, 160, 208, 2, generator=torch.Generator().manual_seed(6)).cuda()
        res = {}
        for prec in ("fp32", "fp16h", "fp16"):
            vgg, od = _random_stem(prec)
            res[prec] = FrozenStem(vgg, od, prec).forward_clip(clip, lay.img_of, lay.n_img).clone()
        assert res["fp16h"].shape == (3, 12, 15, 1536) and res["fp16"].shape == (3, 12, 15, 512)
        assert torch.equal(res["fp16h"][..., :512], res["fp16h"][..., 1024:])
        ref = res["fp32"][..., :512].double()
        pair = res["fp16h"][..., :512].double() + res["fp16h"][..., 512:1024].double()
        e_h, e_p = float((pair - ref).pow(2).mean().sqrt()), float((res["fp16"].double() - ref).pow(2).mean().sqrt())
        assert e_h < 0.8 * e_p, (e_h, e_p)
    finally:
        torch.set_grad_enabled(True)


def test_second_order_round_kernel_matches_the_tensor_recursion():
    """vnqa_second_order_round (one workgroup per output channel, the row in LDS as float64) against the same recursion in host tensor
    arithmetic (blocked updates: another fp64 summation order): same grid, the same objective dw^T H dw."""
    from videonavqa_amd.stem import patch_second_moment, second_order_round
    g = torch.Generator().manual_seed(21)
    x = torch.relu(torch.randn(6, 40, 12, 12, generator=g)) + 0.1
    w = torch.randn(24, 40, 3, 3, generator=g) / 19
    H = patch_second_moment(x, 3)
    q_host = second_order_round(w, H, torch.float16)
    q_dev = second_order_round(w.cuda(), H.cuda(), torch.float16).cpu()
    assert torch.equal(q_dev.half().float(), q_dev)
    err = lambda d: float(((d.reshape(24, -1).double() @ H) * d.reshape(24, -1).double()).sum())
    # (the greedy recursion amplifies a one-ulp difference of the fp64 sums into other roundings downstream: the two results are
    # different, equally good members of the same family — same objective to 15 %, both far below round-to-nearest, each within one
    # grid step of the running value it rounded)
    assert abs(err(q_dev - w) - err(q_host - w)) < 0.15 * err(q_host - w), (err(q_dev - w), err(q_host - w))
    assert err(q_dev - w) < 0.7 * err(w.half().float() - w) and err(q_host - w) < 0.7 * err(w.half().float() - w)      # (weakly correlated toy inputs)
    assert float((q_dev - w).abs().max()) < 8 * float(w.abs().max()) * 2.0 ** -10


def test_second_order_rounded_stem_weights_and_their_reproduction_from_a_checkpoint_calibration(no_mean_shift):
    """The default calibration: stem.second_order_round against the patch second moments of CALIBRATION_FRAMES noise frames.
    (a) the features of precision 'fp16' (every error of that stem is an activation or a weight rounding) are closer to the exact-f32
    stem's than with round-to-nearest weights, on frames unlike the calibration frames too; (b) a stem rebuilt from the calibration
    a checkpoint carries (`stem.calib`: the frames' name + the means) has bit-identical 16-bit weights; (c) a means-only calibration
    (written before round 5) gives the first-order rounding and, in precision 'fp16h', the three-product layers; (d) with the
    second-order weights conv31 / conv32 of 'fp16h' read [hi | lo] against doubled weights."""
    import torch.nn.functional as F
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem
    torch.set_grad_enabled(False)
    try:
        lay = FrameLayout([2, 1], 2, "cuda")
        g = torch.Generator().manual_seed(15)
        noise = torch.rand(2, 3, 224, 224, 2, generator=g).cuda()
        low = torch.rand(4, 3, 14, 14, generator=g)
        smooth = F.interpolate(low, size=(224, 224), mode="bilinear", align_corners=False).view(2, 2, 3, 224, 224).permute(0, 2, 3, 4, 1).contiguous().cuda()
        vgg32, od32 = _random_stem("fp32")
        ref = FrozenStem(vgg32, od32, "fp32")
        vgg, od = _random_stem("fp16")
        so, rtn = FrozenStem(vgg, od, "fp16"), FrozenStem(vgg, od, "fp16", calibration=None)
        assert so.second_order and not rtn.second_order and so.calib["frames"] == "noise" and so._H is None
        for clip in (noise, smooth):
            r = ref.forward_clip(clip, lay.img_of, lay.n_img).double()[..., :512]
            e_so = float((so.forward_clip(clip, lay.img_of, lay.n_img).double() - r).pow(2).mean())
            e_rtn = float((rtn.forward_clip(clip, lay.img_of, lay.n_img).double() - r).pow(2).mean())
            assert e_so < 0.9 * e_rtn, (e_so, e_rtn)
        again = FrozenStem(vgg, od, "fp16", calibration=dict(so.calib))
        assert again.second_order
        for a, b in zip(so.layers_od + so.layers_vgg, again.layers_od + again.layers_vgg):
            wa, wb = a["wt"], b["wt"]
            assert torch.equal(wa.data if hasattr(wa, "data") and not torch.is_tensor(wa) else wa, wb.data if hasattr(wb, "data") and not torch.is_tensor(wb) else wb)
        assert torch.equal(so.first[0], again.first[0])
        plan = so.packed_tensors()       # what Trainer.sync_replicas broadcasts from rank 0: every device tensor of the plan, once
        assert len(plan) >= 20 and all(t.is_cuda for t in plan) and len({t.data_ptr() for t in plan}) == len(plan)
        means_only = {k: v for k, v in so.calib.items() if k != "frames"}
        vggh, odh = _random_stem("fp16h")
        h2, h3 = FrozenStem(vggh, odh, "fp16h"), FrozenStem(vggh, odh, "fp16h", calibration=means_only)
        assert h2.second_order and not h3.second_order
        assert h2.layers_od[4]["wt_split"].shape[2] == 2 * 512 and h3.layers_od[4]["wt_split"].shape[2] == 3 * 512
        assert h2.layers_od[3]["split_out"] == 2 and h3.layers_od[3]["split_out"] == 3
        r = ref.forward_clip(noise, lay.img_of, lay.n_img).double()[..., :512]
        for st in (h2, h3):
            f = st.forward_clip(noise, lay.img_of, lay.n_img)
            assert f.shape[-1] == 1536
            pair = f[..., :512].double() + f[..., 512:1024].double()
            assert float((pair - r).abs().max()) < 3e-3 * float(r.abs().max())
    finally:
        torch.set_grad_enabled(True)


def test_calibration_means_match_a_torch_fp32_pass():
    """stem.calibration_means (the library's exact-f32 stem with tapped layer outputs) against the same means from torch fp32
    convolutions of the reference layer sequence (VGG-16 features[0:10]; models/obj_detector.py:69-86 in eval mode)."""
    import torch.nn.functional as F
    from videonavqa_amd.stem import BN_EPS, calibration_means
    torch.set_grad_enabled(False)
    vgg, od = _random_stem("fp32")
    frames = torch.rand(3, 3, 64, 96, generator=torch.Generator().manual_seed(9))
    got = calibration_means(vgg, od, frames)
    f = vgg.features
    conv = lambda t, c: F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
    bn = lambda t, b: F.batch_norm(t, b.running_mean, b.running_var, b.weight, b.bias, False, 0.0, BN_EPS)
    mean = lambda t: t.double().mean((0, 2, 3)).float().cpu()
    x = frames.cuda()
    want = {"first": mean(x)}
    a = F.relu(conv(x, f["0"])); want["vgg0"] = mean(a)
    a = F.max_pool2d(F.relu(conv(a, f["2"])), 2); want["vgg1"] = mean(a)
    a = F.relu(conv(a, f["5"])); want["vgg2"] = mean(a)
    a = bn(F.max_pool2d(F.relu(conv(a, f["7"])), 2), od.bn_input); want["od0"] = mean(a)
    a = conv(a, od.conv11)                                    # (od1 is formed analytically from od0: exact away from the border)
    a = F.max_pool2d(F.relu(bn(conv(a, od.conv12), od.bn1)), 2); want["od2"] = mean(a)
    a = conv(a, od.conv21); want["od3"] = mean(a)
    a = F.max_pool2d(F.relu(bn(conv(a, od.conv22), od.bn2)), 2); want["od4"] = mean(a)
    a = conv(a, od.conv31); want["od5"] = mean(a)
    for k, w in want.items():
        assert got[k].shape == w.shape, k
        assert float((got[k] - w).abs().max()) < 2e-4 * max(float(w.abs().max()), 1e-3), (k, float((got[k] - w).abs().max()))
    assert got["od1"].shape == (od.conv11.out_channels,)
    torch.set_grad_enabled(True)


def test_coherent_rounding_removes_the_per_channel_offset_of_the_fp16_stem(no_mean_shift):
    """The fp16-storage stem with coherently rounded weights (calibration on noise frames) against round-to-nearest, both compared
    with the exact-f32 stem on a DIFFERENT clip: the per-channel mean of the feature error (what pooling cannot average away) drops
    by more than 2x; the weights differ in a few percent of the entries only."""
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem
    clip = torch.rand(2, 3, 64, 96, 3, generator=torch.Generator().manual_seed(5)).cuda()
    lay = FrameLayout([3, 2], 3, "cuda")
    feats = {}
    for name, prec, cal in (("ref", "fp32", None), ("rtn", "fp16", None), ("coh", "fp16", "noise")):
        vgg, od = _random_stem(prec)
        stem = FrozenStem(vgg, od, prec, calibration=cal)
        assert (stem.calib is not None) == (cal is not None)
        feats[name] = stem.forward_clip(clip, lay.img_of, lay.n_img).float()[:, 1:-1, 1:-1, :512].clone()
    off = lambda k: float((feats[k] - feats["ref"]).mean((0, 1, 2)).pow(2).mean().sqrt())
    rms = lambda k: float((feats[k] - feats["ref"]).pow(2).mean().sqrt())
    assert off("coh") < 0.5 * off("rtn"), (off("coh"), off("rtn"))
    assert rms("coh") < 1.05 * rms("rtn"), (rms("coh"), rms("rtn"))


@pytest.mark.parametrize("case", ["film_attn_s196", "film_attn_full", "film_gp_full", "tmh_ragged"])
def test_fp16h_models_match_the_reference_goldens(case):
    """precision='fp16h' on the reference's golden cases (plain feature input: no pair tensors at 8 channels — the trunk's split
    1x1 / fc weights and the loss-scaled fp16 backward are what runs): train-mode logits and gradients at the fp16 precision's
    tolerances."""
    import torch.nn as nn
    model, g = build_product_model(case, "fp16h")
    model.train()
    model.init_hidden()
    v, q = torch.from_numpy(g["v"]).cuda(), torch.from_numpy(g["q"]).cuda()
    out = model(v, q, torch.from_numpy(g["v_lens"]), torch.from_numpy(g["q_lens"]))
    assert rel_err(out.detach().float().cpu().numpy(), g["train_logits"]) < 1.85e-3        # (measured worst 1.40e-3 + 30 %: 8-channel toy nets)
    loss = nn.CrossEntropyLoss(reduction="sum")(out.float(), torch.from_numpy(g["y"]).cuda())
    loss.backward()
    num = den = 0.0
    for name, p in model.named_parameters():
        key = "grad/" + name
        if key in g and p.grad is not None:
            num += float(((p.grad.float().cpu() - torch.from_numpy(g[key])) ** 2).sum())
            den += float((torch.from_numpy(g[key]) ** 2).sum())
    assert den > 0 and (num / den) ** 0.5 < 0.042                                           # (whole-gradient rel. L2: measured worst 0.032 + 30 %)


def _budget_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("error_budget", os.path.join(ROOT, "tools", "error_budget.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_size_errors(seed, data, batches=12, height=224, width=224, calibration="auto", model="film_attn_pt", frames=35):
    bm = _budget_mod()
    args = argparse.Namespace(precision="fp32", model=model, batch=8, frames=frames, height=height, width=width, blocks=1, channels=512,
                              tail_channels=0, seed=seed, calibration=calibration)
    dev = torch.device("cuda", 0)
    d = bm.batches(args, dev, batches, data)
    ref = bm.run(args, "fp32", dev, d)
    got = bm.run(args, "fp16h", dev, d)
    rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(got, ref)]
    flips = sum(int((a.argmax(1) != b.argmax(1)).sum()) for a, b in zip(got, ref))
    print("fp16h vs fp32, weight seed %d, %s clips, %dx%d, %s: max %.3e rms %.3e" %
          (seed, data, height, width, model, max(rel), (sum(r * r for r in rel) / len(rel)) ** 0.5))
    return rel, flips


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fp16h_meets_1e3_on_twelve_full_size_minibatches_for_every_weight_seed(seed):
    """North star's tolerance as stated (logits within 1e-3 of the reference forward, answer classes equal) at BASELINE.json's size:
    precision 'fp16h' against the exact-f32 precision (itself pinned to the oracle at this size, tests/test_gpu_fullsize.py) on twelve
    seeded minibatches (one full-length, eleven ragged) for FOUR sets of random weights.  Round 6 (mean-shifted storage): measured
    0.36 - 0.47e-3 (round 5: 0.67 - 0.89e-3)."""
    rel, flips = _full_size_errors(seed, "noise")
    assert max(rel) <= 1e-3, (seed, ["%.2e" % r for r in rel])
    assert flips == 0, (seed, flips)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fp16h_meets_1e3_at_the_reference_geometry_160x208(seed):
    """... and at the reference's ONLY real geometry (eval/utils.py:24-25: 160 x 208 frames -> 10 x 13 maps; the hard-coded 130 of
    models/film_attn_pt_stem.py:56): rounds 4-5 silently ran plain fp16 here (1.44e-3 on weight seed 3).  Measured 0.27 - 0.52e-3."""
    rel, flips = _full_size_errors(seed, "noise", height=160, width=208)
    assert max(rel) <= 1e-3, (seed, ["%.2e" % r for r in rel])
    assert flips == 0, (seed, flips)


@pytest.mark.parametrize("data,seed", [("smooth", 0), ("blocks", 0), ("blocks", 1), ("blocks", 2), ("blocks", 3), ("textured", 3)])
def test_fp16h_meets_1e3_on_data_that_does_not_look_like_the_calibration_frames(data, seed):
    """The stem's weights and channel means come from seeded SYNTHETIC frames, half uniform noise and half smooth.  'smooth' clips share only
    the kind with that second half; 'blocks' (piecewise-constant images: a background and 24 drifting rectangles, near-identical frames)
    and 'textured' (flat regions x static texture x illumination ramp) are held-out kinds.  Round 5 read 0.96 - 1.18e-3 on 'blocks' (its
    test stated 1.1e-3); the cause was not coherence but the rounding of large per-channel means — with mean-shifted
    storage: 0.46 - 0.55e-3, asserted at the tolerance itself on all four weight seeds."""
    rel, flips = _full_size_errors(seed, data, batches=8)
    assert max(rel) <= 1e-3, (data, seed, ["%.2e" % r for r in rel])
    assert flips == 0


@pytest.mark.parametrize("model,frames", [("time_multi_hop", 70), ("film_gp_pt", 35)])
def test_fp16h_pooling_heads_at_full_size(model, frames):
    """BASELINE.json configs 5 and 3 in the headline precision, asserted AT the tolerance.  A pooling head hands
    single-frame values to its classifier where the attention head averages over frames: 2.5 - 3 x the sensitivity to the same roundings.
    Two weight seeds x three minibatches: the multi-hop model at T = 70 reads 0.44 - 0.65e-3 on the default stem plan (round 5: 0.84 -
    1.25e-3); the global-pooling model reads 0.61 - 1.34e-3 on it and 0.68 - 0.82e-3 with conv22's / conv31's outputs kept as split
    tensors on top of the mean-shifted storage — which its class asks for (`stem_split_depth = 3`, read by bench.build and the CLIs;
    832 instead of 874 clips/s; round 5: 1.26 - 1.81e-3)."""
    worst = 0.0
    for seed in (0, 1):
        rel, flips = _full_size_errors(seed, "noise", batches=3, model=model, frames=frames)
        worst = max(worst, max(rel))
        assert flips == 0, (model, seed, flips)
    assert worst <= 1e-3, (model, worst)


def test_stem_calibration_on_the_deployments_own_frames_is_measured():
    """`--stem_calibration data` / `FrozenStem(calibration=frames)` (only 'it runs' was tested): calibrating the weight rounding
    and the channel means on 40 frames of the DEPLOYMENT's kind (here: 'blocks' clips of other seeds) keeps the held-out kind inside the
    tolerance and is not worse than the synthetic default by more than sampling noise."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    g = torch.Generator().manual_seed(4711)
    clip = bench.blocks_clip(5, 8, 224, 224, g)                               # [5, 3, H, W, 8]: other clips than the test's minibatches
    frames = clip.permute(0, 4, 1, 2, 3).reshape(-1, 3, 224, 224)[:40].contiguous()
    rel_d, flips_d = _full_size_errors(3, "blocks", batches=6, calibration=frames)
    rel_n, _ = _full_size_errors(3, "blocks", batches=6)
    assert max(rel_d) <= 1e-3 and flips_d == 0, ["%.2e" % r for r in rel_d]
    rms = lambda r: (sum(x * x for x in r) / len(r)) ** 0.5
    assert rms(rel_d) <= 1.25 * rms(rel_n), (rms(rel_d), rms(rel_n))


def test_f32_epilogue_rounds_once_after_the_affine():
    """VNQA_CONV_F32_EPILOGUE (round 6): ONE plain output whose ReLU / pool / affine are applied in fp32 and rounded once — the dual
    epilogue's hi half, on both tile families; the plain epilogue's double rounding shows against a float64 reference."""
    from videonavqa_amd import _lib as L
    from videonavqa_amd import kernels as K
    n, h, w, cin, cout = 4, 28, 28, 512, 512
    x = _padded(n, h, w, cin, 51).half()
    g = torch.Generator().manual_seed(52)
    w4 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda().half().float()
    wt = K.pack_conv_weight(w4, torch.float16)
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    sc = torch.ones(cout, device="cuda")
    sh = -(torch.rand(cout, generator=g) * 0.5 + 0.3).cuda()           # a shift that CANCELS most of the value: the case that matters
    ref = _torch_conv(x.float(), w4, bias, True, True, (sc, sh))
    kw = dict(bias=bias, relu=True, pool2=True, post_scale=sc, post_shift=sh)
    for tile in (L.TILE_STEM_PS_224x256, L.TILE_STEM_256x256):
        one = K.conv2d_igemm(x, wt, tile=tile, f32_epilogue=True, **kw)
        two = K.conv2d_igemm(x, wt, tile=tile, dual_out=2, **kw)
        plain = K.conv2d_igemm(x, wt, tile=tile, **kw)
        assert one.shape == plain.shape and torch.equal(one, two[..., :cout])
        e1 = float((one.double()[:, 1:-1, 1:-1] - ref).pow(2).mean().sqrt())
        e2 = float((plain.double()[:, 1:-1, 1:-1] - ref).pow(2).mean().sqrt())
        assert e1 < 0.8 * e2, (tile, e1, e2)
        assert float((one.double()[:, 1:-1, 1:-1] - ref).abs().max()) <= 2.0 ** -11 * float(ref.abs().max()) * 1.01


def test_mean_shifted_storage_is_exact_algebra_and_lowers_the_stem_error():
    """Round 6: the stem's stored activations minus their calibration channel means (FrozenStem._setup_mean_shift).  (a) The algebra is
    exact: the shifted plan reproduces the un-shifted one up to storage roundings — incl. the image border (the halo holds -mu), the
    composed pair's border ring (per-edge bias corrections) and conv1_1's LDS-resident output — at 224 x 224 and at the reference's
    160 x 208; (b) its features are closer to the exact-f32 stem's, on noise frames and — by more — on piecewise-constant frames."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from videonavqa_amd import stem as S
    from videonavqa_amd.models.common import FrameLayout
    torch.set_grad_enabled(False)
    try:
        lay = FrameLayout([2, 1], 2, "cuda")
        for (H, W) in ((224, 224), (160, 208)):
            g = torch.Generator().manual_seed(61)
            clips = {"noise": torch.rand(2, 3, H, W, 2, generator=g).cuda(),
                     "blocks": bench.blocks_clip(2, 2, H, W, g).cuda()}
            vgg32, od32 = _random_stem("fp32")
            ref = S.FrozenStem(vgg32, od32, "fp32")
            vgg, od = _random_stem("fp16")
            S.MEAN_SHIFT = 1
            on = S.FrozenStem(vgg, od, "fp16")
            S.MEAN_SHIFT = 0
            off = S.FrozenStem(vgg, od, "fp16")
            S.MEAN_SHIFT = 1
            assert on.shift and not off.shift and float(on.shift["c22"].abs().max()) > 0
            for kind, clip in clips.items():
                r = ref.forward_clip(clip, lay.img_of, lay.n_img).double()[:, 1:-1, 1:-1, :512]
                fa = on.forward_clip(clip, lay.img_of, lay.n_img)
                assert on.feature_shift is not None and fa.shape[-1] == 512 and off.feature_shift is None
                a = on.plain_features(fa).double()[:, 1:-1, 1:-1, :512]
                b = off.plain_features(off.forward_clip(clip, lay.img_of, lay.n_img)).double()[:, 1:-1, 1:-1, :512]
                scale = float(r.abs().max())
                e_on, e_off = float((a - r).pow(2).mean().sqrt()), float((b - r).pow(2).mean().sqrt())
                # exact algebra: both within a few storage roundings of the exact stem EVERYWHERE (a wrong border / ring / halo term would
                # show as an error of the order of the activations themselves on the border pixels)
                assert float((a - r).abs().max()) < 6e-3 * scale, (H, W, kind, float((a - r).abs().max()) / scale)
                border = torch.zeros(r.shape[1], r.shape[2], dtype=torch.bool, device=r.device)
                border[0, :] = border[-1, :] = border[:, 0] = border[:, -1] = True
                eb = float((a - r)[:, border].pow(2).mean().sqrt())
                ei = float((a - r)[:, ~border].pow(2).mean().sqrt())
                assert eb < 2.0 * ei + 1e-4 * scale, (H, W, kind, eb, ei)
                assert e_on < 0.95 * e_off, (H, W, kind, e_on, e_off)        # (at the FEATURES the later layers' own roundings dilute the gain:
                # measured 0.80-0.85; the logits error halves — tests below, profiles/r06_mean_shift.txt)
    finally:
        S.MEAN_SHIFT = 1
        torch.set_grad_enabled(True)


def test_calibration_without_feature_means_keeps_split_features():
    """A calibration written before round 6 carries the channel means of the stem's inner tensors but not of the FEATURES ('feat'): the
    inner tensors are stored mean-shifted, the features go out as the round-5 split tensor [hi | lo | hi] and conv_init runs its
    three products — every combination stays a supported, tested plan."""
    from videonavqa_amd.models.common import FrameLayout
    from videonavqa_amd.stem import FrozenStem
    torch.set_grad_enabled(False)
    try:
        lay = FrameLayout([2, 1], 2, "cuda")
        clip = torch.rand(2, 3, 224, 224, 2, generator=torch.Generator().manual_seed(71)).cuda()
        vgg32, od32 = _random_stem("fp32")
        ref = FrozenStem(vgg32, od32, "fp32")
        vgg, od = _random_stem("fp16h")
        full = FrozenStem(vgg, od, "fp16h")
        assert full.feature_shift is not None and full.feature_segs == 2 and full.split_active == 0
        # (a means-only calibration: with its "frames" entry the stem would re-run the calibration pass and find the feature means itself)
        old = {k: v for k, v in full.calib.items() if k not in ("feat", "frames")}
        part = FrozenStem(vgg, od, "fp16h", calibration=old)
        assert part.shift and part.feature_shift is None and part.feature_segs == 3 and part.split_active == 1
        r = ref.forward_clip(clip, lay.img_of, lay.n_img).double()[:, 1:-1, 1:-1, :512]
        for st in (full, part):
            f = st.plain_features(st.forward_clip(clip, lay.img_of, lay.n_img)).double()[:, 1:-1, 1:-1, :512]
            assert float((f - r).abs().max()) < 3e-3 * float(r.abs().max())
        plain = FrozenStem(vgg, od, "fp16h", split_features=False)          # consumers that read plain, un-shifted features (MACNetwork)
        assert plain.feature_shift is None and plain.feature_segs == 1
        f = plain.forward_clip(clip, lay.img_of, lay.n_img)
        assert f.shape[-1] == 512 and float(f[:, 0].abs().max()) == 0
        assert float((f.double()[:, 1:-1, 1:-1] - r).abs().max()) < 3e-3 * float(r.abs().max())
    finally:
        torch.set_grad_enabled(True)


@pytest.mark.parametrize("prec,hw", [("fp32", (224, 224)), ("fp32", (160, 208)), ("fp16", (224, 224)), ("fp16", (160, 208))])
def test_composed_edge_border_correction_equals_the_two_step_form(prec, hw):
    """Round 6: the composed pair's border correction as four composed 1x5 / 5x1 edge convs + a corner term (vnqa_conv2d_border_edge_fwd,
    vnqa_ring_assemble_corners) against the two-step form (conv11 on the outside ring, then conv12's edge taps).  Exact-f32 precision: the
    same features to fp32 summation order everywhere incl. the four corner pixels; fp16: within storage roundings, and both equally close
    to the exact stem on the border."""
    from videonavqa_amd import stem as S
    from videonavqa_amd.models.common import FrameLayout
    H, W = hw
    torch.set_grad_enabled(False)
    try:
        lay = FrameLayout([2, 1], 2, "cuda")
        clip = torch.rand(2, 3, H, W, 2, generator=torch.Generator().manual_seed(81)).cuda()
        vgg, od = _random_stem(prec)
        feats = {}
        for flag in (True, False):
            S.RING_COMPOSED_EDGES = flag
            st = S.FrozenStem(vgg, od, prec)
            assert st.composed is not None
            feats[flag] = st.plain_features(st.forward_clip(clip, lay.img_of, lay.n_img)).double()[:, 1:-1, 1:-1, :512].clone()
        S.RING_COMPOSED_EDGES = True
        a, b = feats[True], feats[False]
        scale = float(b.abs().max())
        if prec == "fp32":
            assert float((a - b).abs().max()) < 2e-5 * scale, float((a - b).abs().max()) / scale
        else:
            vgg32, od32 = _random_stem("fp32")
            r = S.FrozenStem(vgg32, od32, "fp32").forward_clip(clip, lay.img_of, lay.n_img).double()[:, 1:-1, 1:-1, :512]
            ea, eb = float((a - r).pow(2).mean().sqrt()), float((b - r).pow(2).mean().sqrt())
            assert ea < 1.1 * eb and float((a - r).abs().max()) < 6e-3 * scale, (ea, eb)
    finally:
        S.RING_COMPOSED_EDGES = True
        torch.set_grad_enabled(True)


@pytest.mark.parametrize("calibration", ["noise", None])
def test_composed_pair_on_2d_patch_stationary_tiles_equals_the_flat_igemm_tile(calibration):
    """Round 6: at 224 x 224 the composed 5x5 runs on the patch-stationary kernel's 8 x 28-pixel tiles (conv_ps_kernel<28, 2, 1>, with the
    per-channel ReLU floor of mean-shifted storage, the border correction and the fused pool) instead of the implicit-GEMM tile's flat 256
    pixels: the same 16-bit operands and fp32 accumulation in another order — the stored 16-bit outputs agree to one rounding step, nearly
    all of them bit for bit.  At 160 x 208 (40 x 52 maps: no whole 2-D tiles) the plan keeps the igemm tile."""
    from videonavqa_amd import stem as S
    from videonavqa_amd.models.common import FrameLayout
    assert S.COMPOSED_PS
    torch.set_grad_enabled(False)
    try:
        lay = FrameLayout([2, 1], 2, "cuda")
        vgg, od = _random_stem("fp16")
        st = S.FrozenStem(vgg, od, "fp16", calibration=calibration)
        assert st.composed is not None and st.composed.get("wt_ps") is not None
        outs = {}
        for hw in ((224, 224), (160, 208)):
            clip = torch.rand(2, 3, hw[0], hw[1], 2, generator=torch.Generator().manual_seed(83)).cuda()
            for flag in (True, False):
                S.COMPOSED_PS = flag
                st._tap = {}
                st.timing = []
                st.forward_clip(clip, lay.img_of, lay.n_img)
                names = [ev[3] for ev in st.timing if len(ev) > 3]
                y = [v for v in st._tap.values() if v.shape[1:] == (hw[0] // 8 + 2, hw[1] // 8 + 2, 512)]
                outs[hw, flag] = (y[0].float().clone(), names)
                st._tap, st.timing = None, None
        S.COMPOSED_PS = True
        a, b = outs[(224, 224), True], outs[(224, 224), False]
        assert "conv_ps_kernel<28,5x5>" in a[1] and "conv_ps_kernel<28,5x5>" not in b[1] and "conv_igemm_kernel" in b[1]
        scale = float(b[0].abs().max())
        d = (a[0] - b[0]).abs()
        assert float(d.max()) <= 2.0 ** -10 * scale, float(d.max()) / scale
        assert float((d > 0).float().mean()) < 0.02
        c, e = outs[(160, 208), True], outs[(160, 208), False]
        assert "conv_ps_kernel<28,5x5>" not in c[1] and torch.equal(c[0], e[0])
    finally:
        S.COMPOSED_PS = True
        torch.set_grad_enabled(True)
