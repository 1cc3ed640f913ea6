"""GPU parity: HIP conv kernels (through the C ABI) vs a plain PyTorch fp32 reference of the same
op on identical seeded inputs.  Tolerances: f32 path 2e-5 relative to max|ref| (exact-f32 MFMA,
summation order differs); bf16 path is checked against the reference evaluated on the SAME
bf16-rounded operands, 1e-2 relative (bf16 output rounding = 2^-9)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, LOW_DTYPE]


def _tol(dt):
    return 2e-5 if dt == torch.float32 else 1e-2


def _q(t, dt):
    """round operands to the kernel's storage type so that only accumulation order differs"""
    return t.to(dt).float()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, taps, relu, pool, post
    (3, 10, 13, 64, 64, 9, True, False, False),
    (2, 14, 14, 128, 192, 9, False, False, False),
    (2, 16, 12, 64, 128, 9, True, True, True),
    (5, 14, 14, 64, 320, 1, True, False, False),
    (1, 28, 28, 256, 512, 9, True, True, False),
    (7, 6, 8, 64, 72, 9, False, False, True),
])
def test_conv_igemm_vs_torch(dt, cfg):
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, taps, relu, pool, post = cfg
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]))
    k = 3 if taps == 9 else 1
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * taps) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() * torch.where(torch.rand(Cout, generator=g) > 0.3, 1.0, -1.0).cuda()
    sh = torch.randn(Cout, generator=g).cuda() * 0.2
    xq, wq = _q(x, dt), _q(w, dt)
    ref = F.conv2d(xq, wq, b, padding=k // 2)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    if post:
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    xn = K.nchw_to_nhwc(x, dt, c_pad=Cin)
    wt = K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin)
    y = K.conv2d_igemm(xn, wt, bias=b, relu=relu, pool2=pool, post_scale=sc if post else None,
                       post_shift=sh if post else None)
    got = K.nhwc_to_nchw(y, Cout)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert _rel(got, ref) < _tol(dt), _rel(got, ref)
    # the halo must stay zero
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0
    assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 18, 19])
def test_conv_igemm_tiles_agree(tile):
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(tile)
    N, H, W, Cin, Cout = 4, 14, 14, 128, 256
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    dt = LOW_DTYPE
    ref = F.relu(F.conv2d(_q(x, dt), _q(w, dt), None, padding=1))
    y = K.conv2d_igemm(K.nchw_to_nhwc(x, dt, c_pad=Cin), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin),
                       relu=True, tile=tile)
    assert _rel(K.nhwc_to_nchw(y, Cout), ref) < 1e-2


@pytest.mark.parametrize("dt", DTYPES)
def test_channel_padding_small_channels(dt):
    """tiny-channel layers (golden configs) run by zero-padding channels to 64."""
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(5)
    N, H, W, Cin, Cout = 3, 10, 13, 8, 12
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(_q(x, dt), _q(w, dt), b, padding=1)
    y = K.conv2d_igemm(K.nchw_to_nhwc(x, dt), K.pack_conv_weight(w, dt), bias=K.pad_vec(b, 64))
    assert y.shape[-1] == 64
    assert _rel(K.nhwc_to_nchw(y, Cout), ref) < _tol(dt)
    assert float(y[..., Cout:].abs().max()) == 0


@pytest.mark.parametrize("dt", DTYPES)
def test_dgrad_via_flipped_weights(dt):
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(9)
    N, H, W, Cin, Cout = 3, 10, 13, 64, 128
    x = torch.randn(N, Cin, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    F.conv2d(x, _q(w, dt), None, padding=1).backward(_q(dy, dt))
    wt_d = K.pack_conv_weight(w, dt, transpose_flip=True, c_out_pad=Cout, c_in_pad=Cin)
    dx = K.conv2d_igemm(K.nchw_to_nhwc(dy, dt, c_pad=Cout), wt_d)
    assert _rel(K.nhwc_to_nchw(dx, Cin), x.grad) < _tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(3, 10, 13, 64, 64, 9), (6, 14, 14, 128, 320, 9), (4, 14, 14, 512, 64, 1),
                                 (40, 14, 14, 256, 256, 9),
                                 # valid-pixel contraction edge cases: maps smaller than one 64-pixel K-step (several image
                                 # wraps per stage), a single row, a single column (falls back to the all-padded-pixels plan)
                                 (37, 3, 5, 64, 64, 9), (70, 1, 7, 64, 128, 9), (9, 6, 1, 64, 64, 9), (150, 2, 2, 64, 64, 9)])
def test_wgrad_vs_torch(dt, cfg):
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, taps = cfg
    k = 3 if taps == 9 else 1
    g = torch.Generator(device="cpu").manual_seed(sum(cfg))
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = torch.zeros(Cout, Cin, k, k).cuda().requires_grad_(True)
    b = torch.zeros(Cout).cuda().requires_grad_(True)
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    F.conv2d(_q(x, dt), w, b, padding=k // 2).backward(_q(dy, dt))
    dwt, dbias = K.conv2d_wgrad(K.nchw_to_nhwc(x, dt, c_pad=Cin), K.nchw_to_nhwc(dy, dt, c_pad=Cout), taps)
    dw = K.unpack_conv_wgrad(dwt, Cout, Cin)
    tol = 2e-5 if dt == torch.float32 else 2e-5  # products of bf16 operands are exact in fp32: order only
    assert _rel(dw, w.grad) < 5e-5, _rel(dw, w.grad)
    assert _rel(dbias, b.grad) < 5e-5


@pytest.mark.parametrize("cfg", [(40, 14, 14, 512, 512, 9, 1), (23, 14, 14, 512, 512, 9, 3), (12, 20, 26, 256, 320, 9, 1), (280, 14, 14, 512, 512, 1, 1),
                                 (5, 7, 7, 128, 64, 9, 1), (3, 40, 52, 64, 128, 9, 1)])
def test_wgrad_four_wave_ring_form_matches_the_eight_wave_form(cfg):
    """The default 16-bit weight-gradient kernel (4 waves, ring of four 32-pixel stages, quadrant-rolling fragment reads) against the
    first form (VNQA_WGRAD_EIGHT_WAVES): same tiles, same split-K slices and slab order — the partial sums differ only by the
    association inside a slice (32- vs 64-pixel MFMA chains are the same chain: bit-identical is expected, 1e-6 is asserted) —
    on 14 x 14 trunk maps (several slices), ragged image counts, widths that do not divide 32, a [hi | lo | hi] x, 1 x 1 taps."""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, taps, segs = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg))
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    xp, dyp = K.nchw_to_nhwc(x, dt, c_pad=Cin), K.nchw_to_nhwc(dy, dt, c_pad=Cout)
    if segs == 3:
        xp = torch.cat([xp, torch.randn_like(xp), xp], dim=-1).contiguous()
    a, ba = K.conv2d_wgrad(xp, dyp, taps, x_segs=segs)
    b, bb = K.conv2d_wgrad(xp, dyp, taps, x_segs=segs, eight_waves=True)
    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), float((a - b).abs().max())
    assert torch.equal(ba, bb)
    k = 3 if taps == 9 else 1
    w = torch.zeros(Cout, Cin, k, k).cuda().requires_grad_(True)
    F.conv2d(_q(x, dt), w, None, padding=k // 2).backward(_q(dy, dt))
    assert _rel(K.unpack_conv_wgrad(a, Cout, Cin), w.grad) < 5e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_first_vs_torch(dt):
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, H, W = 2, 5, 20, 36
    clip = torch.rand(B, 3, H, W, T, generator=g).cuda()
    w = (torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5).cuda()
    b = torch.randn(64, generator=g).cuda() * 0.1
    img_of = torch.full((B * T,), -1, dtype=torch.int32)
    order = [(bb, t) for t in range(T) for bb in range(B) if not (bb == 1 and t >= 3)]
    for n, (bb, t) in enumerate(order):
        img_of[bb * T + t] = n
    y = K.conv_first(clip, w, b, img_of.cuda(), len(order), dt)
    got = K.nhwc_to_nchw(y, 64)
    for n, (bb, t) in enumerate(order):
        ref = F.relu(F.conv2d(_q(clip[bb:bb + 1, :, :, :, t], dt), _q(w, dt), b, padding=1))
        assert _rel(got[n:n + 1], ref) < _tol(dt), (n, _rel(got[n:n + 1], ref))


def test_feat_to_nhwc_roundtrip():
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(4)
    B, C, h, w, T = 3, 8, 10, 13, 6
    v = torch.randn(B, C, h, w, T, generator=g).cuda()
    img_of = torch.arange(B * T, dtype=torch.int32).view(B, T).t().contiguous()  # image = t*B + b
    img_of_bt = torch.empty(B * T, dtype=torch.int32)
    for bb in range(B):
        for t in range(T):
            img_of_bt[bb * T + t] = t * B + bb
    y = K.feat_to_nhwc(v, img_of_bt.cuda(), B * T, torch.float32)
    back = K.nhwc_to_nchw(y, C)  # [T*B, C, h, w]
    ref = v.permute(4, 0, 1, 2, 3).reshape(T * B, C, h, w)
    assert torch.equal(back, ref)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(18, 64, 4096), (280, 128, 8192), (300, 192, 64), (5, 64, 64), (320, 128, 4096), (257, 64, 2048)])
def test_gemm_nt(dt, mnk):
    from videonavqa_amd import kernels as K
    M, N, Kd = mnk
    g = torch.Generator(device="cpu").manual_seed(M + N + Kd)
    a = (torch.randn(M, Kd, generator=g) / Kd ** 0.5).cuda()
    b = torch.randn(N, Kd, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = F.relu(_q(a, dt) @ _q(b, dt).t() + bias)
    out = K.gemm_nt(a.to(dt), b.to(dt), bias=bias, relu=True)
    assert _rel(out.float(), ref) < _tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(64, 4096, 18), (128, 1024, 280), (64, 64, 7), (192, 320, 100)])
def test_gemm_tn(dt, mnk):
    from videonavqa_amd import kernels as K
    M, N, Kd = mnk
    g = torch.Generator(device="cpu").manual_seed(M + N + Kd)
    a = torch.randn(Kd, M, generator=g).cuda()
    b = torch.randn(Kd, N, generator=g).cuda()
    ref = _q(a, dt).t() @ _q(b, dt)
    out = K.gemm_tn(a.to(dt), b.to(dt))
    assert _rel(out, ref) < 5e-5


@pytest.mark.parametrize("tile", [7, 8, 9, 18])
@pytest.mark.parametrize("cfg", [(3, 10, 13, 64, 64, 9, True, False), (2, 16, 12, 64, 128, 9, True, True),
                                 (5, 14, 14, 128, 320, 1, True, False), (9, 28, 28, 64, 64, 9, False, True)])
def test_conv_igemm_ring_pipeline(tile, cfg):
    """4-stage ring / counted-vmcnt main loop: short K loops (1..3 stages), pooling, 1x1, ragged tiles."""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, taps, relu, pool = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]) + tile)
    k = 3 if taps == 9 else 1
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * taps) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    ref = F.conv2d(_q(x, dt), _q(w, dt), b, padding=k // 2)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    y = K.conv2d_igemm(K.nchw_to_nhwc(x, dt, c_pad=Cin), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin),
                       bias=b, relu=relu, pool2=pool, tile=tile)
    assert _rel(K.nhwc_to_nchw(y, Cout), ref) < 1e-2


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cout, relu, pool, post
    (3, 32, 48, 64, True, True, False),
    (2, 16, 16, 128, True, False, False),
    (5, 20, 26, 64, False, False, True),     # ragged tiles (not multiples of 16)
    (300, 16, 32, 64, True, True, True),     # more tiles than workgroups: patch double-buffer ring
    (1, 48, 16, 192, True, True, False),
])
@pytest.mark.parametrize("shape4", [False, True])
def test_conv_c64_direct_vs_torch(cfg, shape4):
    from videonavqa_amd import kernels as K
    N, H, W, Cout, relu, pool, post = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:4]))
    x = torch.randn(N, 64, H, W, generator=g).cuda()
    w = (torch.randn(Cout, 64, 3, 3, generator=g) / 24.0).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda()
    sh = torch.randn(Cout, generator=g).cuda() * 0.2
    ref = F.conv2d(_q(x, dt), _q(w, dt), b, padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    if post:
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    y = K.conv2d_c64(K.nchw_to_nhwc(x, dt, c_pad=64), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=64), bias=b,
                     relu=relu, pool2=pool, post_scale=sc if post else None, post_shift=sh if post else None,
                     shape4=shape4)
    got = K.nhwc_to_nchw(y, Cout)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2, _rel(got, ref)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 15, 18, 19])
def test_conv_igemm_pretiled_weights(dt, tile):
    """pre-tiled (LDS-image) weight layout gives the same result as the row layout"""
    from videonavqa_amd import kernels as K
    if dt == torch.float32 and tile not in (4, 5):
        pytest.skip("f32 instantiates the 128-row tiles only")
    g = torch.Generator(device="cpu").manual_seed(tile)
    N, H, W, Cin, Cout = 3, 14, 14, 128, 320
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda()
    ref = F.relu(F.conv2d(_q(x, dt), _q(w * sc.view(-1, 1, 1, 1), dt), None, padding=1))
    wt = K.pack_conv_weight_tiled(w, dt, tile, out_scale=sc, c_out_pad=Cout, c_in_pad=Cin)
    y = K.conv2d_igemm(K.nchw_to_nhwc(x, dt, c_pad=Cin), wt, relu=True)
    assert _rel(K.nhwc_to_nchw(y, Cout), ref) < _tol(dt)


@pytest.mark.parametrize("tile", [11, 12, 20, 21])
@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, relu, pool, post
    (3, 28, 28, 128, 512, True, True, True),      # 8x28 tiles straddle image boundaries (28 = 3.5 x 8)
    (5, 14, 14, 64, 256, False, False, False),    # 16x14 tiles spanning up to three images; ragged last tile
    (2, 56, 56, 64, 320, True, False, False),     # two column blocks; c_out not a multiple of the 256-wide tile
    (1, 112, 28, 192, 64, True, True, False),     # tall images, three channel chunks (patch double buffer wraps)
    (7, 8, 14, 64, 72, False, True, True),        # H < tile rows
])
def test_conv_patch_tile_vs_torch(tile, cfg):
    """LDS-resident activation patch igemm (conv_patch.hip) vs torch on the same bf16-rounded operands."""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, relu, pool, post = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]) + tile)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() * torch.where(torch.rand(Cout, generator=g) > 0.3, 1.0, -1.0).cuda()
    sh = torch.randn(Cout, generator=g).cuda() * 0.2
    ref = F.conv2d(_q(x, dt), _q(w, dt), b, padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    if post:
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    y = K.conv2d_igemm(K.nchw_to_nhwc(x, dt, c_pad=Cin), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin),
                       bias=b, relu=relu, pool2=pool, post_scale=sc if post else None,
                       post_shift=sh if post else None, tile=tile)
    got = K.nhwc_to_nchw(y, Cout)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2, _rel(got, ref)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0
    assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0
    # bit-identical to the row-tile kernel is not required (different summation order), but it must agree closely
    y1 = K.conv2d_igemm(K.nchw_to_nhwc(x, dt, c_pad=Cin), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin),
                        bias=b, relu=relu, pool2=pool, post_scale=sc if post else None,
                        post_shift=sh if post else None, tile=1)
    assert _rel(y.float(), y1.float()) < 1e-2


@pytest.mark.parametrize("tile", [20, 21])
@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, relu, pool, post: widths that are not a multiple of 14 — the last column block overlaps its neighbour
    (7, 20, 26, 256, 256, True, True, True),      # the reference's 160 x 208 frames: ObjDetectCNN's 20 x 26 maps
    (5, 20, 26, 128, 256, True, False, False),
    (3, 14, 16, 64, 64, False, False, False),     # overlap of 12 columns
    (2, 28, 40, 64, 320, True, True, False),      # three column blocks, the last two overlap by 2
])
def test_conv_ps_tile_overlapping_column_blocks(tile, cfg):
    """conv_ps.hip on widths its 14-column tiles do not divide: vs torch, halo untouched, and against the row-tile igemm."""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, relu, pool, post = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]) + tile)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() * torch.where(torch.rand(Cout, generator=g) > 0.3, 1.0, -1.0).cuda()
    sh = torch.randn(Cout, generator=g).cuda() * 0.2
    ref = F.conv2d(_q(x, dt), _q(w, dt), b, padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    if post:
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    xn, wt = K.nchw_to_nhwc(x, dt, c_pad=Cin), K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin)
    kw = dict(bias=b, relu=relu, pool2=pool, post_scale=sc if post else None, post_shift=sh if post else None)
    y = K.conv2d_igemm(xn, wt, tile=tile, **kw)
    got = K.nhwc_to_nchw(y, Cout)
    assert got.shape == ref.shape and _rel(got, ref) < 1e-2, _rel(got, ref)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0
    assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0
    assert _rel(y.float(), K.conv2d_igemm(xn, wt, tile=1, **kw).float()) < 1e-2


def test_conv_patch_tile_rejects_unsupported_geometry():
    from videonavqa_amd import kernels as K
    from videonavqa_amd._lib import VnqaError
    dt = LOW_DTYPE
    x = torch.zeros(2, 12, 15, 64, dtype=dt, device="cuda")      # 10x13 maps: width not a multiple of 14
    wt = torch.zeros(64, 9, 64, dtype=dt, device="cuda")
    with pytest.raises(VnqaError):
        K.conv2d_igemm(x, wt, tile=11)


@pytest.mark.parametrize("cfg", [
    # B, T, H, W, Cout, pool, ragged
    (2, 3, 32, 48, 64, True, False),
    (1, 2, 40, 24, 128, False, False),     # H, W not multiples of the 16-pixel tile; two 64-channel slices
    (3, 4, 16, 16, 64, True, True),        # ragged frame list (img_of = -1 entries)
])
def test_conv_first_fused_into_c64(cfg):
    """conv1_1 + ReLU evaluated inside the C_in=64 direct kernel (vnqa_clip_to_nhwc4 + vnqa_conv_first_c64_fwd) vs
    torch on the same bf16-rounded operands, and vs the two-launch HIP path."""
    from videonavqa_amd import kernels as K
    B, T, H, W, Cout, pool, ragged = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]))
    clip = torch.rand(B, 3, H, W, T, generator=g).cuda()
    w1 = (torch.randn(64, 3, 3, 3, generator=g) * 0.3).cuda()
    b1 = (torch.randn(64, generator=g) * 0.1).cuda()
    w2 = (torch.randn(Cout, 64, 3, 3, generator=g) / 24.0).cuda()
    b2 = (torch.randn(Cout, generator=g) * 0.1).cuda()
    order = [(b, t) for t in range(T) for b in range(B) if not (ragged and (b + t) % 3 == 2)]
    img_of = torch.full((B * T,), -1, dtype=torch.int32)
    for n, (b, t) in enumerate(order):
        img_of[b * T + t] = n
    img_of = img_of.cuda()
    N = len(order)
    frames = torch.stack([clip[b, :, :, :, t] for b, t in order])                       # [N,3,H,W]
    a1 = F.relu(F.conv2d(_q(frames, dt), _q(w1, dt), b1, padding=1))
    ref = F.relu(F.conv2d(_q(a1, dt), _q(w2, dt), b2, padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    wt = K.pack_conv_weight(w2, dt, c_out_pad=Cout, c_in_pad=64)
    img4 = K.clip_to_nhwc4(clip, img_of, N)
    assert float(img4[:, :2].abs().max()) == 0 and float(img4[:, :, -2:].abs().max()) == 0 and float(img4[..., 3].abs().max()) == 0
    assert torch.equal(img4[:, 2:-2, 2:-2, :3].float(), _q(frames, dt).permute(0, 2, 3, 1))
    y = K.conv_first_c64(img4, w1, b1, wt, bias=b2, relu=True, pool2=pool)
    got = K.nhwc_to_nchw(y, Cout)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2, _rel(got, ref)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0
    assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0
    two = K.conv2d_c64(K.conv_first(clip, w1, b1, img_of, N, dt), wt, bias=b2, relu=True, pool2=pool)
    assert _rel(y.float(), two.float()) < 1e-2


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(16, 8, 10, 13, 64, 64), (128, 512, 14, 14, 512, 128), (20, 70, 5, 4, 128, 24)])
def test_fc_weight_pack_unpack(dt, cfg):
    """vnqa_pack_fc_weight / vnqa_unpack_fc_wgrad vs the permute+pad restatement (exact: pure data movement + rounding)."""
    from videonavqa_amd import kernels as K
    rows, C, h, w, c_pad, rows_pad = cfg
    g = torch.Generator(device="cpu").manual_seed(rows + C)
    wt = torch.randn(rows, C * h * w, generator=g).cuda()
    nat, nat_t = K.pack_fc_weight(wt, C, h, w, c_pad, rows_pad, dt)
    ref = F.pad(wt.view(rows, C, h, w).permute(0, 2, 3, 1), (0, c_pad - C, 1, 1, 1, 1, 0, rows_pad - rows)).reshape(rows_pad, -1)
    assert torch.equal(nat.float(), ref.to(dt).float())
    assert torch.equal(nat_t.float(), ref.to(dt).float().t())
    gnat = torch.randn(rows_pad, (h + 2) * (w + 2) * c_pad, generator=g).cuda()
    back = K.unpack_fc_wgrad(gnat, rows, C, h, w, c_pad)
    ref_b = gnat.view(rows_pad, h + 2, w + 2, c_pad)[:rows, 1:-1, 1:-1, :C].permute(0, 3, 1, 2).reshape(rows, -1)
    assert torch.equal(back, ref_b)


@pytest.mark.parametrize("cfg", [(280, 256 * 512), (7, 128), (320, 5 * 128), (33, 300 * 128), (321, 3 * 128), (1120, 40 * 128)])
def test_fc_dx_matches_generic_gemm(cfg):
    """vnqa_fc_dx (dX of fc_embed_attn from the forward operand, resident dout + transposed LDS reads) against the generic
    path it replaces (vnqa_gemm_nt on the transposed copy: same 16-bit operands, fp32 accumulation over 128 terms, one
    rounding) and against torch's fp32 product of the same rounded operands."""
    from videonavqa_amd import kernels as K
    from helpers import LOW_DTYPE
    m, kn = cfg
    g = torch.Generator(device="cpu").manual_seed(m + kn)
    dout = torch.randn(m, 128, generator=g).cuda().to(LOW_DTYPE)
    nat = (torch.randn(128, kn, generator=g) * 0.1).cuda().to(LOW_DTYPE)
    assert K.fc_dx_supported(m, 128, kn, LOW_DTYPE)
    dx = K.fc_dx(dout, nat)
    ref = dout.float() @ nat.float()
    assert dx.shape == ref.shape
    assert _rel(dx.float(), ref) < 6e-3, _rel(dx.float(), ref)
    old = K.gemm_nt(dout, nat.t().contiguous())
    # both accumulate the same 128 products in fp32 (in different orders) and round once: at most one ulp apart
    ulp = 2.0 ** -7 if LOW_DTYPE == torch.bfloat16 else 2.0 ** -10
    assert float(((dx.float() - old.float()).abs() / (ref.abs() + 1e-3)).max()) <= 2 * ulp
    assert not K.fc_dx_supported(m, 64, kn, LOW_DTYPE) and not K.fc_dx_supported(m, 128, kn + 64, LOW_DTYPE)
    assert not K.fc_dx_supported(m, 128, kn, torch.float32)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 6, 8, 64, 128, 192), (3, 14, 12, 128, 512, 512), (1, 4, 4, 64, 64, 64)])
def test_composed_conv_pair_matches_two_step(dt, cfg):
    """FrozenStem._compose_pair: conv(3x3) -> conv(3x3) + eval BN + ReLU + pool as ONE 5x5 igemm launch with the exact
    border correction, vs torch running the two convolutions (fp32: 1e-4; bf16 operands: 2e-2)."""
    import torch.nn as nn
    from videonavqa_amd import kernels as K
    from videonavqa_amd.stem import FrozenStem
    N, H, W, Ci, Cm, Co = cfg
    torch.manual_seed(sum(cfg))
    c1, c2, bn = nn.Conv2d(Ci, Cm, 3, padding=1), nn.Conv2d(Cm, Co, 3, padding=1), nn.BatchNorm2d(Co)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
        c1.bias.normal_(0, 0.5)
    c1, c2, bn = c1.cuda(), c2.cuda(), bn.cuda().eval()
    x = torch.randn(N, Ci, H, W).cuda()
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(bn(c2(c1(x)))), 2, 2)
    stem = FrozenStem(None, None, LOW if dt == LOW_DTYPE else "fp32")
    stem.composed = stem._compose_pair(c1, c2, bn)
    xn = F.pad(K.nchw_to_nhwc(x, dt, c_pad=Ci), (0, 0, 1, 1, 1, 1))          # halo 2
    with torch.no_grad():
        y = stem._run_composed(xn, ("t",))
    got = K.nhwc_to_nchw(y, Co)
    assert got.shape == ref.shape
    tol = 1e-4 if dt == torch.float32 else 2e-2
    assert _rel(got, ref) < tol, _rel(got, ref)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0
    # the border really needs the correction: without it the ring of output pixels is wrong
    with torch.no_grad():
        cp = stem.composed
        bad = K.conv2d_igemm(xn, cp["wt"], bias=cp["bias"], relu=True, pool2=True, x_halo=2, y_halo=1, tile=cp["tile"])
    assert _rel(K.nhwc_to_nchw(bad, Co), ref) > 5 * tol


@pytest.mark.parametrize("dt", DTYPES)
def test_grouped_gemm_and_all_edge_gather_match_per_edge_calls(dt):
    """vnqa_gemm_nt_grouped / vnqa_ring_edge_gather_all (one launch for the four edge products of the composed conv's
    border correction) against the per-edge entry points: bit-identical rows."""
    from videonavqa_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(11)
    n, H, W, c, co = 3, 6, 9, 64, 72
    R = 2 * (W + 2) + 2 * H
    y1 = torch.randn(n * R, c, generator=g).cuda().to(dt)
    wts = (torch.randn(4, co, 3 * c, generator=g) / (3 * c) ** 0.5).cuda().to(dt)
    ops = K.ring_edge_gather_all(y1, n, H, W)
    res = K.gemm_nt_grouped(ops, wts)
    for e in range(4):
        ln = W if e < 2 else H
        one = K.ring_edge_gather(y1, n, H, W, e)
        assert torch.equal(ops[e, :n * ln], one)
        ref = K.gemm_nt(one, wts[e].contiguous(), split_k=False)
        assert torch.equal(res[e, :n * ln], ref)
        exact = one.float() @ wts[e].float().t()
        assert _rel(res[e, :n * ln].float(), exact) < _tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_ring_conv_implicit_gemm_matches_im2col_gemm(dt):
    """conv3x3 at the outside-ring positions of halo-2 images (the first operand of the composed conv's border correction):
    the implicit-GEMM form (vnqa_conv2d_ring_fwd) against the materialised im2col + GEMM it replaces."""
    from videonavqa_amd import kernels as K
    torch.manual_seed(3)
    n, H, W, ci, co = 5, 10, 14, 128, 192
    x = torch.zeros(n, H + 4, W + 4, ci)
    x[:, 2:-2, 2:-2] = torch.randn(n, H, W, ci)
    x = x.to(dt).cuda()
    w = (torch.randn(co, ci, 3, 3) * 0.05).cuda()
    b = torch.randn(co).cuda()
    wt = K.pack_conv_weight(w, dt)                                   # [co][9][ci]
    ref = K.gemm_nt(K.ring_im2col(x, H, W), wt.view(co, -1), bias=b, split_k=False).float()
    got = K.conv2d_ring(x, wt, b, H, W).float()
    assert got.shape == ref.shape == (n * (2 * (W + 2) + 2 * H), co)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())
    # and against torch: conv evaluated on the (H+2)x(W+2) grid, ring positions picked out
    xin = x[:, 1:-1, 1:-1].float().permute(0, 3, 1, 2)              # halo-1 view: a 'valid' conv gives the (H+2)x(W+2)... no:
    full = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(dt).float(), b)   # valid conv over the halo-2 image: [n,co,H+2,W+2]
    ring = torch.cat([full[:, :, 0, :], full[:, :, -1, :], full[:, :, 1:-1, 0], full[:, :, 1:-1, -1]], 2)   # top, bottom, left, right
    ring = ring.permute(0, 2, 1).reshape(-1, co)
    assert float((got - ring).abs().max()) <= (1e-4 if dt == torch.float32 else 2e-2) * float(ring.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(5, 10, 14), (3, 56, 56), (70, 6, 10), (2, 40, 52)])
def test_ring_conv_edge_by_edge_with_three_taps_gives_the_one_launch_forms_bits(dt, geom):
    """vnqa_conv2d_ring_edge_fwd x 4 (each edge with the three taps that can see the image: 1x3 along image row 0 / H-1, 3x1 down image
    column 0 / W-1) against vnqa_conv2d_ring_fwd (nine taps, six of them in the zero halo): the same sums in the same order — bit
    for bit — and the separator rows of the padded layout stay untouched."""
    from videonavqa_amd import kernels as K
    n, H, W = geom
    ci, co = 128, 192
    g = torch.Generator().manual_seed(n * 7 + H)
    x = torch.zeros(n, H + 4, W + 4, ci)
    x[:, 2:-2, 2:-2] = torch.randn(n, H, W, ci, generator=g)
    x = x.to(dt).cuda()
    wt = K.pack_conv_weight((torch.randn(co, ci, 3, 3, generator=g) * 0.05).cuda(), dt)        # [co][9][ci]
    b = torch.randn(co, generator=g).cuda()
    R = 2 * (W + 2) + 2 * H
    one = torch.zeros(n, R + 4, co, dtype=dt, device="cuda")
    K.conv2d_ring(x, wt, b, H, W, out_padded=one)
    four = torch.zeros(n, R + 4, co, dtype=dt, device="cuda")
    edges = K.ring_edge_weights(wt)
    assert [tuple(e.shape) for e in edges] == [(co, 3, ci)] * 4
    K.conv2d_ring_edges(x, edges, b, H, W, four)
    assert torch.equal(four, one)
    for z in (2 * (W + 2), 2 * (W + 2) + H + 1, 2 * (W + 2) + H + 2, R + 3):
        assert float(four[:, z].abs().max()) == 0


@pytest.mark.parametrize("dt", DTYPES)
def test_ring_edge_convs_match_gathered_edge_gemms(dt):
    """The four edge products of the border correction as implicit 1x3 convs along the zero-separated ring layout against
    the gathered-operand GEMMs they replace (incl. the corner slots that must read zeros on the left / right columns)."""
    from videonavqa_amd import kernels as K
    torch.manual_seed(4)
    n, H, W, ci, cm, co = 3, 6, 10, 64, 128, 64
    x = torch.zeros(n, H + 4, W + 4, ci)
    x[:, 2:-2, 2:-2] = torch.randn(n, H, W, ci)
    x = x.to(dt).cuda()
    wt = K.pack_conv_weight((torch.randn(cm, ci, 3, 3) * 0.1).cuda(), dt)
    b = torch.randn(cm).cuda()
    R = 2 * (W + 2) + 2 * H
    y1 = K.conv2d_ring(x, wt, b, H, W)                                           # [n*R, cm]
    y1p = torch.zeros(n, R + 4, cm, dtype=dt, device="cuda")
    K.conv2d_ring(x, wt, b, H, W, out_padded=y1p)
    flat = y1.view(n, R, cm)
    assert torch.equal(y1p[:, :2 * (W + 2)], flat[:, :2 * (W + 2)])
    assert torch.equal(y1p[:, 2 * (W + 2) + 1:2 * (W + 2) + 1 + H], flat[:, 2 * (W + 2):2 * (W + 2) + H])
    assert torch.equal(y1p[:, 2 * (W + 2) + H + 3:2 * (W + 2) + 2 * H + 3], flat[:, 2 * (W + 2) + H:])
    for z in (2 * (W + 2), 2 * (W + 2) + H + 1, 2 * (W + 2) + H + 2, R + 3):
        assert float(y1p[:, z].abs().max()) == 0                                # separator rows never written
    for e in range(4):
        we = (torch.randn(co, 3 * cm) * 0.05).to(dt).cuda()
        ref = K.gemm_nt(K.ring_edge_gather(y1, n, H, W, e), we, split_k=False).float()
        got = K.ring_edge_conv(y1p, we, H, W, e).float()
        tol = 1e-5 if dt == torch.float32 else 1e-2
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= tol * float(ref.abs().max()), e


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, pool, post, y_halo      (the three geometries vnqa_conv2d_wreg_fwd serves)
    (3, 16, 32, 128, 128, True, True, 2),       # conv2_2 (+ bn_input affine, halo-2 output for the composed 5x5 conv)
    (2, 8, 16, 128, 128, True, False, 1),
    (5, 24, 48, 64, 128, False, False, 1),      # conv2_1
    (3, 32, 32, 64, 64, True, False, 1),        # conv1_2
    (70, 16, 16, 128, 128, True, True, 1),      # more tiles than one round of workgroups... 
    (300, 16, 32, 64, 128, False, True, 1),     # several rounds: the patch double buffer wraps many times
    (150, 32, 32, 64, 64, True, True, 1),
    # widths that are not a multiple of 16: the last tile of a row overlaps its neighbour (80 x 104 = the reference's 160 x 208 frames)
    (6, 16, 104, 128, 128, True, True, 2),
    (5, 16, 104, 64, 128, False, True, 1),
    (4, 16, 24, 128, 128, True, False, 1),
    (3, 8, 40, 64, 128, False, False, 1),
    (3, 32, 40, 64, 64, True, False, 1),
    (2, 8, 18, 64, 128, False, False, 1),
    (40, 80, 104, 64, 128, False, False, 1),
])
def test_conv_wreg_vs_torch(cfg):
    """weights-in-registers persistent direct conv (csrc/conv_wreg.hip) vs torch on the same storage-rounded operands, and
    bit-for-bit against the implicit-GEMM kernel (same products, fp32 accumulation order differs only inside the MFMA K)"""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, pool, post, yh = cfg
    dt = LOW_DTYPE
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]) + yh)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() * 0.1
    sc = ((torch.rand(Cout, generator=g) + 0.5) * torch.where(torch.rand(Cout, generator=g) > 0.3, 1.0, -1.0)).cuda()
    sh = torch.randn(Cout, generator=g).cuda() * 0.2
    ref = F.relu(F.conv2d(_q(x, dt), _q(w, dt), b, padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    if post:
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    xn = K.nchw_to_nhwc(x, dt, c_pad=Cin)
    wt = K.pack_conv_weight(w, dt, c_out_pad=Cout, c_in_pad=Cin)
    assert K.conv2d_wreg_supported(xn, wt, pool2=pool, y_halo=yh)
    y = K.conv2d_wreg(xn, wt, bias=b, relu=True, pool2=pool, post_scale=sc if post else None,
                      post_shift=sh if post else None, y_halo=yh)
    inner = y[:, yh - 1:y.shape[1] - (yh - 1), yh - 1:y.shape[2] - (yh - 1)].contiguous()
    got = K.nhwc_to_nchw(inner, Cout)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-2, _rel(got, ref)
    # the halo ring is never written
    assert float(y[:, :yh].abs().max()) == 0 and float(y[:, -yh:].abs().max()) == 0
    assert float(y[:, :, :yh].abs().max()) == 0 and float(y[:, :, -yh:].abs().max()) == 0
    # against the igemm kernel on the same packed operands (halo 1 form)
    y2 = K.conv2d_igemm(xn, wt, bias=b, relu=True, pool2=pool, post_scale=sc if post else None,
                        post_shift=sh if post else None)
    d = (inner.float() - y2.float()).abs().max() / (y2.float().abs().max() + 1e-12)
    # (with an affine the igemm tile rounds to storage, applies it and rounds again; this kernel applies it in fp32 and rounds once —
    # round 6, tools/stem_layer_errors.py: up to two storage roundings apart.  Without an affine the two agree to the MFMA's K order.)
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert float(d) < (4e-3 if not post else max(4e-3, 2.5 * ulp)), float(d)


def test_conv_wreg_unsupported_geometry_is_refused():
    from videonavqa_amd import kernels as K
    from videonavqa_amd import _lib as L
    dt = LOW_DTYPE
    x = torch.zeros(1, 12, 22, 128, dtype=dt, device="cuda")       # 10 x 20: not whole 8 x 16 tiles
    wt = torch.zeros(128, 9, 128, dtype=dt, device="cuda")
    assert not K.conv2d_wreg_supported(x, wt, pool2=True)
    with pytest.raises(L.VnqaError):
        K.conv2d_wreg(x, wt, relu=True, pool2=True)
