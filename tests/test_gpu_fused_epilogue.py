"""GPU: the igemm's fused trunk epilogues (BIAS_RELU_BNSTATS / BIAS_FILM_RELU_RES) against the separate
elementwise kernels they replace, and the one-node trunk (ops.FilmTrunkFn) against the op-by-op autograd graph."""
import os

import pytest
import torch

from helpers import LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


def _padded(n, h, w, c, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(n, h + 2, w + 2, c)
    x[:, 1:-1, 1:-1] = torch.randn(n, h, w, c, generator=g) * scale
    return x.to(dtype).cuda()


@pytest.mark.parametrize("dtype", [LOW_DTYPE, torch.float32])
@pytest.mark.parametrize("geom", [(8, 14, 14, [3, 2, 2, 1]), (5, 10, 13, [1, 1, 1, 1, 1]), (12, 14, 14, [8, 4])])
def test_conv_bnstats_epilogue_matches_separate_stats(dtype, geom):
    """conv + ReLU with the per-frame statistics taken in the epilogue: output bit-identical to the plain conv, mean/var
    equal to vnqa_frame_bn_stats on that output (frames of 1..8 images: tiles straddle 2 and 3 frames)."""
    from videonavqa_amd import kernels as K
    n_img, h, w, cts = geom
    assert sum(cts) == n_img
    C_in, C_out = 64, 128
    x = _padded(n_img, h, w, C_in, dtype, 1)
    g = torch.Generator().manual_seed(2)
    wgt = torch.randn(C_out, C_in, 3, 3, generator=g).cuda() * 0.05
    bias = torch.randn(C_out, generator=g).cuda()
    wt = K.pack_conv_weight(wgt, dtype)
    off = [0]
    for ct in cts:
        off.append(off[-1] + ct)
    frame_off = torch.tensor(off, dtype=torch.int32, device="cuda")
    frame_of = torch.tensor([f for f, ct in enumerate(cts) for _ in range(ct)], dtype=torch.int32, device="cuda")
    ref = K.conv2d_igemm(x, wt, bias=bias, relu=True)
    m_ref, v_ref = K.frame_bn_stats(ref, frame_off, len(cts))
    got = K.conv2d_igemm_bnstats(x, wt, bias, True, frame_of, frame_off, len(cts), min(cts))
    assert got is not None
    y, mean, var = got
    assert torch.equal(y, ref)
    assert float((mean - m_ref).abs().max()) < 1e-5 * float(m_ref.abs().max()) + 1e-6
    assert float((var - v_ref).abs().max()) < 1e-4 * float(v_ref.abs().max()) + 1e-6
    y2, mean2, var2 = K.conv2d_igemm_bnstats(x, wt, bias, True, frame_of, frame_off, len(cts), min(cts))
    assert torch.equal(mean, mean2) and torch.equal(var, var2)                # no atomics: run-to-run identical


def test_conv_bnstats_reports_unsupported_for_tiny_frames():
    from videonavqa_amd import kernels as K
    x = _padded(6, 4, 6, 64, LOW_DTYPE, 1)              # 24 pixels per image: a 256-pixel tile would span > 3 frames
    wt = K.pack_conv_weight(torch.randn(64, 64, 3, 3).cuda(), LOW_DTYPE)
    frame_of = torch.arange(6, dtype=torch.int32, device="cuda")
    frame_off = torch.arange(7, dtype=torch.int32, device="cuda")
    assert K.conv2d_igemm_bnstats(x, wt, None, True, frame_of, frame_off, 6, 1) is None


@pytest.mark.parametrize("dtype", [LOW_DTYPE, torch.float32])
@pytest.mark.parametrize("C", [128, 40])
def test_conv_film_res_epilogue_is_bit_identical_to_separate_kernels(dtype, C):
    """3x3 conv + FiLM affine + ReLU + residual in one launch vs conv then vnqa_film_relu_res_fwd; gamma/beta are column
    slices of a wider matrix (row stride 2*C*blocks); C = 40 exercises the channel padding (c_pad = 64)."""
    from videonavqa_amd import _lib as L, kernels as K
    n_img, h, w = 9, 14, 14
    c_pad = L.round_up(C, 64)
    res = _padded(n_img, h, w, c_pad, dtype, 3)
    res[..., C:] = 0
    g = torch.Generator().manual_seed(4)
    wgt = torch.randn(C, C, 3, 3, generator=g).cuda() * 0.05
    bias = torch.randn(C, generator=g).cuda()
    film = torch.randn(n_img, 6 * C, generator=g).cuda().relu()          # 3 blocks' worth; use block 1
    gamma, beta = film[:, 2 * C:3 * C], film[:, 3 * C:4 * C]
    wt = K.pack_conv_weight(wgt, dtype, c_out_pad=c_pad, c_in_pad=c_pad)
    b = K.pad_vec(bias, c_pad)
    z_ref = K.conv2d_igemm(res, wt, bias=b)
    pad = lambda t: torch.nn.functional.pad(t, (0, c_pad - C)).contiguous()
    out_ref = K.film_relu_res_fwd(z_ref, res, pad(gamma), pad(beta))
    z, out = K.conv2d_igemm_film_res(res, wt, b, gamma, beta, C, res)
    assert torch.equal(z, z_ref) and torch.equal(out, out_ref)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0     # zero halo kept
    # backward on the slices: gradients land in the same column range of a zero matrix
    dout = _padded(n_img, h, w, c_pad, dtype, 5)
    dz_ref, dg_ref, db_ref = K.film_relu_res_bwd(dout, z_ref, pad(gamma), pad(beta))
    dfilm = torch.zeros_like(film)
    dz = K.film_relu_res_bwd_ld(dout, z, gamma, beta, C, dfilm[:, 2 * C:3 * C], dfilm[:, 3 * C:4 * C])
    assert torch.equal(dz, dz_ref)
    assert torch.equal(dfilm[:, 2 * C:3 * C], dg_ref[:, :C]) and torch.equal(dfilm[:, 3 * C:4 * C], db_ref[:, :C])
    assert float(dfilm[:, :2 * C].abs().max()) == 0 and float(dfilm[:, 4 * C:].abs().max()) == 0


@pytest.mark.parametrize("precision", ["fp32", LOW])
@pytest.mark.parametrize("kind", ["film_attn", "tmh"])
def test_one_node_trunk_matches_op_by_op_graph(precision, kind, monkeypatch):
    """Same model, same inputs: VNQA_FUSED_TRUNK=1 (ops.FilmTrunkFn, fused epilogues) vs =0 (one autograd node per op).
    14x14 maps, 64 channels, ragged clips: logits, BN running statistics and every parameter gradient must agree."""
    import videonavqa_amd.models as M
    torch.manual_seed(0)
    B, T, Cin, C = 4, 5, 64, 64
    kw = dict(num_input_channels=Cin, num_res_block_channels=C, num_res_blocks=2, hidden_size=16, vocab_size=20)
    if kind == "film_attn":
        model = M.FiLMAttnPretrainedStem(B, 12, 7, at_hidden_size=16, max_num_frames=T, spatial_size=196,
                                         precision=precision, **kw).cuda()
    else:
        model = M.TimeMultiHopFiLMPretrainedStem(B, 12, 7, num_tail_channels=4, spatial_size=196, precision=precision,
                                                 **kw).cuda()
    g = torch.Generator().manual_seed(1)
    v = torch.rand(B, Cin, 14, 14, T, generator=g).cuda()
    q = torch.randint(1, 20, (B, 9), generator=g).cuda()
    v_lens, q_lens = torch.tensor([5, 4, 2, 2]), torch.tensor([9, 3, 5, 7])
    y = torch.randint(0, 7, (B,), generator=g).cuda()
    results = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("VNQA_FUSED_TRUNK", flag)
        model.train()
        model.zero_grad()
        model.bn_init.reset_running_stats()
        model.init_hidden()
        logits = model(v, q, v_lens, q_lens)
        torch.nn.functional.cross_entropy(logits, y, reduction="sum").backward()
        results[flag] = (logits.detach().clone(), model.bn_init.running_mean.clone(), model.bn_init.running_var.clone(),
                         {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    a, b = results["0"], results["1"]
    # fp32: the same kernels in the same order; bf16: the statistics come from one-pass sums in the conv epilogue
    tol = 1e-5 if precision == "fp32" else 2e-3
    assert float((a[0] - b[0]).abs().max()) <= tol * float(a[0].abs().max())
    assert float((a[1] - b[1]).abs().max()) <= tol * float(a[1].abs().max()) + 1e-7
    assert float((a[2] - b[2]).abs().max()) <= tol * float(a[2].abs().max()) + 1e-7
    assert set(a[3]) == set(b[3])
    gmax = max(float(t.abs().max()) for t in a[3].values())
    for n in a[3]:
        scale = float(a[3][n].abs().max())
        # (analytically zero gradients, e.g. fc_hidden_attn of the attention model, are float noise: absolute floor)
        assert float((a[3][n] - b[3][n]).abs().max()) <= (1e-4 if precision == "fp32" else 3e-2) * scale + 1e-6 * gmax, n


@pytest.mark.parametrize("dtype", [LOW_DTYPE, torch.float32])
def test_conv_add_mask_epilogue_is_bit_identical_to_conv_plus_relu_bwd(dtype):
    """dgrad + residual join + ReLU mask in one launch (VNQA_EPI_ADD_MASK) vs conv2d_igemm followed by relu_bwd(a, y, b)."""
    from videonavqa_amd import kernels as K
    n_img, h, w, C = 7, 14, 14, 128
    dz = _padded(n_img, h, w, C, dtype, 11)
    dout = _padded(n_img, h, w, C, dtype, 12)
    res = _padded(n_img, h, w, C, dtype, 13).relu()
    g = torch.Generator().manual_seed(14)
    wt = K.pack_conv_weight((torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda(), dtype, transpose_flip=True)
    ref = K.relu_bwd(K.conv2d_igemm(dz, wt), res, dout)
    got = K.conv2d_igemm_add_mask(dz, wt, dout, res)
    assert torch.equal(got, ref)
    assert float(got[:, 0].abs().max()) == 0 and float(got[:, :, -1].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [LOW_DTYPE, torch.float32])
@pytest.mark.parametrize("cfg", [(5, 14, 14, 64, 128, 9, False), (3, 10, 13, 64, 64, 1, False), (4, 12, 16, 64, 192, 9, True),
                                 (7, 6, 6, 128, 64, 9, False)])
def test_conv_zeroes_the_halo_of_a_poisoned_output_buffer(dt, cfg):
    """VNQA_CONV_ZERO_HALO: fresh conv outputs come from torch.empty and the kernel writes the halo ring itself.  The caching
    allocator is primed with NaN-filled blocks of exactly the output's size, so a halo position the kernel misses shows up as
    NaN; interior values must equal the same conv into a pre-zeroed buffer (the path without the flag)."""
    from videonavqa_amd import kernels as K
    N, H, W, Cin, Cout, taps, pool = cfg
    g = torch.Generator(device="cpu").manual_seed(sum(cfg[:5]))
    x = torch.zeros(N, H + 2, W + 2, Cin, dtype=dt, device="cuda")
    x[:, 1:-1, 1:-1] = torch.randn(N, H, W, Cin, generator=g).cuda().to(dt)
    k = 3 if taps == 9 else 1
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).cuda()
    wt = K.pack_conv_weight(w, dt)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = K.conv2d_igemm(x, wt, relu=True, pool2=pool, out=torch.zeros(N, Ho + 2, Wo + 2, Cout, dtype=dt, device="cuda"))
    for _ in range(3):
        junk = [torch.full((N, Ho + 2, Wo + 2, Cout), float("nan"), dtype=dt, device="cuda") for _ in range(2)]
        del junk
        y = K.conv2d_igemm(x, wt, relu=True, pool2=pool)
        assert torch.isfinite(y.float()).all()
        assert torch.equal(y, ref)
        assert float(y[:, 0].abs().max()) == 0 and float(y[:, -1].abs().max()) == 0
        assert float(y[:, :, 0].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(9, 14, 14, 256), (5, 28, 28, 256), (35, 14, 14, 512), (6, 20, 26, 256)])
def test_patch_stationary_tile_carries_film_res_and_add_mask(geom):
    """The FiLM block's two fused 3x3 launches on the patch-stationary kernel (tile id 20, what the train-mode trunk uses at
    14x14 / 28x28 maps): bit-identical to the igemm forms — z, the FiLM+ReLU+residual output, the masked dgrad sum — and the halo
    ring of the torch.empty outputs is zeroed (allocator primed with NaN blocks of the same size)."""
    from videonavqa_amd import _lib as L, kernels as K
    if not L.is_half(LOW_DTYPE):
        pytest.skip("16-bit storage only")
    n_img, h, w, C = geom
    res = _padded(n_img, h, w, C, LOW_DTYPE, 3).relu()
    g = torch.Generator().manual_seed(4)
    wt = K.pack_conv_weight(torch.randn(C, C, 3, 3, generator=g).cuda() * 0.03, LOW_DTYPE)
    b = torch.randn(C, generator=g).cuda()
    film = torch.randn(n_img, 4 * C, generator=g).cuda().relu()
    gamma, beta = film[:, C:2 * C], film[:, 2 * C:3 * C]
    assert K.ps_fused_tile(res) == L.TILE_PS_224x256
    z_ref, out_ref = K.conv2d_igemm_film_res(res, wt, b, gamma, beta, C, res)
    dz = _padded(n_img, h, w, C, LOW_DTYPE, 11)
    dout = _padded(n_img, h, w, C, LOW_DTYPE, 12)
    sum_ref = K.conv2d_igemm_add_mask(dz, wt, dout, res)
    poison = [torch.full_like(res, float("nan")) for _ in range(4)]
    del poison
    z, out = K.conv2d_igemm_film_res(res, wt, b, gamma, beta, C, res, tile=L.TILE_PS_224x256)
    got = K.conv2d_igemm_add_mask(dz, wt, dout, res, tile=L.TILE_PS_224x256)
    assert torch.equal(z, z_ref) and torch.equal(out, out_ref) and torch.equal(got, sum_ref)
    for t in (z, out, got):
        assert float(t[:, 0].abs().max()) == 0 and float(t[:, -1].abs().max()) == 0
        assert float(t[:, :, 0].abs().max()) == 0 and float(t[:, :, -1].abs().max()) == 0
