"""GPU: the small fp32 HIP ops of the question path / classifier / loss (csrc/glue.hip, packed temporal attention)
against plain PyTorch fp32 references of the same ops."""
import pytest
import torch

import torch_partners
import torch.nn as nn
import torch.nn.functional as F

from helpers import LOW, LOW_DTYPE

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5):
    return float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-7


@pytest.mark.parametrize("m,n,k", [(280, 1024, 128), (8, 70, 4480), (3, 5, 7), (130, 65, 33)])
def test_sgemm_nt_nn_tn_bias_relu_mask_gather_scatter(m, n, k):
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g).cuda()
    w = torch.randn(n, k, generator=g).cuda()
    b = torch.randn(n, generator=g).cuda()
    assert _close(K.linear_nt(x, w, bias=b, relu=True), F.relu(x @ w.t() + b))
    assert _close(K.linear_nt(x, w), x @ w.t())
    dy = torch.randn(m, n, generator=g).cuda()
    mask = torch.randn(m, n, generator=g).cuda()
    dym = dy * (mask > 0)
    assert _close(K.matmul_nn(dy, w, a_mask=mask), dym @ w)
    assert _close(K.matmul_tn(dy, x, a_mask=mask), dym.t() @ x)
    assert _close(K.colsum(dy, mask), dym.sum(0)) and _close(K.colsum(dy), dy.sum(0))
    assert _close(K.colsum(dy.to(LOW_DTYPE)), dy.to(LOW_DTYPE).float().sum(0), 1e-5)
    # row gather on A (with a negative = zero row) and row scatter on C, accumulation
    rows = torch.randperm(m, generator=g)[: max(m // 2, 1)].to(torch.int32)
    rows_neg = rows.clone()
    rows_neg[0] = -1
    xs = x[rows.long()].clone()
    xs[0] = 0
    assert _close(K.linear_nt(x, w, bias=b, a_rows=rows_neg.cuda(), m=rows.numel()), xs @ w.t() + b)
    out = torch.zeros(m, k, device="cuda")
    K.matmul_nn(dy[: rows.numel()].contiguous(), w, out=out, c_rows=rows.cuda())
    ref = torch.zeros(m, k, device="cuda")
    ref[rows.long().cuda()] = dy[: rows.numel()] @ w
    assert _close(out, ref)
    acc = torch.ones(n, k, device="cuda")
    K.matmul_tn(dy, x, out=acc, accumulate=True)
    assert _close(acc, 1 + dy.t() @ x)
    # strided operand views (column slices of wider matrices)
    wide = torch.randn(m, k + 9, generator=g).cuda()
    assert _close(K.linear_nt(wide[:, 4:4 + k], w), wide[:, 4:4 + k] @ w.t())


@pytest.mark.parametrize("padding_idx", [None, 0])
def test_embed_proj_forward_backward_vs_torch(padding_idx):
    from videonavqa_amd import ops
    torch.manual_seed(3)
    V, E, H, B, Lq = 23, 12, 16, 5, 9
    emb = nn.Embedding(V, E, padding_idx=padding_idx).cuda()
    lstm = nn.LSTM(E, H).cuda()
    tokens = torch.randint(0, V, (B, Lq)).cuda()
    tokens[0, 3] = 0
    ref = F.linear(emb(tokens), lstm.weight_ih_l0, lstm.bias_ih_l0 + lstm.bias_hh_l0)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    want = [p.grad.clone() for p in (emb.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)]
    for p in (emb.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0):
        p.grad = None
    got = ops.embed_proj(tokens, emb.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, padding_idx)
    assert _close(got, ref)
    got.backward(dy)
    for p, wnt in zip((emb.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0), want):
        assert _close(p.grad, wnt, 1e-4)


def test_linear_fn_with_row_gather_vs_torch():
    from videonavqa_amd import ops
    torch.manual_seed(5)
    hs = torch.randn(40, 16, device="cuda", requires_grad=True)
    lin = nn.Linear(16, 24).cuda()
    rows = torch.tensor([3, 17, 39, 0, 8], dtype=torch.int32, device="cuda")
    ref = F.relu(lin(hs[rows.long()]))
    dy = torch.randn_like(ref)
    ref.backward(dy)
    want = (hs.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    hs.grad = lin.weight.grad = lin.bias.grad = None
    got = ops.linear(hs, lin.weight, lin.bias, relu=True, rows=rows)
    assert _close(got, ref)
    got.backward(dy)
    for a, b in zip((hs.grad, lin.weight.grad, lin.bias.grad), want):
        assert _close(a, b, 1e-5)


@pytest.mark.parametrize("reduction", ["sum", "mean"])
@pytest.mark.parametrize("weighted", [False, True])
def test_ce_loss_vs_torch(reduction, weighted):
    from videonavqa_amd import ops
    torch.manual_seed(7)
    B, Kc = 11, 70
    logits = (torch.randn(B, Kc, device="cuda") * 3).requires_grad_(True)
    ys = torch.randint(0, Kc, (B,), device="cuda")
    perm = torch.randperm(B).to(torch.int32).cuda()
    w = (torch.rand(Kc, device="cuda") + 0.2) if weighted else None
    ref = nn.CrossEntropyLoss(weight=w, reduction=reduction)(logits, ys[perm.long()])
    ref.backward()
    want = logits.grad.clone()
    logits.grad = None
    got = ops.cross_entropy(logits, ys, row_perm=perm, weight=w, reduction=reduction)
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))
    (got * 2.0).backward()
    assert _close(logits.grad, 2.0 * want, 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, LOW_DTYPE])
def test_packed_temporal_attention_vs_dense_op(dtype):
    """packed kernel (frame-major image list, validity / masks formed on the fly) == the dense [B,T,A] op it replaces,
    ragged clips incl. frames past the longest video (un-masked on purpose)."""
    from videonavqa_amd import ops
    from videonavqa_amd.models.common import FrameLayout
    torch.manual_seed(9)
    B, T, A, ld = 4, 7, 16, 64
    lay = FrameLayout([5, 4, 2, 2], T, "cuda")
    f = torch.zeros(lay.n_img, ld, device="cuda")
    f[:, :A] = torch.randn(lay.n_img, A, device="cuda")
    f = f.to(dtype).requires_grad_(True)
    w = torch.randn(1, A, device="cuda", requires_grad=True)
    bias = torch.randn(1, device="cuda", requires_grad=True)
    # dense reference path (the op-by-op model code)
    dense = torch.zeros(B, T, A, device="cuda").index_put((lay.sample_of, lay.frame_of), f[:, :A].float())
    valid = torch.zeros(B, T, device="cuda").index_put((lay.sample_of, lay.frame_of), torch.ones(lay.n_img, device="cuda"))
    processed = torch.zeros(1, T, device="cuda")
    processed[:, :lay.n_frames] = 1
    masks = (processed - valid) * float(-(1 << 31))
    ctxt_ref, coef_ref = ops.temporal_attention(dense, valid, masks, w, bias)
    dctxt = torch.randn_like(ctxt_ref)
    ctxt_ref.backward(dctxt)
    want = (f.grad.clone(), w.grad.clone(), bias.grad.clone())
    f.grad = w.grad = bias.grad = None
    ctxt, coef = ops.temporal_attention_packed(f, lay.frame_off_i32, lay.n_frames, B, T, A, w, bias)
    assert _close(ctxt, ctxt_ref, 1e-5) and _close(coef, coef_ref, 1e-5)
    ctxt.backward(dctxt)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert _close(f.grad.float(), want[0].float(), tol) and float(f.grad[:, A:].abs().max()) == 0
    assert _close(w.grad, want[1], 1e-4) and _close(bias.grad, want[2], 1e-4)


def test_bn_running_update_matches_per_frame_ema():
    from videonavqa_amd import kernels as K
    torch.manual_seed(11)
    Fr, C, S = 6, 70, 196
    cts = [4, 4, 3, 3, 1, 1]
    off = torch.tensor([0] + list(torch.tensor(cts).cumsum(0)), dtype=torch.int32, device="cuda")
    mean = torch.randn(Fr, 128, device="cuda")
    var = torch.rand(Fr, 128, device="cuda") + 0.1
    rm, rv = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5
    em, ev = rm.clone(), rv.clone()
    for f in range(Fr):
        n = cts[f] * S
        em = 0.9 * em + 0.1 * mean[f, :C]
        ev = 0.9 * ev + 0.1 * var[f, :C] * n / (n - 1)
    K.bn_running_update(mean, var, off, Fr, S, rm, rv, 0.1)
    assert _close(rm, em, 1e-6) and _close(rv, ev, 1e-6)


@pytest.mark.parametrize("dt", [torch.float32, LOW_DTYPE])
@pytest.mark.parametrize("v_lens,T,tail,h,w", [([5, 5, 5], 5, 16, 4, 6), ([6, 4, 4, 1], 6, 32, 3, 5), ([3, 2], 7, 8, 14, 14)])
def test_frame_max_tail_vs_dense_stack(dt, v_lens, T, tail, h, w):
    """segmented max over frames from the packed image list (vnqa_frame_max_fwd / _bwd, csrc/pool_tail.hip) against the
    reference's dense zero-padded [T, B, ...] stack + max(dim=0) (film_global_pooling_pt_stem.py:230-235) and its autograd"""
    from videonavqa_amd import ops
    from videonavqa_amd.models.common import FrameLayout
    lay = FrameLayout(v_lens, T, "cuda")
    B, c_pad = len(v_lens), 64
    g = torch.Generator().manual_seed(sum(v_lens) + tail)
    maps = torch.zeros(lay.n_img, h + 2, w + 2, c_pad)
    vals = torch.relu(torch.randn(lay.n_img, h, w, tail, generator=g))
    vals[:, 0, 0, :] = 0.0                                    # a position where every frame is zero: no gradient, argmax -1
    vals[:, 1, 1, 0] = 0.7                                    # an exact tie between all frames: the FIRST frame wins
    maps[:, 1:-1, 1:-1, :tail] = vals
    maps = maps.cuda().to(dt).requires_grad_(True)
    pooled, argmax = ops.frame_max(maps, lay, tail, 2.0)
    # reference: dense stack, rows of absent (frame, sample) pairs stay zero
    mref = maps.detach().float().requires_grad_(True)
    dense = torch.zeros(lay.n_frames, B, h + 2, w + 2, c_pad, device="cuda")
    dense = dense.index_put((lay.frame_of, lay.sample_of), mref)
    ref = dense.max(dim=0)[0][:, 1:-1, 1:-1, :tail].permute(0, 3, 1, 2).reshape(B, -1)       # NCHW-flattened
    assert pooled.shape == ref.shape and torch.equal(pooled, ref)
    dp = torch.randn(pooled.shape, generator=g).cuda()
    pooled.backward(dp)
    ref.backward(dp)
    assert maps.grad.shape == maps.shape
    gref = 2.0 * mref.grad.to(dt).float()
    got = maps.grad.float().clone()
    # Where a sample's maximum is 0 every frame ties with the zero padding rows: torch routes the gradient to one of those
    # zeros (the relu backward of the real model then zeroes it), this kernel routes none — compare where the maximum is
    # positive, require zero elsewhere.  The forced 0.7 tie: the whole gradient goes to exactly one frame either way.
    pos = (ref.detach().view(B, tail, h, w) > 0).permute(0, 2, 3, 1)[lay.sample_of]         # [n_img, h, w, tail]
    gi, ri = got[:, 1:-1, 1:-1, :tail], gref[:, 1:-1, 1:-1, :tail]
    assert float((gi * (~pos)).abs().max()) == 0
    tie_got, tie_ref = gi[:, 1, 1, 0].clone(), ri[:, 1, 1, 0].clone()
    gi[:, 1, 1, 0] = 0
    ri[:, 1, 1, 0] = 0
    assert _close(gi * pos, ri * pos, 1e-6 if dt == torch.float32 else 1e-2)
    for b in range(B):
        sel = lay.sample_of == b
        assert abs(float(tie_got[sel].sum()) - float(tie_ref[sel].sum())) <= 1e-2 * abs(float(tie_ref[sel].sum())) + 1e-6
        assert int((tie_got[sel] != 0).sum()) <= 1
    assert float(maps.grad[:, 0].abs().max()) == 0 and float(maps.grad[..., tail:].abs().max()) == 0
    am = argmax.view(B, tail, h, w)
    assert int(am[:, :, 0, 0].max()) == -1                    # all-zero position
    assert torch.equal(am[:, 0, 1, 1], lay.frame_off_i32[0] + torch.arange(B, device="cuda", dtype=torch.int32))   # tie -> frame 0


@pytest.mark.parametrize("hidden", [128, 16, 64])
@pytest.mark.parametrize("v_lens,q_lens,blocks", [([4, 4, 4], [5, 3, 7], 2), ([6, 3, 1, 1], [2, 9, 4, 4], 3), ([2], [11], 1)])
def test_multi_hop_generator_hip_vs_torch(v_lens, q_lens, blocks, hidden):
    """the multi-hop FiLM generator on HIP kernels (csrc/hop_gen.hip + vnqa_sgemm, one autograd node) against the same
    generator op by op on stock torch (models/time_multi_hop_pt_stem.py:124-184): FiLM matrices and every gradient"""
    from videonavqa_amd.models import TimeMultiHopFiLMPretrainedStem
    from videonavqa_amd.models.common import FrameLayout
    B, T, C, H = len(v_lens), max(v_lens) + 1, 16, hidden
    torch.manual_seed(sum(q_lens) + blocks)
    model = TimeMultiHopFiLMPretrainedStem(B, 24, 7, num_input_channels=64, num_res_block_channels=C, num_res_blocks=blocks,
                                           num_tail_channels=8, hidden_size=H, vocab_size=30, spatial_size=12,
                                           precision="fp32").cuda()
    with torch.no_grad():
        for p in (model.encoder_norm.weight, model.decoder_norm.weight):
            p.uniform_(0.5, 1.5)
        for p in (model.encoder_norm.bias, model.decoder_norm.bias, model.fc_hidden_attn.bias):
            p.normal_(0, 0.3)
        model.fc_hidden_attn.weight.normal_(0, 0.5)
    lay = FrameLayout(v_lens, T, "cuda")
    ql = torch.tensor(q_lens)
    q = torch.randint(1, 30, (B, max(q_lens) + 2))
    q = (q * (torch.arange(q.shape[1]).unsqueeze(0) < ql.unsqueeze(1))).cuda()
    outs, grads = {}, {}
    for mode in ("hip", "torch"):
        model.zero_grad(set_to_none=True)
        model.init_hidden()
        films = model._generator_hip(q, ql, lay) if mode == "hip" else torch_partners.multi_hop_generator_torch(model, q, ql, lay)
        g = torch.Generator().manual_seed(3)
        loss = sum((f * torch.randn(f.shape, generator=g).cuda()).sum() for f in films)
        loss.backward()
        outs[mode] = [f.detach().clone() for f in films]
        grads[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    for a, b in zip(outs["hip"], outs["torch"]):
        assert a.shape == b.shape and _close(a, b, 2e-5)
    assert set(grads["hip"]) == set(grads["torch"])
    for n in grads["torch"]:
        ref = grads["torch"][n]
        if n == "fc_hidden_attn.bias":             # identically 0 (softmax shift invariance); torch returns rounding noise
            assert float(grads["hip"][n].abs().max()) < 1e-4 and float(ref.abs().max()) < 1e-3, n
            continue
        assert _close(grads["hip"][n], ref, 3e-4), (n, float((grads["hip"][n] - ref).abs().max()), float(ref.abs().max()))


def test_reserved_cu_stream_runs_kernels_and_persistent_grids_follow_per_call():
    """vnqa_stream_create_reserved: a CU-masked stream (32 CUs left to other streams) computes the same conv as the default
    stream; the persistent kernels take their CU reservation PER CALL (VNQA_CONV_RESERVE_CUS in the descriptor's flags — the
    library keeps no process-wide setting) and compute the same result with a smaller grid."""
    from videonavqa_amd import _lib as L, kernels as K
    x = torch.zeros(2, 18, 18, 64, device="cuda", dtype=L.half_dtype())
    x[:, 1:-1, 1:-1] = torch.randn(2, 16, 16, 64, device="cuda").to(x.dtype)
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wt = K.pack_conv_weight(w, x.dtype)
    ref = K.conv2d_igemm(x, wt, relu=True)
    assert not hasattr(L.lib(), "vnqa_set_persistent_reserve_") and L.conv_reserve_flags(32) == 4 << 8
    st = L.reserved_stream(32)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        got = K.conv2d_igemm(x, wt, relu=True)
        got_c64 = K.conv2d_c64(x, wt, relu=True, reserve_cus=32)          # persistent direct conv, grid of 224 workgroups
    st.synchronize()
    assert torch.equal(got, ref)
    plain_c64 = K.conv2d_c64(x, wt, relu=True)
    assert torch.equal(got_c64, plain_c64)
    with pytest.raises(L.VnqaError):
        L.reserved_stream(8)            # not a whole share of every XCD's shader engines


@pytest.mark.parametrize("shape", [(280, 512, 512), (37, 70, 200), (8, 70, 4480)])
def test_sgemm2_second_output_from_the_same_epilogue(shape):
    """vnqa_sgemm2: y = x w^T + addend and y2 = y * col[n] * mul[m, n] (with and without split-K slices)."""
    from videonavqa_amd import kernels as K
    m, n, k = shape
    g = torch.Generator().manual_seed(m + n + k)
    x, w = torch.randn(m, k, generator=g).cuda(), torch.randn(n, k, generator=g).cuda()
    add, col, mul = torch.randn(m, n, generator=g).cuda(), torch.randn(n, generator=g).cuda(), torch.randn(m, n, generator=g).cuda()
    ref = x.double() @ w.double().t() + add.double()
    for c_, m_ in ((col, None), (None, mul), (col, mul)):
        y, y2 = K.linear_nt2(x, w, addend=add, out2_col=c_, out2_mul=m_)
        r2 = ref * (1 if c_ is None else c_.double()) * (1 if m_ is None else m_.double())
        assert _close(y, ref.float(), 2e-5) and _close(y2, r2.float(), 2e-5)


def test_sgemm_batch_is_the_separate_calls():
    """vnqa_sgemm_batch: three independent products of different shapes / forms (x w^T + addend with a scaled second output,
    x w^T + bias, accumulating x w with a ReLU) in one launch vs float64 references; the nt form bit-identical to
    K.linear_nt2 (same kernel body, one pass over K)."""
    from videonavqa_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    r = lambda *sh: torch.randn(*sh, generator=g).cuda()
    m, d = 280, 512
    x1, w1, add1, col1 = r(m, d), r(d, d), r(m, d), r(d)
    x2, w2, b2 = r(m, d), r(200, d), r(200)
    x3, w3, c3 = r(70, 96), r(96, 40), r(70, 40)
    c1, o1, c2 = torch.empty(m, d).cuda(), torch.empty(m, d).cuda(), torch.empty(m, 200).cuda()
    c3_ref = torch.relu(c3.double() + x3.double() @ w3.double())
    K.sgemm_batch([
        dict(a=x1, b=w1, c=c1, addend=add1, out2=o1, out2_col=col1, a_rs=d, a_cs=1, b_rs=1, b_cs=d, ldc=d, m=m, n=d, k=d),
        dict(a=x2, b=w2, c=c2, bias=b2, a_rs=d, a_cs=1, b_rs=1, b_cs=d, ldc=200, m=m, n=200, k=d),
        dict(a=x3, b=w3, c=c3, a_rs=96, a_cs=1, b_rs=40, b_cs=1, ldc=40, m=70, n=40, k=96, relu=1, accumulate=1)])
    y, y2 = K.linear_nt2(x1, w1, addend=add1, out2_col=col1)
    assert torch.equal(c1, y) and torch.equal(o1, y2)
    assert _close(c1, (x1.double() @ w1.double().t() + add1.double()).float(), 2e-5)
    assert _close(c2, (x2.double() @ w2.double().t() + b2.double()).float(), 2e-5)
    assert _close(c3, c3_ref.float(), 2e-5)
    with pytest.raises(RuntimeError):
        K.sgemm_batch([dict(a=x1, b=w1, c=c1, a_rs=d, a_cs=1, b_rs=1, b_cs=d, ldc=d, m=m, n=d, k=d)] * 5)
