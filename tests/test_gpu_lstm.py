"""GPU numerics: the persistent HIP LSTM (forward + BPTT) against a plain PyTorch fp32 reference of
the same op: nn.LSTM stepped over each sample's tokens repeated n_rep times with carried state."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _torch_reference(lstm, emb, q_lens, n_rep, h0, c0):
    """per-sample python loop over nn.LSTM (CPU, fp32): the repeated sequence with carried state"""
    B = emb.shape[0]
    outs, hN, cN = [], [], []
    S = int(q_lens.max()) * n_rep
    for b in range(B):
        ql = int(q_lens[b])
        seq = emb[b, :ql].repeat(n_rep, 1).unsqueeze(1)           # [ql*n_rep, 1, E]
        o, (h, c) = lstm(seq, (h0[b].view(1, 1, -1), c0[b].view(1, 1, -1)))
        pad = torch.zeros(S - ql * n_rep, o.shape[-1])
        outs.append(torch.cat([o[:, 0], pad], 0))
        hN.append(h.view(-1))
        cN.append(c.view(-1))
    return torch.stack(outs), torch.stack(hN), torch.stack(cN)


@pytest.mark.parametrize("H,E,B,Lq,n_rep", [(16, 12, 3, 9, 6), (128, 128, 8, 25, 35), (64, 32, 5, 7, 3), (128, 128, 8, 1, 35)])
def test_lstm_seq_forward_backward(H, E, B, Lq, n_rep):
    from videonavqa_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(H + B)
    lstm = nn.LSTM(E, H)
    emb = torch.randn(B, Lq, E, requires_grad=True)
    q_lens = torch.randint(1, Lq + 1, (B,))
    q_lens[0] = Lq
    h0 = torch.randn(B, H) * 0.3
    c0 = torch.randn(B, H) * 0.3
    ref_out, ref_h, ref_c = _torch_reference(lstm, emb, q_lens, n_rep, h0, c0)
    S = ref_out.shape[1]
    wout = torch.randn(B, S, H)
    wh, wc = torch.randn(B, H), torch.randn(B, H)
    loss = (ref_out * wout).sum() + (ref_h * wh).sum() + (ref_c * wc).sum()
    loss.backward()
    ref_grads = [emb.grad.clone()] + [p.grad.clone() for p in lstm.parameters()]

    emb_d = emb.detach().cuda().requires_grad_(True)
    lstm_d = nn.LSTM(E, H).cuda()
    lstm_d.load_state_dict(lstm.state_dict())
    xg = F.linear(emb_d, lstm_d.weight_ih_l0, lstm_d.bias_ih_l0 + lstm_d.bias_hh_l0)
    out, hN, cN = ops.lstm_seq(xg, lstm_d.weight_hh_l0, h0.cuda(), c0.cuda(), q_lens.to(torch.int32).cuda(), n_rep, S)
    loss_d = (out * wout.cuda()).sum() + (hN * wh.cuda()).sum() + (cN * wc.cuda()).sum()
    loss_d.backward()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))

    assert rel(out.detach(), ref_out.detach()) < 2e-5
    assert rel(hN.detach(), ref_h.detach()) < 2e-5 and rel(cN.detach(), ref_c.detach()) < 2e-5
    got = [emb_d.grad] + [p.grad for p in lstm_d.parameters()]
    for g, r in zip(got, ref_grads):
        assert rel(g, r) < 2e-4, rel(g, r)


@pytest.mark.parametrize("B,T,A", [(3, 6, 16), (8, 35, 128), (5, 70, 200)])
def test_temporal_attention_vs_torch(B, T, A):
    """fused temporal softmax-attention kernel vs the plain PyTorch statement of film_attn_pt_stem.py:268-290"""
    from videonavqa_amd import ops
    torch.manual_seed(B * T)
    feat = torch.randn(B, T, A)
    valid = (torch.rand(B, T) > 0.3).float()
    valid[:, 0] = 1
    processed = torch.ones(B, T)
    processed[:, T - 2:] = 0                      # frames past the longest video: un-masked, zero features
    valid = valid * processed
    feat = feat * valid.unsqueeze(2)
    mask = (processed - valid) * float(-(1 << 31))
    w = torch.randn(1, A) * 0.3
    b = torch.randn(1) * 0.1
    wctx = torch.randn(B, A)

    def ref(feat, w, b):
        score = (feat @ w.t() + b).squeeze(2) * valid + mask
        coef = torch.softmax(score, dim=1)
        return torch.bmm(coef.unsqueeze(1), feat).squeeze(1), coef

    fr, wr, br = feat.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    cr, coef_r = ref(fr, wr, br)
    (cr * wctx).sum().backward()
    fd = feat.cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    cd, coef_d = ops.temporal_attention(fd, valid.cuda(), mask.cuda(), wd, bd)
    (cd * wctx.cuda()).sum().backward()

    def rel(a, b_):
        return float((a.cpu() - b_).abs().max() / (b_.abs().max() + 1e-12))

    assert rel(cd.detach(), cr.detach()) < 2e-5 and rel(coef_d, coef_r.detach()) < 2e-5
    assert rel(fd.grad, fr.grad) < 1e-4 and rel(wd.grad, wr.grad) < 1e-4
    # d bias = sum_t dscore_t is a cancellation (exactly 0 when every frame is valid): absolute scale of d w
    assert float((bd.grad.cpu() - br.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max()) + 1e-6
