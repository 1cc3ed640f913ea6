"""TEST INFRASTRUCTURE: op-by-op torch forms of fused library nodes — the A/B partners the GPU tests compare the C-ABI nodes with.
They lived in the product until round 4 (a second code path nobody benchmarks); the product keeps the HIP path only."""
import torch

from videonavqa_amd import _lib as L
from videonavqa_amd import kernels as K
from videonavqa_amd import ops
from videonavqa_amd.models.common import repeated_question_lstm
from videonavqa_amd.ops import MacReadFn, MacReadState, _ret, _into, sink_of      # noqa: F401


class MacCoreTorchFn(torch.autograd.Function):
    """(VNQA_MAC_CORE_TORCH=1: the node issued op by op from Python on torch / rocBLAS GEMMs; the A/B partner of MacCoreFn.)
    One MAC reasoning step (ControlUnit, ReadUnit and WriteUnit.concat of models/mac.py:28-42,53-62,82-85) for all
    packed images as ONE autograd node: inside, plain torch GEMMs and the fused attention kernels run without graph
    recording, and the backward is written out by hand.  Motivation: the MAC training step was bound by the launch
    thread (autograd bookkeeping of ~75 small ops per step and direction), not by the GPU.

      cq      = control Wc^T + pq                       (pq = position_aware_i(question) Wp^T + b, hoisted by the caller)
      control'= pool(ctx, cq * w_ca, b_ca) [* mask]     (attention over the question words)
      mem     = memory Wm^T + bm ;  v = control' * w_ra ;  u = mem * (v W1)
      read    = pool(know, pre; u, v, b_ra)             (re-associated ReadUnit, see models/mac.py)
      concat  = read Wr^T + memory Wmm^T + bw
    Returns (control', concat); self-attention / memory gate / the memory dropout mask stay with the caller."""

    @staticmethod
    def forward(ctx_, control, memory, pq, ctxw, know, pre, mask_c, wc, w_ca, b_ca, wm, bm, w1, w_ra, b_ra, wr, wmm, bw,
                state, Lq, S):
        N, d = control.shape
        cq = torch.addmm(pq, control, wc.t())
        qv = (cq * w_ca).contiguous()
        p_c, cnew = K.mac_read_fwd(ctxw, None, qv, None, b_ca.detach().float().contiguous(), N, Lq, d)
        if mask_c is not None:
            cnew = cnew * mask_c
        mem = torch.addmm(bm, memory, wm.t())
        v = (cnew * w_ra).contiguous()
        t = v @ w1
        u = (mem * t).contiguous()
        p_r, read = K.mac_read_fwd(know, pre, u, v, b_ra.detach().float().contiguous(), N, S, d)
        concat = torch.addmm(bw, read, wr.t()).addmm_(memory, wmm.t())
        ctx_.save_for_backward(control, memory, ctxw, know, pre, mask_c, wc, w_ca, wm, w1, w_ra, wr, wmm,
                               cq, qv, p_c, cnew, mem, v, t, u, p_r, read)
        ctx_.state, ctx_.dims, ctx_.index = state, (N, d, Lq, S), state.n_calls
        state.n_calls += 1
        return cnew, concat

    @staticmethod
    def backward(ctx_, d_cnew, d_concat):
        (control, memory, ctxw, know, pre, mask_c, wc, w_ca, wm, w1, w_ra, wr, wmm,
         cq, qv, p_c, cnew, mem, v, t, u, p_r, read) = ctx_.saved_tensors
        N, d, Lq, S = ctx_.dims
        st = ctx_.state
        d_concat = d_concat.contiguous()
        # Parameter gradients are ACCUMULATED in the shared state (GEMM with beta = 1 / GEMV against a ones vector, in
        # place) and handed to autograd once, by the first step's node: 12 steps x 12 parameters would otherwise be
        # ~150 AccumulateGrad adds and ~60 reductions of their own.
        G = st.grads
        if not G:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=control.device)
            # wca / wra (gradients of the two attention weight vectors) are sums over images AND steps of an elementwise
            # product: accumulated per image with one addcmul_ per step and reduced over the images once, by the last node;
            # bca / bra (sums of the score gradients) come from the stacked score gradients the same node already holds
            G.update(wc=z(d, d), wca=z(N, d), wm=z(d, d), bm=z(d), w1=z(d, d), wra=z(N, d), wr=z(d, d),
                     wmm=z(d, d), bw=z(d), ones=torch.ones(N, dtype=torch.float32, device=control.device))
        ones = G["ones"]
        # WriteUnit.concat
        d_read = (d_concat @ wr).contiguous()
        d_memory = d_concat @ wmm
        G["wr"].addmm_(d_concat.t(), read)
        G["wmm"].addmm_(d_concat.t(), memory)
        G["bw"].addmv_(d_concat.t(), ones)
        # ReadUnit attention
        ds_r, du, dv = K.mac_read_bwd(know, pre, p_r, d_read, N, S, d)
        st.read.append((ds_r, p_r, u, v, d_read))
        d_mem, d_t = du * t, du * mem
        dv = dv.addmm_(d_t, w1.t())
        G["w1"].addmm_(v.t(), d_t)
        G["wra"].addcmul_(dv, cnew)
        d_c = dv * w_ra if d_cnew is None else torch.addcmul(d_cnew, dv, w_ra)
        d_memory = d_memory.addmm_(d_mem, wm)
        G["wm"].addmm_(d_mem.t(), memory)
        G["bm"].addmv_(d_mem.t(), ones)
        if mask_c is not None:
            d_c = d_c * mask_c
        d_c = d_c.contiguous()
        # ControlUnit attention
        ds_c, dqv, _ = K.mac_read_bwd(ctxw, None, p_c, d_c, N, Lq, d)
        st.ctrl.append((ds_c, p_c, qv, d_c))
        d_cq = dqv * w_ca
        G["wca"].addcmul_(dqv, cq)
        d_control = d_cq @ wc
        G["wc"].addmm_(d_cq.t(), control)
        d_ctxw = d_know = d_pre = None
        g = [None] * 11
        if ctx_.index == 0:      # runs last: every later step depends on this one's outputs
            f = [torch.stack(x) for x in zip(*st.read)]
            d_know, d_pre = K.mac_read_accum(f[0], f[1], f[2], f[3], f[4], N, S, d, know.shape[-1], know.dtype)
            c = [torch.stack(x) for x in zip(*st.ctrl)]
            d_ctxw, _ = K.mac_read_accum(c[0], c[1], c[2], None, c[3], N, Lq, d, ctxw.shape[-1], ctxw.dtype)
            g = [G["wc"], G["wca"].sum(0, keepdim=True), c[0].sum().view(1), G["wm"], G["bm"], G["w1"],
                 G["wra"].sum(0, keepdim=True), f[0].sum().view(1), G["wr"], G["wmm"], G["bw"]]
            st.read, st.ctrl, st.grads = [], [], {}
        return (d_control, d_memory, d_cq, d_ctxw, d_know, d_pre, None, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7],
                g[8], g[9], g[10], None, None, None)




def mac_core_torch(control, memory, pq_all, step, *rest):
    """Drop-in for ops.mac_core: the reasoning step on torch / rocBLAS GEMMs (monkeypatch ops.mac_core with it, ops.MAC_CHAIN = False)."""
    return MacCoreTorchFn.apply(control, memory, pq_all[step], *rest)


def multi_hop_generator_torch(self, q_input, q_lens, lay):
    """the same generator op by op on stock torch (CPU / VNQA_HOP_TORCH=1 cross-check)"""
    B, Fn, Hq = lay.B, lay.n_frames, self.hidden_size
    emb = self.embed(q_input)
    h0, c0 = self._question_state(B, Hq, q_lens, q_input.device)
    h_last, states, (hn, cn) = repeated_question_lstm(self.q_encoder, emb, q_lens, Fn, h0, c0,
                                                      want_states=True, wgrad_dtype=self._lstm_wgrad_dtype())
    self._store_question_state(hn, cn, q_lens)
    enc = self.encoder_norm(h_last)                                   # [B,F,H]   :148
    # per-image context, reset at every frame (:157-158); the hop chain runs over blocks
    hv = enc[lay.sample_of, lay.frame_of]                             # [n_img,H]
    st = states[lay.sample_of, lay.frame_of]                          # [n_img,Lmax,H] zero past q_len
    film_per_block = []
    for _ in range(self.num_res_blocks):
        prod = hv.unsqueeze(1) * st                                   # :170
        coefs = torch.softmax(self.fc_hidden_attn(prod), dim=1)       # unmasked over words :171-172
        hv = torch.bmm(coefs.permute(0, 2, 1), prod).squeeze(1)       # :175-176
        film_per_block.append(self.decoder_norm(self.fc_attn_out(hv)))  # :179,184
    return film_per_block



def gp_tail_torch(model, x, lay, h, w):
    """The pooling heads' tail (film_global_pooling_pt_stem.py:228-238) as the dense torch stack: relu(c1x1_tail) -> zero-padded
    [T, B, ...] stack -> max over frames -> out_linear; the partner of models.common.FiLMTrunkBase._gp_tail's HIP kernels."""
    import torch.nn.functional as F
    gs = getattr(model, "_trunk_grad_scale", 1.0)
    t = ops.conv(x, model.c1x1_tail.weight, model.c1x1_tail.bias, relu=True, grad_scale=gs)
    tail = model.c1x1_tail.out_channels
    n_img, hp, wp, tp = t.shape
    dense = torch.zeros(lay.n_frames, lay.B, hp, wp, tp, device=t.device, dtype=torch.float32)
    dense = dense.index_put((lay.frame_of, lay.sample_of), ops.scale_grad(t.float(), gs))
    pooled = dense.max(dim=0)[0].reshape(lay.B, -1)
    rows = model.out_linear.out_features
    w4 = model.out_linear.weight.view(rows, tail, h, w).permute(0, 2, 3, 1)
    w4 = F.pad(w4, (0, tp - tail, 1, 1, 1, 1, 0, 0))
    return pooled @ w4.reshape(rows, -1).t() + model.out_linear.bias
