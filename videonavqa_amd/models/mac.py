"""MACNetwork — drop-in for models/mac.py of the reference (`--model mac`, eval/q_and_v_eval.py:288-293).

Same constructor signature, parameter names (state_dict keys) and forward() contract.  What differs is
how the work is laid out for the MI355X:

  * the reference runs conv stack + MAC cell chain once PER FRAME (mac.py:226-243).  Frames are
    independent given the question encoding, so all valid (sample, frame) pairs of the minibatch form one
    packed image list and the `max_step` reasoning steps run ONCE over that list;
  * the three 3x3 convs run on the MFMA implicit-GEMM kernels (ops.conv) in padded NHWC;
  * ReadUnit (mac.py:53-62) applies a dim*2 -> dim Linear to [mem*know ; know] at every position and
    every step.  The score it feeds is linear in that projection, so it is re-associated:
        score[n,s] = know[n,s,:] . (mem[n] * (W1^T v[n])) + (know W2^T + b)[n,s,:] . v[n] + b_attn,
        v = control * w_attn, [W1 | W2] = concat.weight,
    leaving ONE position-wise GEMM per forward (know W2^T, step-invariant) instead of max_step of twice
    the size; the per-step work is three sweeps over the knowledge base in one fused HIP kernel
    (csrc/mac_read.hip: scores, softmax over positions, weighted read; backward with the outer-product
    gradients of all steps formed in a single pass);
  * both nn.LSTMs (bidirectional question encoder, 3*dim tail) run on the step-wise wide-LSTM HIP
    kernels (ops.lstm_wide) directly on PackedSequence batch sizes.
"""
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import kernels as K
from .. import ops
from .common import FrameLayout, NativeFeatures, compute_dtype


def _xavier_linear(n_in, n_out):
    """mac.py:7-13: Xavier-uniform weight, zero bias."""
    lin = nn.Linear(n_in, n_out)
    nn.init.xavier_uniform_(lin.weight)
    nn.init.zeros_(lin.bias)
    return lin


class _Unit(nn.Module):
    """Parameter holder; compute lives in MACNetwork._reason."""


def _control_unit(dim, max_step):        # mac.py:16-26
    u = _Unit()
    u.position_aware = nn.ModuleList([_xavier_linear(2 * dim, dim) for _ in range(max_step)])
    u.control_question = _xavier_linear(2 * dim, dim)
    u.attn = _xavier_linear(dim, 1)
    return u


def _read_unit(dim):                     # mac.py:46-51
    u = _Unit()
    u.mem = _xavier_linear(dim, dim)
    u.concat = _xavier_linear(2 * dim, dim)
    u.attn = _xavier_linear(dim, 1)
    return u


def _write_unit(dim, self_attention, memory_gate):   # mac.py:66-80
    u = _Unit()
    u.concat = _xavier_linear(2 * dim, dim)
    if self_attention:
        u.attn = _xavier_linear(dim, 1)
        u.mem = _xavier_linear(dim, dim)
    if memory_gate:
        u.control = _xavier_linear(dim, 1)
    return u


class MACNetwork(nn.Module):
    """Positional signature and defaults of mac.py:169-171; keyword-only extras: precision."""

    # train.Trainer keeps this many CUs free of the frozen stem's kernels (CU-masked stem stream, 16 CUs of every XCD): this
    # model's trunk is a ~1 000-launch dependent chain of small kernels that otherwise queue behind stem workgroups owning a
    # whole CU's LDS, so stem (6.6 ms) and chain (11.3 ms) ran back to back.  With the chain on the Trainer's high-priority
    # stream AND half the chip to itself they overlap: `tools/ab_mac_reserve.sh`, same box — reserve 0: 413 clips/s, 96: 492,
    # 128: 503-505, 160: 445, 192: 316 (the masked stem alone takes 8.5 ms at 128, the chain 11.3 + contention).  Without the
    # high-priority trunk stream a reservation makes things worse (64 CUs: 275), which is what round 3 measured first.
    # With the chain 1.9 ms shorter (wide-LSTM rewrite, question directions sharing launches): 64: 542, 96: 574, 128: 561-564.
    stem_reserve_cus = 96

    def __init__(self, n_vocab, dim, embed_hidden=300, max_step=12, self_attention=False, memory_gate=False,
                 classes=28, dropout=0.15, max_num_frames=35, *, precision='fp16h'):
        super(MACNetwork, self).__init__()
        self.compute_dtype = compute_dtype(precision)
        self.conv = nn.Sequential(nn.Conv2d(512, dim, 3, padding=1), nn.ELU(),      # mac.py:174-179
                                  nn.Conv2d(dim, dim, 3, padding=1), nn.ELU(),
                                  nn.Conv2d(dim, dim, 3, padding=1), nn.ELU())
        self.embed = nn.Embedding(n_vocab, embed_hidden, padding_idx=0)
        self.lstm = nn.LSTM(embed_hidden, dim, batch_first=True, bidirectional=True)
        self.lstm_proj = nn.Linear(dim * 2, dim)
        mac = _Unit()
        mac.control = _control_unit(dim, max_step)
        mac.read = _read_unit(dim)
        mac.write = _write_unit(dim, self_attention, memory_gate)
        mac.mem_0 = nn.Parameter(torch.zeros(1, dim))
        mac.control_0 = nn.Parameter(torch.zeros(1, dim))
        self.mac = mac
        self.lstm_tail = nn.LSTM(dim * 3, dim * 3)
        self.classifier = nn.Sequential(_xavier_linear(dim * 3, dim * 2), nn.ELU(), _xavier_linear(dim * 2, classes))
        self.max_step, self.dim, self.max_num_frames = max_step, dim, max_num_frames
        self.self_attention, self.memory_gate = self_attention, memory_gate
        mac.dropout = dropout       # where upstream keeps it (MACUnit.dropout, mac.py:123)
        self.grad_clamp = 1.0        # eval/q_and_v_eval.py:348-351 (read by train.Trainer)
        self.dropout_masks = None    # tests: (control_mask [n_img,dim], memory_mask [n_img,dim]) instead of bernoulli
        self.reset()

    def reset(self):
        """mac.py:199-207: embedding U(0,1) (padding row included), He-uniform on the FIRST TWO convs and the
        first classifier layer; the third conv and lstm_proj keep PyTorch's defaults."""
        with torch.no_grad():
            self.embed.weight.uniform_(0, 1)
            for k in (0, 2):
                nn.init.kaiming_uniform_(self.conv[k].weight)
                self.conv[k].bias.zero_()
            nn.init.kaiming_uniform_(self.classifier[0].weight)

    def extra_state_tensors(self):
        return {}

    def load_reference_tensors(self, tensors):
        own = self.state_dict()
        with torch.no_grad():
            for k, v in tensors.items():
                own[k].copy_(torch.as_tensor(v).to(own[k].device).view_as(own[k]))

    # ---- question side (mac.py:203-221) ------------------------------------------------------------
    def _encode_question(self, question, q_lens, B, dev):
        ql = q_lens.detach().cpu().long()
        lens_sorted, perm = torch.sort(ql, dim=0, descending=True, stable=True)
        perm_d = L.to_device_async(perm, dev)
        Lmax = int(lens_sorted[0])
        emb = F.embedding(question[:B], self.embed.weight, padding_idx=0)[perm_d][:, :Lmax]
        bsz = ops.packed_batch_sizes(lens_sorted)
        H = self.dim
        xgs = []
        for sfx in ("", "_reverse"):
            w_ih = getattr(self.lstm, "weight_ih_l0" + sfx)
            bias = getattr(self.lstm, "bias_ih_l0" + sfx) + getattr(self.lstm, "bias_hh_l0" + sfx)
            xgs.append(F.linear(emb, w_ih, bias).transpose(0, 1).contiguous())   # [Lmax,B,4H]
        # both directions' chains share their launches (ops.LstmWideBidirFn)
        outs = ops.lstm_wide_bidir(xgs[0], xgs[1], self.lstm.weight_hh_l0, self.lstm.weight_hh_l0_reverse, bsz)   # 2 x [Lmax,B,H]
        finals = [outs[0][L.to_device_async(lens_sorted - 1, dev), torch.arange(B, device=dev)], outs[1][0]]
        lstm_out = torch.cat(outs, 2).transpose(0, 1)                             # [B,Lmax,2H] sorted order
        inv = L.to_device_async(torch.sort(perm, dim=0)[1], dev)
        context = self.lstm_proj(lstm_out[inv])                                   # :217-220 (pad rows -> bias)
        hq = torch.cat(finals, 1)                                                 # :221: stays in SORTED order
        return context, hq

    # ---- image side ----------------------------------------------------------------------------------
    def _prepare_input(self, images, v_lens):
        if isinstance(images, NativeFeatures):
            assert images.data.dtype == self.compute_dtype
            return images.data, images.layout, images.h, images.w
        assert images.is_cuda, "the HIP path needs device tensors (no CPU fallback)"
        B, C, h, w, T = images.shape
        lay = FrameLayout(v_lens, T, images.device)
        return K.feat_to_nhwc(images, lay.img_of, lay.n_img, self.compute_dtype), lay, h, w

    def _knowledge(self, x):
        """conv -> ELU three times (mac.py:174-179,236); returns the dense interior [n_img*S, c_pad]."""
        for k in (0, 2, 4):
            x = ops.conv(x, self.conv[k].weight, self.conv[k].bias, relu=2)     # ELU in the conv epilogue; elu(0)=0 keeps the halo
        n_img, hp, wp, c_pad = x.shape
        return x[:, 1:-1, 1:-1, :].reshape(n_img * (hp - 2) * (wp - 2), c_pad), n_img, (hp - 2) * (wp - 2), c_pad

    def _masks(self, n_img, dev):
        if not self.training:
            return None
        if self.dropout_masks is not None:
            return self.dropout_masks
        keep = 1.0 - self.mac.dropout                                                 # mac.py:125-129
        return tuple(torch.empty(n_img, self.dim, device=dev).bernoulli_(keep) / keep for _ in range(2))

    def _question_terms(self, question, question_len, lay, dev):
        """Everything of the forward that depends on the question only: the encoder (mac.py:203-221), the per-image context
        rows and all steps' position-aware terms.  forward() runs it on a side stream next to the conv stack."""
        dim, m = self.dim, self.mac
        context, hq = self._encode_question(question, question_len, lay.B, dev)
        so = lay.sample_of
        Lq = context.shape[1]
        ctx = context[so].reshape(lay.n_img * Lq, dim).contiguous()               # [N*L,dim] fp32
        # all position_aware projections (one per reasoning step, mac.py:29) and their share of control_question
        # (:31-32: Linear([control ; position_aware])) in two batched products, hoisted out of the step loop
        pw = torch.stack([l.weight for l in m.control.position_aware])            # [steps,dim,2dim]
        pb = torch.stack([l.bias for l in m.control.position_aware])              # [steps,dim]
        pa_all = torch.matmul(hq, pw.transpose(1, 2)) + pb.unsqueeze(1)           # [steps,B,dim]
        wcq = m.control.control_question.weight
        pq_all = (torch.matmul(pa_all, wcq[:, dim:].t()) + m.control.control_question.bias)[:, so].contiguous()   # [steps,N,dim]
        return hq, ctx, pq_all, Lq

    def _reason(self, ctx, pq_all, Lq, kd, n_img, S, c_pad, lay):
        """MACUnit.forward for every image at once (mac.py:131-155 with the units at :28-42,53-62,82-105)."""
        dim, m = self.dim, self.mac
        dev = kd.device
        wcq = m.control.control_question.weight
        # step-invariant half of ReadUnit.concat on the MFMA GEMM: know W2^T + b (kept in the compute dtype)
        w2 = F.pad(m.read.concat.weight[:, dim:], (0, c_pad - dim, 0, c_pad - dim))
        pre = ops.linear_nt(kd, w2, F.pad(m.read.concat.bias, (0, c_pad - dim)))
        state = ops.MacCoreState(self.max_step)
        # contiguous halves of the [dim, 2 dim] weights ONCE (autograd-tracked copies): the node's `.contiguous()` of a column
        # slice otherwise copies 4 x 1 MB per reasoning step (48 launches per forward on the dependent chain)
        wc, w1 = wcq[:, :dim].contiguous(), m.read.concat.weight[:, :dim].contiguous()
        wr, wmm = m.write.concat.weight[:, :dim].contiguous(), m.write.concat.weight[:, dim:].contiguous()
        masks = self._masks(n_img, dev)
        control = m.control_0.expand(n_img, dim)
        memory = m.mem_0.expand(n_img, dim)
        if masks is not None:
            control, memory = control * masks[0], memory * masks[1]
        if not self.self_attention and not self.memory_gate and ops.MAC_CHAIN:
            # the reference's default configuration: all steps as ONE autograd node (ops.MacChainFn), the loop over steps in C++
            return ops.mac_chain(control.contiguous(), memory.contiguous(), pq_all, ctx, kd, pre,
                                 None if masks is None else masks[0], None if masks is None else masks[1], wc,
                                 m.control.attn.weight, m.control.attn.bias, m.read.mem.weight, m.read.mem.bias, w1,
                                 m.read.attn.weight, m.read.attn.bias, wr, wmm, m.write.concat.bias, Lq, S)
        controls, memories = [control], [memory]
        for i in range(self.max_step):
            # ControlUnit + ReadUnit + WriteUnit.concat as ONE autograd node (ops.MacCoreFn)
            prev = memories[-1]
            control, concat = ops.mac_core(control.contiguous(), prev.contiguous(), pq_all, i, ctx, kd, pre,
                                           None if masks is None else masks[0], wc, m.control.attn.weight,
                                           m.control.attn.bias, m.read.mem.weight, m.read.mem.bias, w1,
                                           m.read.attn.weight, m.read.attn.bias, wr, wmm, m.write.concat.bias,
                                           state, Lq, S)
            controls.append(control)
            nxt = concat
            if self.self_attention:
                cc = torch.stack(controls[:-1], 1)                                 # [N,i+1,dim]
                sa = torch.bmm(cc, (control * m.write.attn.weight).unsqueeze(2)) + m.write.attn.bias
                sa = F.softmax(sa, 1)
                nxt = m.write.mem((sa * torch.stack(memories, 1)).sum(1)) + concat
            if self.memory_gate:
                gate = torch.sigmoid(m.write.control(control))
                nxt = gate * prev + (1 - gate) * nxt
            memory = nxt
            if masks is not None:
                memory = memory * masks[1]
            memories.append(memory)
        return memory

    def forward(self, images, question, v_lens, question_len, actions=None, dropout=0.15):
        """images fp32 [B,512,h,w,T] (or NativeFeatures from the stem), v_lens sorted descending.
        Returns fp32 logits [B, classes] (mac.py:199-257)."""
        x, lay, h, w = self._prepare_input(images, v_lens)
        dev = x.device
        B = lay.B
        if torch.is_grad_enabled() and x.is_cuda:
            # the question side (embedding, bidirectional LSTM = 24 dependent launches, projections) has no input from the conv
            # stack: it runs on its own high-priority stream next to the three convs, and autograd runs its backward there too,
            # next to the convs' backward — both off the model's dependent chain
            main = torch.cuda.current_stream()
            side = getattr(self, "_q_stream", None)
            if side is None:
                side = self._q_stream = torch.cuda.Stream(priority=-1)
                # (the question encoder's parameters accumulate their gradients on this stream by design)
                if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                hq, ctx, pq_all, Lq = self._question_terms(question, question_len, lay, dev)
            kd, n_img, S, c_pad = self._knowledge(x)
            main.wait_stream(side)
            for t in (hq, ctx, pq_all):
                t.record_stream(main)
        else:
            hq, ctx, pq_all, Lq = self._question_terms(question, question_len, lay, dev)
            kd, n_img, S, c_pad = self._knowledge(x)
        memory = self._reason(ctx, pq_all, Lq, kd, n_img, S, c_pad, lay)
        out = torch.cat([memory, hq[lay.sample_of]], 1)                            # :240
        outs = torch.zeros(lay.n_frames, B, 3 * self.dim, device=dev).index_put((lay.frame_of, lay.sample_of), out)
        t = self.lstm_tail
        xg = F.linear(outs, t.weight_ih_l0, t.bias_ih_l0 + t.bias_hh_l0)
        hs = ops.lstm_wide(xg, t.weight_hh_l0, lay.cts, False)                     # packed by v_lens (:249-251)
        vl = L.to_device_async(torch.as_tensor([int(v) for v in (v_lens.tolist() if torch.is_tensor(v_lens) else v_lens)]), dev)
        last = hs[vl - 1, torch.arange(B, device=dev)]                             # :252-255
        return self.classifier(last)
