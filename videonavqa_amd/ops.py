"""torch.autograd.Function wrappers over the HIP kernels (videonavqa_amd/kernels.py).

Activations are padded-NHWC tensors [N, H+2, W+2, Cpad] in the compute dtype (bf16 or fp32) with
a ZERO halo; every op here preserves that invariant (gradients included), because the conv
kernels get their zero padding — and wgrad its summation domain — from it.
"""

import torch

from . import _lib as L
from . import kernels as K


class GradSink(object):
    """Where a parameter's gradient lives when the training driver owns it (train.FlatParams: a slice of the flat gradient
    buffer, zeroed by the fused clip+Adam kernel).  The FIRST gradient producer of a parameter after the buffer was zeroed
    writes straight into the slice and its autograd node returns None: no temporary gradient tensor and no AccumulateGrad
    add kernel (for fc_embed_attn.weight that add alone moved 150 MB).  Any FURTHER producer before the next zeroing — a
    second backward pass without an optimizer step (gradient accumulation, gradient checks on a Trainer-owned model), or a
    weight shared by two ops — finds `written` set and takes the ordinary route (temporary + AccumulateGrad add), so the
    slice accumulates exactly as `p.grad` would.  `train.FlatParams.mark_zeroed()` clears the flag wherever the buffer is
    zeroed.  `on_ready` (optional) is the data-parallel reducer's hook for parameters whose all-reduce starts early."""

    def __init__(self, view):
        self.view, self.on_ready = view, None
        self.written = False      # the slice already holds a gradient of the current accumulation window
        self._handed = False      # _into() gave the view to a kernel whose _ret() has not run yet

    def done(self):
        if self.on_ready is not None:
            self.on_ready()


def sink_of(param):
    """The parameter's GradSink, or None (plain autograd accumulation).  None as well when `param.grad` is no longer the
    sink's slice (e.g. a stock optimizer's zero_grad(set_to_none=True)): writing there would lose the gradient."""
    s = getattr(param, "_vnqa_grad_sink", None)
    if s is None or not param.requires_grad or not torch.is_grad_enabled():
        return None
    g = param.grad
    if g is None or g.data_ptr() != s.view.data_ptr():
        return None
    return s


def _into(sink, shape=None):
    """Output buffer for a gradient kernel: the sink's view (optionally reshaped), or None = allocate (no sink, or the
    slice already holds a gradient: the value then goes through AccumulateGrad)."""
    if sink is None or sink.written:
        return None
    sink._handed = True
    return sink.view if shape is None else sink.view.view(shape)


def _ret(sink, value):
    """What the autograd node returns for a parameter: None when the kernel wrote into the sink, else the value."""
    if sink is None or not sink._handed:
        return value
    sink._handed = False
    sink.written = True
    sink.done()
    return None


class ConvFn(torch.autograd.Function):
    """y = act(conv2d(x, weight) + bias), 3x3 pad 1 or 1x1; weight/bias are the reference-layout
    fp32 parameters (OIHW).  relu: False / True, or 2 = ELU in the conv's epilogue (MACNetwork, models/mac.py:174-179;
    computed on the fp32 accumulator, rounded once).  Backward: dgrad = the same igemm on flipped weights, wgrad = MFMA
    split-K kernel.  Replaces nn.Conv2d at models/film_attn_pt_stem.py:211,219,224."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, mask_in_backward=True, grad_scale=1.0, split_weights=False):
        ctx.grad_scale = float(grad_scale)      # d y arrives multiplied by this (fp16 loss scale): dW, db are divided by it
        c_out, c_in, k, _ = weight.shape
        cdt = x.dtype
        c_in_pad = x.shape[-1]
        c_out_pad = L.round_up(c_out, 64)
        ctx.x_segs = 3 if (L.is_half(x.dtype) and x.shape[-1] == 3 * L.round_up(c_in, 64)) else 1     # [hi | lo | hi] split features (precision 'fp16h')
        ctx.elu = relu is not True and relu == 2
        if ctx.x_segs == 3:
            c_in_pad //= 3
            wt = K.pack_conv_weight(weight, torch.float32, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            y = K.conv2d_igemm(x, wt, bias=K.pad_vec(bias, c_out_pad), relu=relu, split_in=True)
        else:
            sw = bool(split_weights) and L.is_half(x.dtype)
            wt = K.pack_conv_weight(weight, torch.float32 if sw else x.dtype, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            y = K.conv2d_igemm(x, wt, bias=K.pad_vec(bias, c_out_pad), relu=relu, split_weights=sw,
                               tile=L.TILE_AUTO)      # (plain convs stay on the igemm tiles: the patch-stationary kernel is faster
                                                      # alone, 219 vs 257 us, and -1.5 % end to end beside MACNetwork's masked stem)
        ctx.relu = bool(relu) and not ctx.elu and mask_in_backward   # False: the consumer's backward applies the ReLU mask
        ctx.dims = (c_out, c_in, k, c_out_pad, c_in_pad)
        ctx.save_for_backward(x, weight, y if (ctx.relu or ctx.elu) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        return ConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        c_out, c_in, k, c_out_pad, c_in_pad = ctx.dims
        dy = dy.contiguous()
        if ctx.relu:
            dy = K.relu_bwd(dy, y)
        elif ctx.elu:                          # d/dv elu(v) from the result: 1 where y > 0, else y + 1
            dy = torch.ops.aten.elu_backward(dy, 1.0, 1.0, 1.0, True, y)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            inv = 1.0 / ctx.grad_scale
            dwt, dbias = K.conv2d_wgrad(x, dy, k * k, x_segs=ctx.x_segs)
            dw = K.unpack_conv_wgrad(dwt, c_out, c_in, alpha=inv)
            db = dbias[:c_out].clone() if inv == 1.0 else dbias[:c_out] * inv
        if ctx.needs_input_grad[0]:
            wt_d = K.pack_conv_weight(weight, dy.dtype, transpose_flip=True, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            dx = K.conv2d_igemm(dy, wt_d)
        return dx, dw, db, None, None, None, None


class LinearNTFn(torch.autograd.Function):
    """out = x @ w.T + bias on the MFMA GEMM (split-K).  x [M,K] compute dtype, w [N,K] fp32 in the
    kernel-native column order, N a multiple of 64.  Replaces nn.Linear fc_embed_attn
    (models/film_attn_pt_stem.py:244) and out_linear of the pooling models."""

    @staticmethod
    def forward(ctx, x, w, bias):
        wc = w.to(x.dtype).contiguous()
        out = K.gemm_nt(x.contiguous(), wc, bias=bias.float().contiguous())
        ctx.save_for_backward(x, wc)
        return out

    @staticmethod
    def backward(ctx, dout):
        return LinearNTFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, wc = ctx.saved_tensors
        dout = dout.to(x.dtype).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = K.gemm_nt(dout, wc.t().contiguous())
        if ctx.needs_input_grad[1]:
            dw = K.gemm_tn(dout, x.contiguous())
        if ctx.needs_input_grad[2]:
            db = dout.float().sum(0)
        return dx, dw, db


def conv(x, weight, bias, relu=False, mask_in_backward=True, grad_scale=1.0, split_weights=False):
    """split_weights (precision 'fp16h'): the forward as two products against [w_hi | w_lo]."""
    return ConvFn.apply(x, weight, bias, relu, mask_in_backward, grad_scale, split_weights)


class ScaleGradFn(torch.autograd.Function):
    """Identity whose backward multiplies the gradient by `scale` — the entry point of the fp16 loss scale where the
    gradient is still fp32 (placed right after the `.float()` of a 16-bit activation: the cast's backward then rounds
    scale * g, not g, to fp16)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = float(scale)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def scale_grad(x, scale):
    return x if scale == 1.0 else ScaleGradFn.apply(x, scale)


class FrameBNTrainFn(torch.autograd.Function):
    """Train-mode BatchNorm2d applied frame by frame to a ReLU output (film_attn_pt_stem.py:211):
    statistics per (frame, channel), two HIP launches forward, two backward; the backward also applies
    the mask of the producing ReLU (x > 0).  Returns (y, mean [F,C], biased var [F,C])."""

    @staticmethod
    def forward(ctx, x, gamma, beta, frame_of_i32, frame_off_i32, n_frames, eps, relu_input):
        mean, var = K.frame_bn_stats(x, frame_off_i32, n_frames)
        rstd = torch.rsqrt(var + eps)
        g = gamma.detach().float().contiguous()
        y = K.frame_bn_apply(x, frame_of_i32, mean, rstd, g, beta.detach().float().contiguous())
        ctx.save_for_backward(x, mean, rstd, g, frame_of_i32, frame_off_i32)
        ctx.n_frames, ctx.relu_input = n_frames, relu_input
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, mean, rstd, g, frame_of_i32, frame_off_i32 = ctx.saved_tensors
        dx, s1, s2 = K.frame_bn_bwd(dy.contiguous(), x, frame_of_i32, frame_off_i32, mean, rstd, g, ctx.n_frames,
                                    ctx.relu_input)
        return dx, s2.sum(0), s1.sum(0), None, None, None, None, None


class FilmReluResFn(torch.autograd.Function):
    """out = relu(gamma[n] * z + beta[n]) + res (film_attn_pt_stem.py:229-241); gamma/beta fp32 [N, Cpad]."""

    @staticmethod
    def forward(ctx, z, res, gamma, beta):
        gamma = gamma.float().contiguous()
        beta = beta.float().contiguous()
        out = K.film_relu_res_fwd(z, res, gamma, beta)
        ctx.save_for_backward(z, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, gamma, beta = ctx.saved_tensors
        dout = dout.contiguous()
        dz, dgamma, dbeta = K.film_relu_res_bwd(dout, z, gamma, beta)
        return dz, dout, dgamma, dbeta


def shift_bias_correction(conv_w, shift, c_pad):
    """sum_{c, taps} W[o, c, tap] * shift[c], fp32 [c_pad]: what a per-channel constant `shift` of a conv's INPUT adds to every output —
    the bias term of mean-shifted storage (x = x' + shift everywhere, the halo included: it holds -shift).  Exact fp32 weights."""
    c_out, c_in = conv_w.shape[0], conv_w.shape[1]
    v = torch.mv(conv_w.float().sum((2, 3)), shift[:c_in].float())          # (two launches; c_out == c_pad for the reference's widths)
    if c_out == c_pad:
        return v
    out = torch.zeros(c_pad, dtype=torch.float32, device=conv_w.device)
    out[:c_out] = v
    return out


def unshift_features(x, shift):
    """Mean-shifted padded-NHWC features -> plain ones (x' + shift on the interior, zero halo), for consumers that read plain tensors."""
    y = x.float() + shift[:x.shape[-1]].view(1, 1, 1, -1)
    y[:, 0] = 0
    y[:, -1] = 0
    y[:, :, 0] = 0
    y[:, :, -1] = 0
    return y.to(x.dtype)


def is_split(x, c_in):
    """x is a SPLIT tensor [hi | lo | hi] of a layer with c_in input channels (precision 'fp16h': the stem's dual-output features,
    laid out for a three-product consumer)."""
    return L.is_half(x.dtype) and x.shape[-1] == 3 * L.round_up(c_in, 64)


# conv_init on split features (precision 'fp16h', 777 GFLOP at the headline): True — the patch-stationary kernel (1.3-1.4 PFLOP/s on
# 14 x 14 maps) + the two-pass statistics kernel; False — the implicit GEMM with the BNSTATS epilogue (one launch less, 0.3-0.4 of peak)
HEAD_CONV_PS = True
HEAD_SPLIT_OUT = True


class FilmTrunkHeadFn(torch.autograd.Function):
    """First half of the TRAIN-mode conv trunk (film_attn_pt_stem.py:211) — the part that needs nothing from the question:

        r = relu(conv_init(x))      + per-frame BatchNorm statistics in the conv's epilogue  (VNQA_EPI_BNSTATS)
        h = bn_init(r)              (batch statistics per frame)

    Its own autograd node so that the question side (the FiLM generator's LSTM chain, forked onto a side stream by the models)
    overlaps it in BOTH directions: forward, the chain runs beside conv_init / BN; backward, FilmTrunkBlocksFn hands d gamma /
    d beta to the generator's BPTT first and this node's BN backward + conv_init wgrad run beside it.
    Returns (h, mean [F, Cpad], var [F, Cpad]) — the statistics for the caller's running-stat update."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn_w, bn_b, meta):
        lay, C = meta.layout, meta.channels
        cdt = x.dtype
        c_pad = L.round_up(C, 64)
        # precision 'fp16h': split features [hi | lo | hi] from the stem — conv_init as THREE products (a plain conv over 3 C channels
        # against [w_hi | w_hi | w_lo]): the unrounded activation against unrounded weights; the layer's weight rounding alone is
        # 0.14e-6 of the fp16 precision's 0.70e-6 squared logits error, its input rounding 0.07e-6 (profiles/r05_precision_budget.txt)
        # twin: the mean-shifted features written TWICE, [x' | x'] (stem: FEATURE_TWIN) — a plain conv over 2 C channels against split
        # weights [w_hi | w_lo]: x' w_hi + x' w_lo, the weight rounding of the trunk's first layer gone for one more product
        twin = bool(getattr(meta, "in_twin", False)) and L.is_half(cdt) and x.shape[-1] == 2 * L.round_up(conv_w.shape[1], 64)
        split = is_split(x, conv_w.shape[1]) and not twin
        ctx.x_segs = 3 if split else (2 if twin else 1)
        c_in_pad = x.shape[-1] // ctx.x_segs
        wt0 = K.pack_conv_weight(conv_w, torch.float32 if (split or twin) else x.dtype, c_out_pad=c_pad, c_in_pad=c_in_pad)
        if twin:
            wt0 = K.split_weight2(wt0)
        b0 = K.pad_vec(conv_b, c_pad)
        # MEAN-SHIFTED features (round 6; stem.FrozenStem.feature_shift): x holds feature - mu_c, its halo -mu_c.  The conv of the true
        # feature is W * x' + sum_taps(W) mu: the second term goes into the bias in fp32 with the EXACT weights, so the 16-bit weight
        # rounding multiplies a zero-mean input (its coherent part — 0.14e-6 of the squared logits error on a positive-mean input —
        # is gone) and ONE product does what three on [hi | lo | hi] features did.  Backward: dW gets mu (x) db for every tap.
        in_shift = getattr(meta, "in_shift", None) if not split else None
        ctx.in_shift = in_shift
        if in_shift is not None:
            b0 = b0 + shift_bias_correction(conv_w.detach(), in_shift, c_pad)
        fused = None
        ps = split and HEAD_CONV_PS and K.conv_ps_supported(x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[-1], c_pad)
        # (maps the patch-stationary tiles do not serve — the 10 x 13 maps of the reference's 160 x 208 frames — take the same split
        # output from the 256x256 implicit-GEMM tile: round 6)
        hybrid = bool(getattr(meta, "hybrid", False)) and L.is_half(cdt)
        if (hybrid or twin) and not split:      # precision 'fp16h' on plain / twin mean-shifted features: the patch-stationary tile where it serves
            ps = HEAD_CONV_PS and K.conv_ps_supported(x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[-1], c_pad)
        split_out = (split or hybrid) and HEAD_SPLIT_OUT and HEAD_CONV_PS
        if L.is_half(cdt) and not ps and not split_out:     # (the fp32 parity precision keeps the exact two-pass statistics kernel)
            fused = K.conv2d_igemm_bnstats(x, wt0, b0, True, lay.frame_of_i32, lay.frame_off_i32, lay.n_frames, min(lay.cts), split_in=split)
        g = K.pad_vec(bn_w, c_pad)
        if split_out:
            # ... and its OUTPUT kept unrounded into the BatchNorm (hi + lo, VNQA_EPI_SPLIT_OUT): 0.020e-6 of the budget; the backward
            # reads the hi tensor alone (ReLU mask and x-hat)
            r, r_lo = K.conv2d_igemm_split_out(x, wt0, b0, True, split_in=split, tile=L.TILE_PS_224x256 if ps else L.TILE_256x256)
            mean, var = K.frame_bn_stats_split(r, r_lo, lay.frame_off_i32, lay.n_frames)
            rstd = torch.rsqrt(var + meta.eps)
            h = K.frame_bn_apply_split(r, r_lo, lay.frame_of_i32, mean, rstd, g, K.pad_vec(bn_b, c_pad))
        else:
            if fused is None:
                r = K.conv2d_igemm(x, wt0, bias=b0, relu=True, split_in=split, tile=L.TILE_PS_224x256 if ps else L.TILE_AUTO)
                mean, var = K.frame_bn_stats(r, lay.frame_off_i32, lay.n_frames)
            else:
                r, mean, var = fused
            rstd = torch.rsqrt(var + meta.eps)
            h = K.frame_bn_apply(r, lay.frame_of_i32, mean, rstd, g, K.pad_vec(bn_b, c_pad))
        ctx.meta = meta
        with torch.enable_grad():       # gradient sinks (FlatParams): conv_init w/b, bn w/b
            ctx.sinks = [sink_of(t) for t in (conv_w, conv_b, bn_w, bn_b)]
        ctx.save_for_backward(x, r, mean, rstd, g, conv_w)
        ctx.mark_non_differentiable(mean, var)
        return h, mean, var

    @staticmethod
    def backward(ctx, dout, _dm, _dv):
        return FilmTrunkHeadFn._backward(ctx, dout, _dm, _dv)

    @staticmethod
    def _backward(ctx, dout, _dm, _dv):
        meta = ctx.meta
        lay, C = meta.layout, meta.channels
        x, r, mean, rstd, g, conv_w = ctx.saved_tensors
        cdt = x.dtype
        c_pad = L.round_up(C, 64)
        dout = dout.contiguous()
        inv = 1.0 / meta.grad_scale
        scaled = meta.grad_scale != 1.0
        sinks = [None] * 4 if scaled else ctx.sinks      # (scaled small vectors go through a temporary)
        dr, s1, s2 = K.frame_bn_bwd(dout, r, lay.frame_of_i32, lay.frame_off_i32, mean, rstd, g, lay.n_frames, True)
        s_cw = ctx.sinks[0]
        s_cb, s_bw, s_bb = sinks[1:4]
        exact = c_pad == C
        dbn_w = _ret(s_bw if exact else None, K.colsum(s2, out=_into(s_bw) if exact else None)[:C])
        dbn_b = _ret(s_bb if exact else None, K.colsum(s1, out=_into(s_bb) if exact else None)[:C])
        dwt0, dbias0 = K.conv2d_wgrad(x, dr, 9, dbias_out=_into(s_cb) if exact else None, x_segs=ctx.x_segs)
        dw_t = K.unpack_conv_wgrad(dwt0, C, conv_w.shape[1], out=_into(s_cw), alpha=inv)
        if ctx.in_shift is not None:
            # mean-shifted input: the kernel contracted dY with x' (halo -mu included); the true input is x' + mu everywhere, so
            # every tap gets mu[c] * sum_pixels dY[o] = mu (x) db — added BEFORE the sink is handed on (data-parallel early reduce)
            dw_t.view(C, conv_w.shape[1], -1).addcmul_(dbias0[:C].float().view(-1, 1, 1), ctx.in_shift[:conv_w.shape[1]].view(1, -1, 1), value=inv)
        dconv_w = _ret(s_cw, dw_t)
        dconv_b = _ret(s_cb if exact else None, dbias0[:C])
        if scaled:
            dbn_w, dbn_b, dconv_b = dbn_w * inv, dbn_b * inv, dconv_b * inv
        dx = None
        if ctx.needs_input_grad[0]:
            assert ctx.x_segs == 1, "split features come from the frozen stem: no gradient flows into them"
            dx = K.conv2d_igemm(dr, K.pack_conv_weight(conv_w, cdt, transpose_flip=True, c_out_pad=c_pad, c_in_pad=x.shape[-1]))
        return dx, dconv_w, dconv_b, dbn_w, dbn_b, None


class FilmTrunkBlocksFn(torch.autograd.Function):
    """Second half of the TRAIN-mode conv trunk (film_attn_pt_stem.py:224-241), one autograd node for all FiLM residual blocks:

        per block k:  res = relu(conv1x1_k(h))                                         (frozen weights)
                      z, h = conv3x3_k(res), relu(gamma_k * z + beta_k) + res          (ONE launch, VNQA_EPI_FILM_RES)

    and a hand-written backward: FiLM backward (writes d gamma / d beta straight into their column range of the FiLM matrix
    gradient), wgrad + dgrad of the 3x3 conv with the residual join and the 1x1 conv's ReLU mask in the dgrad's epilogue
    (VNQA_EPI_ADD_MASK), the 1x1 dgrad.  Compared with one autograd node per op this removes the AccumulateGrad / add kernels
    of the residual joins and ~30 graph nodes of host bookkeeping.

    forward(h, meta, *tensors): tensors = the distinct FiLM matrices [n_img, ld] fp32 (meta.n_film of them), then
    (w1, b1, w3, b3) per block; meta.film_map[k] = (index of block k's matrix, column of its gamma; beta follows at + C)."""

    @staticmethod
    def forward(ctx, h, meta, *tensors):
        C, blocks = meta.channels, meta.blocks
        films = tensors[:meta.n_film]
        cdt = h.dtype
        c_pad = L.round_up(C, 64)
        saved = []
        for k in range(blocks):
            w1, b1, w3, b3 = tensors[meta.n_film + 4 * k: meta.n_film + 4 * k + 4]
            if getattr(meta, "hybrid", False) and L.is_half(cdt):
                # precision 'fp16h': the frozen 1x1 conv as two products against [w_hi | w_lo] (its weight rounding: 0.04e-6 of the
                # squared logits error for 29 GFLOP), the 3x3 conv below stays ONE product with its fused FiLM epilogue (1e-10)
                wt1 = meta.c1_packs32[k] if meta.c1_packs32 else K.pack_conv_weight(w1, torch.float32, c_out_pad=c_pad, c_in_pad=c_pad)
                res = K.conv2d_igemm(h, wt1, bias=K.pad_vec(b1, c_pad), relu=True, split_weights=True)
            else:
                wt1 = meta.c1_packs[k][0] if meta.c1_packs else K.pack_conv_weight(w1, cdt, c_out_pad=c_pad, c_in_pad=c_pad)
                res = K.conv2d_igemm(h, wt1, bias=K.pad_vec(b1, c_pad), relu=True)
            fi, col = meta.film_map[k]
            film = films[fi]
            z, h = K.conv2d_igemm_film_res(res, K.pack_conv_weight(w3, cdt, c_out_pad=c_pad, c_in_pad=c_pad),
                                           K.pad_vec(b3, c_pad), film[:, col:col + C], film[:, col + C:col + 2 * C], C, res,
                                           tile=K.ps_fused_tile(res))
            saved += [res, z]
        ctx.meta = meta
        with torch.enable_grad():       # gradient sinks (FlatParams): (w3, b3) per block
            ctx.sinks = [sink_of(tensors[meta.n_film + 4 * k + i]) for k in range(blocks) for i in (2, 3)]
        ctx.save_for_backward(*saved, *tensors)
        ctx.split_tail = bool(getattr(meta, "split_tail", False))
        if ctx.split_tail:
            # the LAST block's FiLM affine is differentiated by its own node (FilmTailFn): this node hands it z and res, and h — the
            # fused epilogue's output — as a constant
            ctx.mark_non_differentiable(h)
            return h, saved[-1], saved[-2]
        return h

    @staticmethod
    def backward(ctx, dout, dz_tail=None, dres_tail=None):
        return FilmTrunkBlocksFn._backward(ctx, dout, dz_tail, dres_tail)

    @staticmethod
    def _backward(ctx, dout, dz_tail=None, dres_tail=None):
        meta = ctx.meta
        C, blocks, nf = meta.channels, meta.blocks, meta.n_film
        sv = ctx.saved_tensors
        acts = sv[:2 * blocks]
        tensors = sv[2 * blocks:]
        films = tensors[:nf]
        cdt = acts[0].dtype
        c_pad = L.round_up(C, 64)
        split = ctx.split_tail          # (blocks == 1: FilmTailFn already took the FiLM backward; it hands dz and the residual's gradient)
        dout = dres_tail.contiguous() if split else dout.contiguous()
        inv = 1.0 / meta.grad_scale
        scaled = meta.grad_scale != 1.0
        sinks = [None] * len(ctx.sinks) if scaled else ctx.sinks      # (scaled small vectors go through a temporary)
        # a FiLM matrix whose every column is some block's gamma or beta needs no zero fill (attention / pooling models:
        # ONE matrix [n_img, 2*C*blocks]); multi-hop's per-block matrices are only partly written
        covered = nf == 1 and films[0].shape[1] == 2 * C * blocks and \
            sorted(c for _, c in meta.film_map) == [2 * C * k for k in range(blocks)]
        dfilms = [(torch.empty_like(f) if covered else torch.zeros_like(f)) if (ctx.needs_input_grad[2 + i] and not split) else None
                  for i, f in enumerate(films)]
        grads_blocks = [None] * (4 * blocks)
        for k in reversed(range(blocks)):
            w1, b1, w3, b3 = tensors[nf + 4 * k: nf + 4 * k + 4]
            res, z = acts[2 * k], acts[2 * k + 1]
            fi, col = meta.film_map[k]
            film = films[fi]
            if split:
                dz = dz_tail.contiguous()
            else:
                dfilm = dfilms[fi] if dfilms[fi] is not None else torch.empty_like(film)
                dz = K.film_relu_res_bwd_ld(dout, z, film[:, col:col + C], film[:, col + C:col + 2 * C], C,
                                            dfilm[:, col:col + C], dfilm[:, col + C:col + 2 * C])
            sw, sb = ctx.sinks[2 * k], sinks[2 * k + 1]
            dwt, dbias = K.conv2d_wgrad(res, dz, 9, dbias_out=_into(sb))
            grads_blocks[4 * k + 2] = _ret(sw, K.unpack_conv_wgrad(dwt, C, C, out=_into(sw), alpha=inv))
            direct_b = sb is not None and dbias.data_ptr() == sb.view.data_ptr()      # (only without channel padding)
            grads_blocks[4 * k + 3] = _ret(sb if direct_b else None, dbias[:C] * inv if scaled else dbias[:C])
            # (dgrad(dz) + dout) * [res > 0]: the 3x3 conv's dgrad with the residual join and the 1x1 conv's ReLU mask in its
            # epilogue (VNQA_EPI_ADD_MASK; bit-identical to conv2d_igemm followed by relu_bwd(dres, res, dout))
            gsum = K.conv2d_igemm_add_mask(dz, K.pack_conv_weight(w3, cdt, transpose_flip=True, c_out_pad=c_pad, c_in_pad=c_pad),
                                           dout, res, tile=K.ps_fused_tile(dz))
            # (the 1x1 convs are frozen upstream — never in parameters() — so they get no weight gradient)
            wt1d = meta.c1_packs[k][1] if meta.c1_packs else \
                K.pack_conv_weight(w1, cdt, transpose_flip=True, c_out_pad=c_pad, c_in_pad=c_pad)
            dout = K.conv2d_igemm(gsum, wt1d)
        if scaled:
            dfilms = [None if t is None else t.mul_(inv) for t in dfilms]
        return (dout, None) + tuple(dfilms) + tuple(grads_blocks)


class FilmTailFn(torch.autograd.Function):
    """The FiLM affine + ReLU + residual of the trunk's LAST block as its own autograd node — backward only: the forward value `h` was
    already written by the conv's fused epilogue (VNQA_EPI_FILM_RES) inside FilmTrunkBlocksFn.  Its backward is the FIRST trunk node
    to run and returns d gamma / d beta at once, so the FiLM generator's BPTT (its LSTM chain, ~1.4 ms on the side stream) starts
    beside the block's weight / data gradients instead of after the whole FilmTrunkBlocksFn node has returned."""

    @staticmethod
    def forward(ctx, z, res, h, film, col, C, grad_scale):
        ctx.save_for_backward(z, film)
        ctx.geom = (int(col), int(C), float(grad_scale))
        return h.view_as(h)

    @staticmethod
    def backward(ctx, dout):
        z, film = ctx.saved_tensors
        col, C, grad_scale = ctx.geom
        dout = dout.contiguous()
        dfilm = torch.empty_like(film) if film.shape[1] == 2 * C else torch.zeros_like(film)
        dz = K.film_relu_res_bwd_ld(dout, z, film[:, col:col + C], film[:, col + C:col + 2 * C], C,
                                    dfilm[:, col:col + C], dfilm[:, col + C:col + 2 * C])
        if grad_scale != 1.0:
            dfilm.mul_(1.0 / grad_scale)
        return dz, dout, None, dfilm, None, None, None


def film_trunk_blocks(h, meta, films, block_tensors):
    """FilmTrunkBlocksFn, with the last block's FiLM backward as its own node (FilmTailFn) when the trunk has ONE block and one FiLM
    matrix — the headline configuration (the generator's BPTT then starts as soon as d gamma / d beta exist)."""
    if meta.blocks == 1 and len(films) == 1:
        meta.split_tail = True
        hc, z, res = FilmTrunkBlocksFn.apply(h, meta, *films, *block_tensors)
        return FilmTailFn.apply(z, res, hc, films[0], meta.film_map[0][1], meta.channels, meta.grad_scale)
    meta.split_tail = False
    return FilmTrunkBlocksFn.apply(h, meta, *films, *block_tensors)


class TrunkMeta(object):
    """Non-tensor arguments of FilmTrunkFn."""

    def __init__(self, layout, channels, blocks, n_film, film_map, eps, grad_scale=1.0):
        self.layout, self.channels, self.blocks, self.n_film, self.film_map, self.eps = \
            layout, channels, blocks, n_film, film_map, eps
        # incoming d(out) is `grad_scale` times the true gradient (fp16 loss scale): every fp32 parameter / FiLM gradient
        # leaving the node is divided by it, the activation gradients inside stay scaled
        self.grad_scale = float(grad_scale)
        # per block (forward pack, dgrad pack) of the FROZEN 1x1 conv weights, cached by the model across steps; None = pack here
        self.c1_packs = None
        # precision 'fp16h': the frozen 1x1 convs' FORWARD as two products against split weights (c1_packs32: their fp32 packs)
        self.hybrid = False
        self.c1_packs32 = None
        self.in_shift = None      # mean-shifted features: the per-channel constant conv_init's bias absorbs (FilmTrunkHeadFn)


def film_trunk(x, conv_w, conv_b, bn_w, bn_b, meta, *tensors, join=None):
    """The TRAIN-mode conv trunk as two autograd nodes (FilmTrunkHeadFn, FilmTrunkBlocksFn).  `join` (optional callable) is
    invoked between them: the point where a FiLM generator forked onto a side stream must have delivered `tensors[:n_film]`.
    Returns (h, mean, var)."""
    h, mean, var = FilmTrunkHeadFn.apply(x, conv_w, conv_b, bn_w, bn_b, meta)
    if join is not None:
        join()
    return film_trunk_blocks(h, meta, list(tensors[:meta.n_film]), list(tensors[meta.n_film:])), mean, var


def frame_bn_train(x, gamma, beta, frame_of_i32, frame_off_i32, n_frames, eps, relu_input=True):
    return FrameBNTrainFn.apply(x, gamma, beta, frame_of_i32, frame_off_i32, n_frames, eps, relu_input)


def film_relu_res(z, res, gamma, beta):
    return FilmReluResFn.apply(z, res, gamma, beta)


def linear_nt(x, w, bias):
    return LinearNTFn.apply(x, w, bias)


class LstmSeqFn(torch.autograd.Function):
    """hs, hN, cN = LSTM over each sample's tokens repeated n_rep times (state carried), one HIP
    launch for the whole chain; BPTT in one launch too.  Inputs: xg [B,Lq,4H] (input projection
    with both biases), w_hh [4H,H], h0/c0 [B,H]; q_lens_i32 device int32 [B]; S = max cells."""

    @staticmethod
    def forward(ctx, xg, w_hh, h0, c0, q_lens_i32, n_rep, S, wgrad_dtype=torch.float32):
        ctx.wgrad_dtype = wgrad_dtype
        ctx.set_materialize_grads(False)      # unused outputs (hN, cN) arrive as None instead of freshly filled zero tensors
        xg = xg.float().contiguous()
        w = w_hh.float().contiguous()
        h0 = h0.float().contiguous()
        c0 = c0.float().contiguous()
        hs, gates, hN, cN = K.lstm_seq_fwd(xg, w, q_lens_i32, h0, c0, n_rep, S)
        ctx.save_for_backward(w, h0, c0, hs, gates, q_lens_i32)
        ctx.n_rep, ctx.Lq = n_rep, xg.shape[1]
        with torch.enable_grad():
            ctx.sink_w = sink_of(w_hh)
        return hs, hN, cN

    @staticmethod
    def backward(ctx, dhs, dhN, dcN):
        w, h0, c0, hs, gates, q_lens_i32 = ctx.saved_tensors
        B, S, H = hs.shape
        n_rep, Lq = ctx.n_rep, ctx.Lq
        dhs = torch.zeros_like(hs) if dhs is None else dhs.float().contiguous()
        dhN = None if dhN is None else dhN.float().contiguous()
        dcN = None if dcN is None else dcN.float().contiguous()
        dgates, dh0, dc0 = K.lstm_seq_bwd(w, q_lens_i32, c0, gates, dhs, dhN, dcN, n_rep)
        # dW_hh = sum_{b,t} dgates[b,t]^T h_{t-1}[b]   (rows past a sample's last cell are zero): both operands produced in
        # the GEMM's element type by one kernel.  Exact-f32 MFMA in the fp32 (parity) mode; bf16 operands / fp32 accumulation
        # in the bf16 mode (the f32 matrix path runs at 1/16 of the bf16 rate and this GEMM sits on the trunk's dependent chain)
        a, hprev = K.lstm_wgrad_operands(dgates, hs, h0, ctx.wgrad_dtype)
        dw = _ret(ctx.sink_w, K.gemm_tn(a, hprev, out=_into(ctx.sink_w)))
        # dxg[b][pos] = sum over repeats of dgates at cells t with t % q_len == pos
        dxg = K.lstm_fold_dxg(dgates, q_lens_i32, Lq, n_rep)
        return dxg, dw, dh0, dc0, None, None, None, None


def lstm_seq(xg, w_hh, h0, c0, q_lens_i32, n_rep, S, wgrad_dtype=torch.float32):
    return LstmSeqFn.apply(xg, w_hh, h0, c0, q_lens_i32, n_rep, S, wgrad_dtype)


class Conv3dFn(torch.autograd.Function):
    """y = [relu](conv3d(x, weight, k=3, pad=1) + bias) on padded NDHWC [N, D+2, H+2, W+2, C]
    (nn.Conv3d of models/v_only_cnn3d.py:13-26): forward and dgrad on the igemm with a 27-tap table, wgrad on
    the MFMA wgrad kernel with 3-D tap shifts."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, need_dx):
        c_out, c_in = weight.shape[0], weight.shape[1]
        cdt = x.dtype
        c_in_pad, c_out_pad = x.shape[-1], L.round_up(c_out, 64)
        wt = K.pack_conv_weight(weight, cdt, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        y = K.conv3d_igemm(x, wt, bias=K.pad_vec(bias, c_out_pad), relu=relu)
        ctx.relu, ctx.need_dx = relu, need_dx
        ctx.dims = (c_out, c_in, c_out_pad, c_in_pad)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        c_out, c_in, c_out_pad, c_in_pad = ctx.dims
        dy = dy.contiguous()
        if ctx.relu:
            dy = K.relu_bwd(dy, y)
        dwt, dbias = K.conv3d_wgrad(x, dy)
        dw = K.unpack_conv_wgrad(dwt, c_out, c_in)
        dx = None
        if ctx.need_dx:
            wt_d = K.pack_conv_weight(weight, dy.dtype, transpose_flip=True, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            dx = K.conv3d_igemm(dy, wt_d)
        return dx, dw, dbias[:c_out].clone(), None, None


def conv3d(x, weight, bias, relu=False, need_dx=True):
    return Conv3dFn.apply(x, weight, bias, relu, need_dx)


def ncdhw_to_ndhwc_padded(x, dtype, c_pad=None):
    """dense [N,C,D,H,W] -> padded NDHWC [N,D+2,H+2,W+2,c_pad] (differentiable torch plumbing)."""
    C = x.shape[1]
    c_pad = c_pad or L.round_up(C, 64)
    y = x.permute(0, 2, 3, 4, 1)
    y = torch.nn.functional.pad(y, (0, c_pad - C, 1, 1, 1, 1, 1, 1))
    return y.to(dtype).contiguous()


def ndhwc_padded_to_ncdhw(x, C):
    """padded NDHWC -> dense fp32 [N,C,D,H,W] (differentiable)."""
    return x[:, 1:-1, 1:-1, 1:-1, :C].permute(0, 4, 1, 2, 3).float()


class _HaloPool(object):
    """Zero-halo padded NDHWC buffers handed out to the 3-D conv pipeline and returned after use: the kernels write interiors
    only, so a returned buffer is a ready zero-halo buffer again (no per-step memset of ~1 GB activations)."""

    def __init__(self):
        self.free = {}

    def get(self, shape, dtype, device):
        key = (tuple(shape), dtype, str(device), torch.cuda.current_stream().cuda_stream)
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        return torch.zeros(shape, dtype=dtype, device=device)

    def put(self, t):
        key = (tuple(t.shape), t.dtype, str(t.device), torch.cuda.current_stream().cuda_stream)
        lst = self.free.setdefault(key, [])
        if len(lst) < 2:
            lst.append(t)


_HALO_POOL = _HaloPool()


class _PoolLease(object):
    """Owner of pooled buffers that an autograd node keeps as saved tensors: they go back to the pool exactly ONCE — when the
    node's backward has consumed them (release()), or, for a grad-enabled forward that is never followed by a backward, when the
    graph is freed (the node's ctx drops this object).  A second backward over a retained graph would read buffers a later
    forward may already have rewritten: it is refused with a clear error instead."""

    def __init__(self, pool, *buffers):
        self.pool, self.buffers, self.released = pool, list(buffers), False

    def release(self):
        if not self.released:
            self.released = True
            for b in self.buffers:
                self.pool.put(b)
            self.buffers = []

    def __del__(self):
        try:
            self.release()
        except Exception:      # interpreter shutdown
            pass


class Cnn3dFeaturesFn(torch.autograd.Function):
    """bn_input -> 3 x [relu(conv3d) -> MaxPool3d -> BatchNorm3d] -> flatten (v_only_cnn3d.py:60-74) as ONE autograd node on
    csrc/cnn3d.hip + the 27-tap igemm / small-channel wgrad kernels, 16-bit storage:
      conv1 reads the fp32 clip directly (bn_input folded into its patch load and weights, pool(1,2,2) + bn1 statistics in its
      epilogue); every BatchNorm writes straight into the next conv's padded NDHWC input; the pools keep an arg-max byte per
      output (ReLU mask included) and the backward pool writes the conv's whole padded dY; conv1's backward is one split-K
      GEMM that also yields bn_input's parameter gradients.
    `bns` = the four BatchNorm3d modules (running statistics / momentum / eps live there), `training` selects batch statistics."""

    @staticmethod
    def forward(ctx, x, g0, b0, w1, c1b, g1, b1, w2, c2b, g2, b2, w3, c3b, g3, b3, bns, training, cdt, grad_scale):
        N, _, D, H, W = x.shape
        dev = x.device
        x = x.detach().float().contiguous()
        f = lambda t: t.detach().float().contiguous()
        g0, b0, w1, c1b, g1, b1, g2, b2, g3, b3 = [f(t) for t in (g0, b0, w1, c1b, g1, b1, g2, b2, g3, b3)]

        def stats(bn, partial, count):
            if training:
                return K.bn_finalize(partial, count, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                     bn.running_mean, bn.running_var)
            return bn.running_mean.detach().float().contiguous(), torch.rsqrt(bn.running_var.detach().float() + bn.eps)

        mean0, rstd0 = stats(bns[0], K.c3d_stats_ncdhw(x) if training else None, N * D * H * W)
        p1, idx1, part1 = K.c3d_conv1_fwd(x, w1, c1b, mean0, rstd0, g0, b0, cdt)
        H1, W1 = H // 2, W // 2
        mean1, rstd1 = stats(bns[1], part1, N * D * H1 * W1)
        a1 = _HALO_POOL.get((N, D + 2, H1 + 2, W1 + 2, 64), cdt, dev)
        K.bn_rows_apply(p1.view(-1, 64), a1, K.view_padded_ndhwc(D, H1, W1, 64), mean1, rstd1, g1, b1)
        c2 = w2.shape[0]
        wt2 = K.pack_conv_weight(w2, cdt, c_out_pad=c2, c_in_pad=64)
        y2 = _HALO_POOL.get((N, D + 2, H1 + 2, W1 + 2, c2), cdt, dev)
        K.conv3d_igemm(a1, wt2, bias=f(c2b), relu=True, out=y2)
        p2, idx2, part2 = K.pool444_fwd(y2)
        _HALO_POOL.put(y2)                       # the ReLU mask travels in idx2: y2 itself is not needed again
        D2, H2, W2 = D // 4, H1 // 4, W1 // 4
        mean2, rstd2 = stats(bns[2], part2, N * D2 * H2 * W2)
        a2 = _HALO_POOL.get((N, D2 + 2, H2 + 2, W2 + 2, c2), cdt, dev)
        K.bn_rows_apply(p2.view(-1, c2), a2, K.view_padded_ndhwc(D2, H2, W2, c2), mean2, rstd2, g2, b2)
        c3 = w3.shape[0]
        wt3 = K.pack_conv_weight(w3, cdt, c_out_pad=c3, c_in_pad=c2)
        y3 = _HALO_POOL.get((N, D2 + 2, H2 + 2, W2 + 2, c3), cdt, dev)
        K.conv3d_igemm(a2, wt3, bias=f(c3b), relu=True, out=y3)
        p3, idx3, part3 = K.pool444_fwd(y3)
        _HALO_POOL.put(y3)
        D3, H3, W3 = D2 // 4, H2 // 4, W2 // 4
        mean3, rstd3 = stats(bns[3], part3, N * D3 * H3 * W3)
        feat = torch.empty((N, c3 * D3 * H3 * W3), dtype=torch.float32, device=dev)
        K.bn_rows_apply(p3.view(-1, c3), feat, K.view_nc_flat(D3, H3, W3, c3), mean3, rstd3, g3, b3)
        ctx.save_for_backward(x, g0, b0, w1, g1, g2, g3, w2, w3, mean0, rstd0, mean1, rstd1, mean2, rstd2, mean3, rstd3,
                              p1, idx1, a1, p2, idx2, a2, p3, idx3)
        ctx.meta = (cdt, float(grad_scale), training)
        ctx.lease = _PoolLease(_HALO_POOL, a1, a2)
        if not any(ctx.needs_input_grad):         # no backward will come for them: the padded inputs go back to the pool now
            ctx.lease.release()
        return feat.view(N, c3, D3, H3, W3)

    @staticmethod
    def backward(ctx, dfeat):
        (x, g0, b0, w1, g1, g2, g3, w2, w3, mean0, rstd0, mean1, rstd1, mean2, rstd2, mean3, rstd3,
         p1, idx1, a1, p2, idx2, a2, p3, idx3) = ctx.saved_tensors
        cdt, gs, training = ctx.meta
        if not training:
            raise NotImplementedError("VideoOnlyCNN3D: backward through eval-mode BatchNorm is not part of the reference path")
        if ctx.lease.released:
            raise RuntimeError("VideoOnlyCNN3D features: this graph's pooled activation buffers were already returned (a second "
                               "backward over a retained graph is not supported on the fused 3-D path; VNQA_CNN3D_GENERIC=1 "
                               "keeps ordinary autograd semantics)")
        N, D1, H1, W1, _ = p1.shape
        _, D2, H2, W2, c2 = p2.shape
        _, D3, H3, W3, c3 = p3.shape
        dev = x.device
        dfeat = dfeat.reshape(N, -1)
        dfeat = (dfeat.float() * gs).contiguous() if gs != 1.0 else dfeat.float().contiguous()
        # stage 3
        dp3, dg3, db3 = K.bn_rows_bwd(dfeat, K.view_nc_flat(D3, H3, W3, c3), p3.view(-1, c3), cdt, mean3, rstd3, g3, gs)
        dy3 = _HALO_POOL.get((N, D2 + 2, H2 + 2, W2 + 2, c3), cdt, dev)
        K.pool444_bwd(dp3.view(p3.shape), idx3, dy3)
        dwt3, dbias3 = K.conv3d_wgrad(a2, dy3)
        dw3 = K.unpack_conv_wgrad(dwt3, c3, c2, alpha=1.0 / gs)
        da2 = _HALO_POOL.get((N, D2 + 2, H2 + 2, W2 + 2, c2), cdt, dev)
        K.conv3d_igemm(dy3, K.pack_conv_weight(w3, cdt, transpose_flip=True, c_out_pad=c3, c_in_pad=c2), out=da2)
        _HALO_POOL.put(dy3)
        # stage 2
        dp2, dg2, db2 = K.bn_rows_bwd(da2, K.view_padded_ndhwc(D2, H2, W2, c2), p2.view(-1, c2), cdt, mean2, rstd2, g2, gs)
        _HALO_POOL.put(da2)
        dy2 = _HALO_POOL.get((N, D1 + 2, H1 + 2, W1 + 2, c2), cdt, dev)
        K.pool444_bwd(dp2.view(p2.shape), idx2, dy2)
        dwt2, dbias2 = K.conv3d_wgrad(a1, dy2)
        dw2 = K.unpack_conv_wgrad(dwt2, c2, 64, alpha=1.0 / gs)
        da1 = _HALO_POOL.get((N, D1 + 2, H1 + 2, W1 + 2, 64), cdt, dev)
        K.conv3d_igemm(dy2, K.pack_conv_weight(w2, cdt, transpose_flip=True, c_out_pad=c2, c_in_pad=64), out=da1)
        _HALO_POOL.put(dy2)
        # stage 1
        dp1, dg1, db1 = K.bn_rows_bwd(da1, K.view_padded_ndhwc(D1, H1, W1, 64), p1.view(-1, 64), cdt, mean1, rstd1, g1, gs)
        _HALO_POOL.put(da1)
        dw1, dc1b, dg0, db0 = K.c3d_conv1_bwd(x, w1, mean0, rstd0, g0, b0, dp1, idx1, gs)
        ctx.lease.release()
        inv = 1.0 / gs
        return (None, dg0, db0, dw1, dc1b, dg1, db1, dw2, dbias2[:c2] * inv if gs != 1.0 else dbias2[:c2].clone(), dg2, db2,
                dw3, dbias3[:c3] * inv if gs != 1.0 else dbias3[:c3].clone(), dg3, db3, None, None, None, None)


def cnn3d_features(x, bn_input, conv1, bn1, conv2, bn2, conv3, bn3, training, cdt, grad_scale=1.0):
    return Cnn3dFeaturesFn.apply(x, bn_input.weight, bn_input.bias, conv1.weight, conv1.bias, bn1.weight, bn1.bias, conv2.weight,
                                 conv2.bias, bn2.weight, bn2.bias, conv3.weight, conv3.bias, bn3.weight, bn3.bias,
                                 (bn_input, bn1, bn2, bn3), training, cdt, grad_scale)


class BatchNormRowsFn(torch.autograd.Function):
    """nn.BatchNorm1d over dense fp32 rows [R, C] (bn6 / bn7 of v_only_cnn3d.py:76-79) on the channel-last BatchNorm kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training):
        x = x.contiguous()
        R, C = x.shape
        if training:
            mean, rstd = K.bn_finalize(K.c3d_stats_rows(x), R, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                                       bn.running_mean, bn.running_var)
        else:
            mean, rstd = bn.running_mean.detach().float().contiguous(), torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        y = torch.empty_like(x)
        g = gamma.detach().float().contiguous()
        K.bn_rows_apply(x, y, K.view_dense(1, 1, 1, C), mean, rstd, g, beta.detach().float().contiguous())
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the reference path")
        dx, dg, db = K.bn_rows_bwd(dy.contiguous(), K.view_dense(1, 1, 1, x.shape[1]), x, torch.float32, mean, rstd, g)
        return dx, dg, db, None, None


def batch_norm_rows(x, bn, training):
    return BatchNormRowsFn.apply(x, bn.weight, bn.bias, bn, training)


class TemporalAttnFn(torch.autograd.Function):
    """ctxt, coef = fused temporal attention (film_attn_pt_stem.py:268-290): fc_attn_1 scores on the valid
    (sample, frame) entries, -(1<<31) masks, softmax over frames, weighted sum of the frame features."""

    @staticmethod
    def forward(ctx, feat, valid, mask, w, bias):
        feat = feat.float().contiguous()
        w1 = w.float().reshape(-1).contiguous()
        b1 = bias.float().reshape(-1).contiguous()
        valid = valid.float().contiguous()
        coef, ctxt = K.temporal_attn_fwd(feat, valid, mask.float().contiguous(), w1, b1)
        ctx.save_for_backward(feat, valid, w1, coef)
        ctx.w_shape = w.shape
        ctx.mark_non_differentiable(coef)
        return ctxt, coef

    @staticmethod
    def backward(ctx, dctxt, _dcoef):
        feat, valid, w1, coef = ctx.saved_tensors
        dfeat, dw_part, db_part = K.temporal_attn_bwd(feat, valid, w1, coef, dctxt.float().contiguous())
        return dfeat, None, None, dw_part.sum(0).view(ctx.w_shape), db_part.sum().view(1)


def temporal_attention(feat, valid, mask, w, bias):
    """feat [B,T,A], valid/mask [B,T]; w [1,A], bias [1] (fc_attn_1).  Returns (ctxt [B,A], coef [B,T])."""
    return TemporalAttnFn.apply(feat, valid, mask, w, bias)


class LstmWideFn(torch.autograd.Function):
    """hs = packed LSTM over time-major xg [T,B,4H] (input projection incl. both biases) from a zero state,
    for hidden sizes too wide for the persistent kernel (MACNetwork's nn.LSTMs, models/mac.py:185-186,193).
    One HIP launch per step; BPTT likewise, then dW_hh as ONE GEMM over all steps."""

    @staticmethod
    def forward(ctx, xg, w_hh, batch_sizes, reverse):
        xg = xg.float().contiguous()
        w = w_hh.float().contiguous()
        hs, cs, gates = K.lstm_wide_fwd(xg, w, batch_sizes, reverse)
        ctx.save_for_backward(w, hs, cs, gates)
        ctx.batch_sizes, ctx.reverse = tuple(batch_sizes), reverse
        return hs

    @staticmethod
    def backward(ctx, dhs):
        w, hs, cs, gates = ctx.saved_tensors
        T, B, H = hs.shape
        dgates = K.lstm_wide_bwd(w.t().contiguous(), ctx.batch_sizes, gates, cs, dhs.float().contiguous(), ctx.reverse)
        # h of each step's predecessor in the chain (zero where the sample starts there: hs rows of inactive
        # (step, sample) pairs are zero)
        zero = torch.zeros(1, B, H, device=hs.device)
        hpred = torch.cat([hs[1:], zero], 0) if ctx.reverse else torch.cat([zero, hs[:-1]], 0)
        dw = K.gemm_tn(dgates.view(T * B, 4 * H), hpred.reshape(T * B, H).contiguous())
        return dgates, dw, None, None


def lstm_wide(xg, w_hh, batch_sizes, reverse=False):
    return LstmWideFn.apply(xg, w_hh, batch_sizes, reverse)


class LstmWideBidirFn(torch.autograd.Function):
    """(hs_f, hs_r) = both directions of a bidirectional packed LSTM (MACNetwork's question encoder, models/mac.py:185,
    210-213) with the two independent chains sharing their launches: chain position i of either direction runs in one
    launch, forward and backward — half the launches of two LstmWideFn nodes on the model's dependent chain, same bits."""

    @staticmethod
    def forward(ctx, xg_f, xg_r, w_hh_f, w_hh_r, batch_sizes):
        xg_f, xg_r = xg_f.float().contiguous(), xg_r.float().contiguous()
        wf, wr = w_hh_f.float().contiguous(), w_hh_r.float().contiguous()
        hs, cs, gates = K.lstm_wide_bidir_fwd(xg_f, xg_r, wf, wr, batch_sizes)
        ctx.save_for_backward(wf, wr, hs[0], hs[1], cs[0], cs[1], gates[0], gates[1])
        ctx.batch_sizes = tuple(batch_sizes)
        return hs[0], hs[1]

    @staticmethod
    def backward(ctx, dhs_f, dhs_r):
        wf, wr, hs_f, hs_r, cs_f, cs_r, gates_f, gates_r = ctx.saved_tensors
        T, B, H = hs_f.shape
        zf = lambda d, ref: torch.zeros_like(ref) if d is None else d.float().contiguous()
        dg_f, dg_r = K.lstm_wide_bidir_bwd(wf.t().contiguous(), wr.t().contiguous(), ctx.batch_sizes, gates_f, gates_r, cs_f,
                                           cs_r, zf(dhs_f, hs_f), zf(dhs_r, hs_r))
        zero = torch.zeros(1, B, H, device=hs_f.device)
        hp_f = torch.cat([zero, hs_f[:-1]], 0)           # h of each step's predecessor in its chain
        hp_r = torch.cat([hs_r[1:], zero], 0)
        dw_f = K.gemm_tn(dg_f.reshape(T * B, 4 * H), hp_f.reshape(T * B, H).contiguous())
        dw_r = K.gemm_tn(dg_r.reshape(T * B, 4 * H), hp_r.reshape(T * B, H).contiguous())
        return dg_f, dg_r, dw_f, dw_r, None


def lstm_wide_bidir(xg_f, xg_r, w_hh_f, w_hh_r, batch_sizes):
    return LstmWideBidirFn.apply(xg_f, xg_r, w_hh_f, w_hh_r, batch_sizes)


def packed_batch_sizes(lens_sorted, n_steps=None):
    """PackedSequence.batch_sizes of host lengths sorted descending."""
    lens = [int(v) for v in lens_sorted]
    n_steps = n_steps or lens[0]
    return [sum(1 for v in lens if v > t) for t in range(n_steps)]


class MacReadState(object):
    """Shared by the reasoning steps of ONE MACNetwork forward: collects each step's backward factors so that the
    knowledge-base gradients are formed once (see MacReadFn)."""

    def __init__(self):
        self.n_calls = 0
        self.factors = []


class MacReadFn(torch.autograd.Function):
    """read = ReadUnit attention of one reasoning step on the fused HIP kernel (models/mac.py:53-62 re-associated):
    softmax_s(know.u + pre.v + bias) weighted sum of know.  know/pre [n*s, ld] in the compute dtype, u/v fp32 [n,c].
    With pre = v = None it is a plain attention pool (ControlUnit's attention over the question words, :36-42).

    Backward: du, dv (and the analytically zero bias gradient) per step; the [n,s,c]-sized gradients of know and pre
    are outer products of small per-step factors, so steps > 0 only stash their factors and the FIRST step's backward
    — which the engine necessarily runs last, every later step depends on its output through the memory chain —
    forms both gradients for all steps in one pass (K.mac_read_accum)."""

    @staticmethod
    def forward(ctx, know, pre, u, v, bias, state, s, c):
        n = know.shape[0] // s
        u = u.float().contiguous()
        v = None if v is None else v.float().contiguous()
        p, read = K.mac_read_fwd(know, pre, u, v, bias.detach().float().contiguous(), n, s, c)
        ctx.save_for_backward(know, pre, u, v, p)
        ctx.state, ctx.dims, ctx.index = state, (n, s, c), state.n_calls
        state.n_calls += 1
        return read

    @staticmethod
    def backward(ctx, dread):
        know, pre, u, v, p = ctx.saved_tensors
        n, s, c = ctx.dims
        dread = dread.float().contiguous()
        dscore, du, dv = K.mac_read_bwd(know, pre, p, dread, n, s, c)
        st = ctx.state
        st.factors.append((dscore, p, u, v, dread))
        dknow = dpre = None
        if ctx.index == 0:
            f = [None if t[0] is None else torch.stack(t) for t in zip(*st.factors)]
            dknow, dpre = K.mac_read_accum(f[0], f[1], f[2], f[3], f[4], n, s, c, know.shape[-1], know.dtype)
            st.factors = []
        return dknow, dpre, du, dv, dscore.sum().view(1), None, None, None


def mac_read(know, pre, u, v, bias, state, s, c):
    return MacReadFn.apply(know, pre, u, v, bias, state, s, c)


class FcNativeFn(torch.autograd.Function):
    """out = flatten_NCHW(map) @ weight.T + bias for a map held as padded NHWC (fc_embed_attn, film_attn_pt_stem.py:244):
    x [N, (h+2)(w+2)*c_pad] in the compute dtype, weight the reference-layout fp32 parameter [rows, C*h*w].  The weight is
    re-laid out by two HIP kernels (forward operand + its transpose for dX); forward / dX / dW are MFMA GEMMs and the
    weight gradient returns to the parameter layout in one kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, C, h, w, rows_pad, grad_scale=1.0, split_weights=False):
        ctx.grad_scale = float(grad_scale)      # d(out) arrives multiplied by this (fp16 loss scale): dW, db are divided by it
        rows = weight.shape[0]
        c_pad = x.shape[1] // ((h + 2) * (w + 2))
        need_dx = ctx.needs_input_grad[0]
        # dX from the forward operand alone (vnqa_fc_dx) where its shapes allow: no transposed weight copy per step
        ctx.direct_dx = need_dx and K.fc_dx_supported(x.shape[0], rows_pad, x.shape[1], x.dtype)
        nat, nat_t = K.pack_fc_weight(weight, C, h, w, c_pad, rows_pad, x.dtype, want_t=need_dx and not ctx.direct_dx)
        bias_p = K.pad_vec(bias, rows_pad)
        if split_weights and L.is_half(x.dtype):
            # two-product forward (precision 'fp16h': this layer's weight rounding is 0.03e-6 of the squared logits error for 7 GFLOP):
            # the fp32 operand, split into [hi | lo] by the GEMM wrapper; `nat` serves the backward
            out = K.gemm_nt(x.contiguous(), K.pack_fc_weight(weight, C, h, w, c_pad, rows_pad, torch.float32, want_t=False)[0], bias=bias_p,
                            split_weights=True)
        else:
            out = K.gemm_nt(x.contiguous(), nat, bias=bias_p)
        ctx.save_for_backward(x, nat if ctx.direct_dx else nat_t)
        ctx.geom = (rows, C, h, w, c_pad)
        with torch.enable_grad():
            ctx.sink_w = sink_of(weight)
            ctx.sink_b = sink_of(bias) if rows == rows_pad else None
        return out

    @staticmethod
    def backward(ctx, dout):
        return FcNativeFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, nat_t = ctx.saved_tensors
        rows, C, h, w, c_pad = ctx.geom
        dout = dout.to(x.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.fc_dx(dout, nat_t) if ctx.direct_dx else K.gemm_nt(dout, nat_t)     # (direct: the saved tensor is `nat`)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _ret(ctx.sink_w, K.unpack_fc_wgrad(K.gemm_tn(dout, x.contiguous()), rows, C, h, w, c_pad, out=_into(ctx.sink_w),
                                                    alpha=1.0 / ctx.grad_scale))
        db = None
        if ctx.needs_input_grad[2]:
            sb = ctx.sink_b if ctx.grad_scale == 1.0 else None
            db = _ret(sb, K.colsum(dout, out=_into(sb))[:rows])
            if ctx.grad_scale != 1.0:
                db = db * (1.0 / ctx.grad_scale)
        return dx, dw, db, None, None, None, None, None, None


def fc_native(x, weight, bias, C, h, w, rows_pad, grad_scale=1.0, split_weights=False):
    return FcNativeFn.apply(x, weight, bias, C, h, w, rows_pad, grad_scale, split_weights)


class MacCoreState(object):
    """Shared by the MacCoreFn nodes of one forward.  `n_steps` (the number of reasoning steps that will be issued) lets the
    C-ABI node keep every per-step matrix STEP-STACKED in two slabs (forward factors, backward factors): field f of step i
    lives at slab[f][i], so [n_steps * N, d] views of a field are free and the last node to run forms all parameter
    gradients (vnqa_mac_core_wgrad) and the [N, positions, C] attention-pool gradients (mac_read_accum) in one call each."""

    def __init__(self, n_steps=None):
        self.n_steps = n_steps
        self.n_calls = 0
        self.ctrl = []
        self.read = []
        self.grads = {}
        self.grads_meta = {}
        self.fwd = None          # {field: [n_steps, N, width]} views of the forward slab
        self.bwd = None
        self.outer = {}          # step index -> (control, memory, d_concat): the factors that arrive from outside the node


class MacCoreFn(torch.autograd.Function):
    """One MAC reasoning step (ControlUnit, ReadUnit and WriteUnit.concat of models/mac.py:28-42,53-62,82-85) for all
    packed images as ONE autograd node whose forward and backward are ONE C-ABI call each (vnqa_mac_core_fwd / _bwd,
    csrc/mac_core.hip): the launches of a step are enqueued from C++, its [N, d] x [d, d] products run on the exact-f32 MFMA
    GEMM, and the parameter gradients of all steps are formed once, after the last step's backward, from step-stacked
    factors (vnqa_mac_core_wgrad: one product over K = steps * N per weight instead of `steps` products on the dependent chain).

      cq      = control Wc^T + pq                       (pq = position_aware_i(question) Wp^T + b, hoisted by the caller)
      control'= pool(ctx, cq * w_ca, b_ca) [* mask]     (attention over the question words)
      mem     = memory Wm^T + bm ;  v = control' * w_ra ;  u = mem * (v W1)
      read    = pool(know, pre; u, v, b_ra)             (re-associated ReadUnit, see models/mac.py)
      concat  = read Wr^T + memory Wmm^T + bw
    Returns (control', concat); self-attention / memory gate / the memory dropout mask stay with the caller."""

    FWD = ("cq", "qv", "cnew", "mem", "v", "t", "u", "read", "concat")          # [N, d] each, then p_c [N, Lq], p_r [N, S]
    BWD = ("d_control", "d_memory", "d_cq", "d_read", "d_c", "du", "dv", "dqv", "d_mem", "d_t")   # then ds_r [N,S], ds_c [N,Lq]

    @staticmethod
    def _slab(names, widths, n_steps, N, dev):
        total = sum(widths)
        buf = torch.empty(n_steps * N * total, dtype=torch.float32, device=dev)      # ONE allocation, field-major
        out, o = {}, 0
        for n, wd in zip(names, widths):
            out[n] = buf[o:o + n_steps * N * wd].view(n_steps, N, wd)
            o += n_steps * N * wd
        return out

    @staticmethod
    def forward(ctx_, control, memory, pq, ctxw, know, pre, mask_c, wc, w_ca, b_ca, wm, bm, w1, w_ra, b_ra, wr, wmm, bw,
                state, Lq, S, step=None):
        N, d = control.shape
        dev = control.device
        assert state.n_steps is not None and state.n_calls < state.n_steps, "MacCoreState(n_steps) too small for this forward"
        # `pq` may be the [steps, N, d] tensor of ALL steps' position-aware terms with `step` selecting this node's slice: its
        # gradient is then the backward slab's d_cq field, handed to autograd ONCE by the node that runs last, instead of twelve
        # slice-backward zero-fills + adds of the whole [steps, N, d] tensor
        ctx_.pq_all_shape = None
        if step is not None:
            assert pq.dim() == 3 and pq.is_contiguous()
            ctx_.pq_all_shape = tuple(pq.shape)
            pq = pq[step]
        if state.fwd is None:
            state.fwd = MacCoreFn._slab(MacCoreFn.FWD + ("p_c", "p_r"), [d] * 9 + [Lq, S], state.n_steps, N, dev)
            nb = L.lib().vnqa_mac_core_workspace(N, d)
            state.grads_meta["ws"] = K.workspace(nb, dev) if nb > 0 else None
        i = state.n_calls
        out = {n: v[i] for n, v in state.fwd.items()}
        f32 = lambda t: t.detach().float().contiguous()
        inputs = dict(control=f32(control), memory=f32(memory), pq=f32(pq), ctxw=ctxw, know=know, pre=pre,
                      mask_c=None if mask_c is None else f32(mask_c), wc=f32(wc), w_ca=f32(w_ca), b_ca=f32(b_ca), wm=f32(wm),
                      bm=f32(bm), w1=f32(w1), w_ra=f32(w_ra), b_ra=f32(b_ra), wr=f32(wr), wmm=f32(wmm), bw=f32(bw))
        dims = (N, d, Lq, S, know.shape[-1], L.dtype_id(know.dtype))
        K.mac_core_call("fwd", dims, dict(inputs, workspace=state.grads_meta["ws"], **out))
        ctx_.save_for_backward(*[inputs[k] for k in ("control", "memory", "ctxw", "know", "pre")],
                               inputs["mask_c"], *[inputs[k] for k in ("wc", "w_ca", "b_ca", "wm", "bm", "w1", "w_ra", "b_ra",
                                                                       "wr", "wmm", "bw")])
        ctx_.state, ctx_.dims, ctx_.index = state, dims, i
        state.n_calls += 1
        return out["cnew"], out["concat"]

    @staticmethod
    def backward(ctx_, d_cnew, d_concat):
        sv = ctx_.saved_tensors
        control, memory, ctxw, know, pre, mask_c = sv[:6]
        wc, w_ca, b_ca, wm, bm, w1, w_ra, b_ra, wr, wmm, bw = sv[6:17]
        N, d, Lq, S, ld, did = ctx_.dims
        dev = control.device
        st, i = ctx_.state, ctx_.index
        n_used = st.n_calls
        if st.bwd is None:
            st.bwd = MacCoreFn._slab(MacCoreFn.BWD + ("ds_r", "ds_c"), [d] * 10 + [S, Lq], n_used, N, dev)
        out = {n: v[i] for n, v in st.fwd.items()}
        g = {n: v[i] for n, v in st.bwd.items()}
        d_concat = d_concat.float().contiguous()
        args = dict(control=control, memory=memory, ctxw=ctxw, know=know, pre=pre, mask_c=mask_c, wc=wc, w_ca=w_ca, b_ca=b_ca,
                    wm=wm, bm=bm, w1=w1, w_ra=w_ra, b_ra=b_ra, wr=wr, wmm=wmm, bw=bw,
                    d_cnew=None if d_cnew is None else d_cnew.float().contiguous(), d_concat=d_concat,
                    workspace=st.grads_meta.get("ws"))
        args.update(out)
        args.update(g)
        K.mac_core_call("bwd", (N, d, Lq, S, ld, did), args, defer_wgrad=True)
        st.outer[i] = (control, memory, d_concat)
        d_ctxw = d_know = d_pre = d_cq_all = None
        gp = [None] * 11
        if i == 0:               # runs last: every later step depends on this one's outputs
            F_, B_ = st.fwd, st.bwd
            d_cq_all = B_["d_cq"][:n_used]
            used = lambda t: t[:n_used]
            d_know, d_pre = K.mac_read_accum(used(B_["ds_r"]), used(F_["p_r"]), used(F_["u"]), used(F_["v"]), used(B_["d_read"]),
                                             N, S, d, ld, know.dtype)
            d_ctxw, _ = K.mac_read_accum(used(B_["ds_c"]), used(F_["p_c"]), used(F_["qv"]), None, used(B_["d_c"]), N, Lq, d,
                                         ctxw.shape[-1], ctxw.dtype)
            rows = n_used * N
            order = range(n_used)
            stack = lambda k: torch.stack([st.outer[j][k] for j in order]).view(rows, d)
            flat = lambda t: used(t).reshape(rows, d)
            gw = torch.empty(5 * d * d + 4 * d, dtype=torch.float32, device=dev)
            names = ("g_wc", "g_wm", "g_w1", "g_wr", "g_wmm")
            G = {n: gw[k * d * d:(k + 1) * d * d].view(d, d) for k, n in enumerate(names)}
            for k, n in enumerate(("g_wca", "g_wra", "g_bm", "g_bw")):
                G[n] = gw[5 * d * d + k * d: 5 * d * d + (k + 1) * d]
            nb = L.lib().vnqa_mac_core_wgrad_workspace(rows, d)
            K.mac_wgrad(rows, d, dict(d_concat=stack(2), read=flat(F_["read"]), memory=stack(1), v=flat(F_["v"]),
                                      d_t=flat(B_["d_t"]), d_mem=flat(B_["d_mem"]), d_cq=flat(B_["d_cq"]), control=stack(0),
                                      dv=flat(B_["dv"]), cnew=flat(F_["cnew"]), dqv=flat(B_["dqv"]), cq=flat(F_["cq"]),
                                      workspace=K.workspace(nb, dev), **G))
            gp = [G["g_wc"], G["g_wca"].view(1, d), used(B_["ds_c"]).sum().view(1), G["g_wm"], G["g_bm"],
                  G["g_w1"], G["g_wra"].view(1, d), used(B_["ds_r"]).sum().view(1), G["g_wr"], G["g_wmm"], G["g_bw"]]
            st.outer, st.fwd, st.bwd = {}, None, None
        d_pq = g["d_cq"]
        if ctx_.pq_all_shape is not None:        # whole-tensor form: only the last node to run returns it (None = zero elsewhere)
            d_pq = None
            if i == 0:
                assert ctx_.pq_all_shape[0] == n_used, "pq_all must hold exactly the steps that were run"
                d_pq = d_cq_all
        return (g["d_control"], g["d_memory"], d_pq, d_ctxw, d_know, d_pre, None, gp[0], gp[1], gp[2], gp[3], gp[4],
                gp[5], gp[6], gp[7], gp[8], gp[9], gp[10], None, None, None, None)


class MacChainFn(torch.autograd.Function):
    """ALL reasoning steps of MACNetwork without self-attention / memory gate (the reference's defaults, models/mac.py:131-155)
    as ONE autograd node: forward and backward are one C-ABI call each (vnqa_mac_chain_fwd / _bwd loop over the steps in C++,
    csrc/mac_core.hip), then vnqa_mac_read_accum x 2 and vnqa_mac_core_wgrad over the step-stacked slabs as in MacCoreFn.
    With the step itself down to 5 + 6 launches the per-step nodes had become host-bound — 24 node invocations of ~90 us of
    Python each, the mask multiplies and gradient sums between them; this node is bit-identical to that chain.

      memory_{i+1} = concat_i * mask_m,  control_{i+1} = control'_i;   returns memory_{n_steps}."""

    @staticmethod
    def forward(ctx_, control, memory, pq_all, ctxw, know, pre, mask_c, mask_m, wc, w_ca, b_ca, wm, bm, w1, w_ra, b_ra, wr, wmm, bw,
                Lq, S):
        N, d = control.shape
        n_steps = pq_all.shape[0]
        dev = control.device
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        F_ = MacCoreFn._slab(MacCoreFn.FWD + ("p_c", "p_r"), [d] * 9 + [Lq, S], n_steps, N, dev)
        # control_i = entry i of `ctl` (entry 0 = the initial control, control'_i written to entry i + 1); likewise the memories
        ctl = torch.empty((n_steps + 1, N, d), dtype=torch.float32, device=dev)
        mems = torch.empty((n_steps + 1, N, d), dtype=torch.float32, device=dev)
        ctl[0].copy_(control)
        mems[0].copy_(memory)
        F_["cnew"] = ctl[1:]
        nb = L.lib().vnqa_mac_core_workspace(N, d)
        ws = K.workspace(nb, dev) if nb > 0 else None
        par = dict(ctxw=ctxw, know=know, pre=pre, mask_c=f32(mask_c), wc=f32(wc), w_ca=f32(w_ca), b_ca=f32(b_ca), wm=f32(wm),
                   bm=f32(bm), w1=f32(w1), w_ra=f32(w_ra), b_ra=f32(b_ra), wr=f32(wr), wmm=f32(wmm), bw=f32(bw))
        pq_all = f32(pq_all)
        mask_m = f32(mask_m)
        dims = (N, d, Lq, S, know.shape[-1], L.dtype_id(know.dtype))
        step0 = dict(par, control=ctl[0], memory=mems[0], pq=pq_all[0], workspace=ws, **{n: v[0] for n, v in F_.items()})
        K.mac_chain_call("fwd", dims, n_steps, step0, mems, mask_m)
        ctx_.save_for_backward(ctxw, know, pre, par["mask_c"], mask_m, pq_all, ctl, mems,
                               *[par[k] for k in ("wc", "w_ca", "b_ca", "wm", "bm", "w1", "w_ra", "b_ra", "wr", "wmm", "bw")])
        ctx_.fwd, ctx_.dims, ctx_.ws = F_, dims, ws
        return mems[n_steps]

    @staticmethod
    def backward(ctx_, d_out):
        sv = ctx_.saved_tensors
        ctxw, know, pre, mask_c, mask_m, pq_all, ctl, mems = sv[:8]
        names = ("wc", "w_ca", "b_ca", "wm", "bm", "w1", "w_ra", "b_ra", "wr", "wmm", "bw")
        par = dict(zip(names, sv[8:19]))
        N, d, Lq, S, ld, did = ctx_.dims
        n_steps = pq_all.shape[0]
        dev = ctl.device
        F_ = ctx_.fwd
        B_ = MacCoreFn._slab(MacCoreFn.BWD + ("ds_r", "ds_c", "d_concat"), [d] * 10 + [S, Lq, d], n_steps, N, dev)
        step0 = dict(par, ctxw=ctxw, know=know, pre=pre, mask_c=mask_c, control=ctl[0], memory=mems[0], pq=pq_all[0],
                     workspace=ctx_.ws, **{n: v[0] for n, v in F_.items()})
        step0.update({n: v[0] for n, v in B_.items() if n != "d_concat"})
        K.mac_chain_call("bwd", (N, d, Lq, S, ld, did), n_steps, step0, mems, mask_m, d_out.float().contiguous(), B_["d_concat"])
        d_know, d_pre = K.mac_read_accum(B_["ds_r"], F_["p_r"], F_["u"], F_["v"], B_["d_read"], N, S, d, ld, know.dtype)
        d_ctxw, _ = K.mac_read_accum(B_["ds_c"], F_["p_c"], F_["qv"], None, B_["d_c"], N, Lq, d, ctxw.shape[-1], ctxw.dtype)
        rows = n_steps * N
        flat = lambda t: t.reshape(rows, d)
        gw = torch.empty(5 * d * d + 4 * d, dtype=torch.float32, device=dev)
        G = {n: gw[k * d * d:(k + 1) * d * d].view(d, d) for k, n in enumerate(("g_wc", "g_wm", "g_w1", "g_wr", "g_wmm"))}
        for k, n in enumerate(("g_wca", "g_wra", "g_bm", "g_bw")):
            G[n] = gw[5 * d * d + k * d: 5 * d * d + (k + 1) * d]
        nb = L.lib().vnqa_mac_core_wgrad_workspace(rows, d)
        K.mac_wgrad(rows, d, dict(d_concat=flat(B_["d_concat"]), read=flat(F_["read"]), memory=flat(mems[:n_steps]), v=flat(F_["v"]),
                                  d_t=flat(B_["d_t"]), d_mem=flat(B_["d_mem"]), d_cq=flat(B_["d_cq"]), control=flat(ctl[:n_steps]),
                                  dv=flat(B_["dv"]), cnew=flat(ctl[1:]), dqv=flat(B_["dqv"]), cq=flat(F_["cq"]),
                                  workspace=K.workspace(nb, dev), **G))
        ctx_.fwd = None
        return (B_["d_control"][0], B_["d_memory"][0], B_["d_cq"], d_ctxw, d_know, d_pre, None, None,
                G["g_wc"], G["g_wca"].view(1, d), B_["ds_c"].sum().view(1), G["g_wm"], G["g_bm"], G["g_w1"], G["g_wra"].view(1, d),
                B_["ds_r"].sum().view(1), G["g_wr"], G["g_wmm"], G["g_bw"], None, None)


def mac_chain(control, memory, pq_all, ctxw, know, pre, mask_c, mask_m, *weights_and_dims):
    return MacChainFn.apply(control, memory, pq_all, ctxw, know, pre, mask_c, mask_m, *weights_and_dims)


# MAC_CHAIN: all reasoning steps as ONE autograd node (MacChainFn) where the model's options allow it; False = one node per step
# (MacCoreFn) — the cross-check tests/test_gpu_mac.py runs (they are bit-identical)
MAC_CHAIN = True


def mac_core(control, memory, pq_all, step, *rest):
    """One reasoning step; `pq_all` [steps, N, d] holds every step's position-aware term and `step` selects this one: the C-ABI
    node (MacCoreFn: one call per direction, exact-f32 MFMA products with the elementwise products in their epilogues, parameter
    gradients deferred to one vnqa_mac_core_wgrad call).  (The op-by-op torch / rocBLAS form of the step is test infrastructure:
    tests/torch_partners.py.)"""
    return MacCoreFn.apply(control, memory, pq_all, *rest, step)


# ---- question path / classifier / loss on csrc/glue.hip (no ATen / rocBLAS kernels in the step) --------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x_sel @ w.T + b) in exact fp32 on vnqa_sgemm; `rows` (int32, optional) gathers the rows of x first —
    the LSTM output at the last token of every repeat (film_attn_pt_stem.py:163-171) feeding the FiLM generator's
    Linear+ReLU (:179).  Replaces nn.Linear of film_layer[1], lstm_attn's input half, out_linear and their backward."""

    @staticmethod
    def forward(ctx, x, w, b, relu, rows):
        x = x.float() if x.dtype != torch.float32 else x
        assert x.dim() == 2 and x.stride(1) == 1
        xs = K.gather_rows(x.contiguous(), rows) if rows is not None else x
        y = K.linear_nt(xs, w, bias=b, relu=relu)
        ctx.save_for_backward(xs, w, y if relu else None, rows)
        ctx.relu, ctx.x_rows = relu, x.shape[0]
        with torch.enable_grad():
            ctx.sinks = (sink_of(w), sink_of(b) if b is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, w, y, rows = ctx.saved_tensors
        dy = dy.contiguous()
        mask = y if ctx.relu else None
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if rows is not None:      # adjoint of the row gather: scatter into zeros (every source row is used at most once)
                dx = torch.zeros((ctx.x_rows, w.shape[1]), dtype=torch.float32, device=dy.device)
                K.matmul_nn(dy, w, a_mask=mask, out=dx, c_rows=rows)
            else:
                dx = K.matmul_nn(dy, w, a_mask=mask)
        sw, sb = ctx.sinks
        if ctx.needs_input_grad[1]:
            dw = _ret(sw, K.matmul_tn(dy, xs, a_mask=mask, out=_into(sw)))
        if ctx.needs_input_grad[2]:
            db = _ret(sb, K.colsum(dy, mask, out=_into(sb)))
        return dx, dw, db, None, None


def linear(x, w, b, relu=False, rows=None):
    return LinearFn.apply(x, w, b, relu, rows)


class MultiHopGenFn(torch.autograd.Function):
    """The multi-hop FiLM generator (time_multi_hop_pt_stem.py:146-184) for every packed image as ONE autograd node on the HIP
    kernels of csrc/hop_gen.hip + vnqa_sgemm:
        enc = encoder_norm(hs[last_rows]);  per block: hv = hop(hv, states) ; film_k = decoder_norm(fc_attn_out(hv))
    hs [B*S, H] is the persistent LSTM chain's output; image i's word states are its rows base_row[i] .. + qlen[i] - 1.
    Returns one FiLM matrix [n_img, N] per block.  The backward accumulates d hs in ONE buffer (no per-use gradient tensors)."""

    @staticmethod
    def forward(ctx, hs, tables, lmax, blocks, eps_enc, eps_dec, enc_g, enc_b, w_att, b_att, w_out, b_out, dec_g, dec_b):
        last_rows, base_row, qlen = tables
        enc, m0, r0 = K.layernorm_fwd(hs, enc_g, enc_b, eps_enc, rows=last_rows)
        hv, saved, films = enc, [], []
        for _ in range(blocks):
            hv_next, coefs = K.hop_fwd(hv, hs, base_row, qlen, w_att, b_att, lmax)
            y = K.linear_nt(hv_next, w_out, bias=b_out)
            film, m, r = K.layernorm_fwd(y, dec_g, dec_b, eps_dec)
            saved += [hv, coefs, hv_next, y, m, r]
            films.append(film)
            hv = hv_next
        ctx.save_for_backward(hs, last_rows, base_row, qlen, m0, r0, enc_g, w_att, w_out, dec_g, *saved)
        ctx.blocks = blocks
        with torch.enable_grad():
            ctx.sinks = [sink_of(t) for t in (enc_g, enc_b, w_att, w_out, b_out, dec_g, dec_b)]
        return tuple(films)

    @staticmethod
    def backward(ctx, *dfilms):
        hs, last_rows, base_row, qlen, m0, r0, enc_g, w_att, w_out, dec_g = ctx.saved_tensors[:10]
        saved = ctx.saved_tensors[10:]
        s_eg, s_eb, s_wa, s_wo, s_bo, s_dg, s_db = ctx.sinks
        dev, H, N = hs.device, hs.shape[1], w_out.shape[0]
        n_img = last_rows.numel()
        dhs = torch.zeros_like(hs)
        ones = torch.ones((n_img, 1), dtype=torch.float32, device=dev)
        def pair(a, b):       # gamma / beta gradients are written by ONE kernel: both in place or both through temporaries
            return a is not None and b is not None and not a.written and not b.written
        if pair(s_dg, s_db):
            d_dec_g, d_dec_b = _into(s_dg), _into(s_db)
        else:
            d_dec_g = torch.empty((N,), dtype=torch.float32, device=dev)
            d_dec_b = torch.empty((N,), dtype=torch.float32, device=dev)
            s_dg = s_db = None
        d_wo = _into(s_wo)
        d_wo = d_wo if d_wo is not None else torch.empty((N, H), dtype=torch.float32, device=dev)
        d_bo = _into(s_bo)
        d_bo = d_bo if d_bo is not None else torch.empty((N,), dtype=torch.float32, device=dev)
        d_wa = _into(s_wa, (1, H))
        d_wa = d_wa if d_wa is not None else torch.empty((1, H), dtype=torch.float32, device=dev)
        carry, first = None, True
        for k in reversed(range(ctx.blocks)):
            hv_in, coefs, hv_next, y, m, r = saved[6 * k: 6 * k + 6]
            df = dfilms[k]
            if df is None:
                df = torch.zeros((n_img, N), dtype=torch.float32, device=dev)
            dy, _, _ = K.layernorm_bwd(df.float().contiguous(), y, m, r, dec_g, dgamma=d_dec_g, dbeta=d_dec_b, accumulate=not first)
            K.matmul_tn(dy, hv_next, out=d_wo, accumulate=not first)                       # d W_out += dy^T hv
            K.matmul_tn(ones, dy, out=d_bo.view(1, N), accumulate=not first)               # d b_out += colsum(dy)
            if carry is None:
                carry = K.matmul_nn(dy, w_out)                                             # d hv_next (last block: nothing carried)
            else:
                K.matmul_nn(dy, w_out, out=carry, accumulate=True)                         # + the next hop's d hv_in
            carry, dw_img = K.hop_bwd(carry, hv_in, hs, base_row, qlen, w_att, coefs, dhs)
            K.matmul_tn(ones, dw_img, out=d_wa, accumulate=not first)
            first = False
        both_e = pair(s_eg, s_eb)
        dx_enc, d_enc_g, d_enc_b = K.layernorm_bwd(carry, hs, m0, r0, enc_g, rows=last_rows,
                                                   dgamma=_into(s_eg) if both_e else None, dbeta=_into(s_eb) if both_e else None)
        K.scatter_add_rows(dhs, last_rows, dx_enc)
        return (dhs, None, None, None, None, None,
                _ret(s_eg if both_e else None, d_enc_g), _ret(s_eb if both_e else None, d_enc_b),
                _ret(s_wa, d_wa), torch.zeros((1,), dtype=torch.float32, device=dev),          # d b_att == 0 (softmax shift invariance)
                _ret(s_wo, d_wo), _ret(s_bo, d_bo), _ret(s_dg, d_dec_g), _ret(s_db, d_dec_b))


def multi_hop_generator(hs, tables, lmax, blocks, enc_norm, fc_hidden_attn, fc_attn_out, dec_norm):
    return MultiHopGenFn.apply(hs, tables, lmax, blocks, enc_norm.eps, dec_norm.eps, enc_norm.weight, enc_norm.bias,
                               fc_hidden_attn.weight, fc_hidden_attn.bias, fc_attn_out.weight, fc_attn_out.bias,
                               dec_norm.weight, dec_norm.bias)


class FrameMaxFn(torch.autograd.Function):
    """max over a sample's frames of the relu'd tail maps, straight from the packed image list (the reference's zero-padded
    [T, B, ...] stack + max(dim=0), film_global_pooling_pt_stem.py:230-235): (pooled, argmax) = vnqa_frame_max_fwd; the backward
    routes d pooled to the arg-max image (x `grad_scale`, the fp16 loss scale of the maps' gradient)."""

    @staticmethod
    def forward(ctx, maps, lay, tail, grad_scale, route=None):
        pooled, argmax = K.frame_max_fwd(maps, lay.frame_off_i32, lay.B, lay.n_frames, tail)
        # `route` (diagnostics, bench.precision_parity): arg-max table of ANOTHER run to route the gradient by
        ctx.save_for_backward(argmax if route is None else route)
        ctx.lay, ctx.shape, ctx.dtype, ctx.tail, ctx.gs = lay, tuple(maps.shape), maps.dtype, tail, float(grad_scale)
        ctx.mark_non_differentiable(argmax)
        return pooled, argmax

    @staticmethod
    def backward(ctx, dpooled, _dargmax):
        (argmax,) = ctx.saved_tensors
        dmaps = K.frame_max_bwd(dpooled.float(), argmax, ctx.lay.sample_of_i32, ctx.shape, ctx.dtype, ctx.tail, ctx.gs)
        return dmaps, None, None, None, None


def frame_max(maps, lay, tail, grad_scale=1.0, route=None):
    return FrameMaxFn.apply(maps, lay, tail, grad_scale, route)


class EmbedProjFn(torch.autograd.Function):
    """xg[b,pos] = W_ih embed[tokens[b,pos]] + b_ih + b_hh: nn.Embedding (film_attn_pt_stem.py:146) fused with the input half
    of nn.LSTM (:160).  Backward through per-token sums of d xg (rows holding the same token share their embedding):
    d embed = dsum W_ih, d W_ih = dsum^T embed, d b_ih = d b_hh = colsum(dsum); `padding_idx` row gets no gradient."""

    @staticmethod
    def forward(ctx, tokens, embed, w_ih, b_ih, b_hh, padding_idx):
        xg, rows = K.embed_proj_fwd(tokens.contiguous(), None, embed, w_ih, b_ih, b_hh)
        ctx.save_for_backward(rows, embed, w_ih)
        ctx.padding_idx = padding_idx
        with torch.enable_grad():
            ctx.sinks = [sink_of(t) for t in (embed, w_ih, b_ih, b_hh)]
        return xg

    @staticmethod
    def backward(ctx, dxg):
        rows, embed, w_ih = ctx.saved_tensors
        V = embed.shape[0]
        se, sw, sb1, sb2 = ctx.sinks
        dsum = K.token_dsum(rows, dxg.contiguous(), V)                              # [V, 4H]
        dembed = K.matmul_nn(dsum, w_ih, out=_into(se))                             # [V, E]
        if ctx.padding_idx is not None:
            dembed[ctx.padding_idx].zero_()
        dw = K.matmul_tn(dsum, embed, out=_into(sw))                                # [4H, E]
        db1 = K.colsum(dsum, out=_into(sb1))
        db2 = K.colsum(dsum, out=_into(sb2)) if sb2 is not None else db1
        return None, _ret(se, dembed), _ret(sw, dw), _ret(sb1, db1), _ret(sb2, db2), None


def embed_proj(tokens, embed, w_ih, b_ih, b_hh, padding_idx=None):
    return EmbedProjFn.apply(tokens, embed, w_ih, b_ih, b_hh, padding_idx)


class TemporalAttnPackedFn(torch.autograd.Function):
    """ctxt, coef = temporal attention (film_attn_pt_stem.py:245-290) straight from the packed fc_embed_attn output
    f [n_img, ld] (compute dtype): no dense [B,T,A] scatter, validity grid or mask tensors."""

    @staticmethod
    def forward(ctx, f, frame_off_i32, n_frames, B, T, A, w, bias, grad_scale=1.0):
        ctx.grad_scale = float(grad_scale)      # d f leaves the backward kernel multiplied by this (fp16 loss scale)
        w1 = w.reshape(-1)
        b1 = bias.reshape(-1)
        coef, ctxt = K.temporal_attn_packed_fwd(f, frame_off_i32, n_frames, B, T, A, w1, b1)
        ctx.save_for_backward(f, frame_off_i32, w1, coef)
        ctx.dims, ctx.w_shape = (n_frames, B, T, A), w.shape
        with torch.enable_grad():
            ctx.sinks = (sink_of(w), sink_of(bias))
        ctx.mark_non_differentiable(coef)
        return ctxt, coef

    @staticmethod
    def backward(ctx, dctxt, _dcoef):
        f, frame_off_i32, w1, coef = ctx.saved_tensors
        n_frames, B, T, A = ctx.dims
        df, dw_part, db_part = K.temporal_attn_packed_bwd(f, frame_off_i32, n_frames, B, T, A, w1, coef, dctxt.contiguous(),
                                                          ctx.grad_scale)
        sw, sb = ctx.sinks
        dw = K.colsum(dw_part, out=None if sw is None else sw.view.view(-1)).view(ctx.w_shape)
        db = K.colsum(db_part, out=None if sb is None else sb.view.view(-1))
        return df, None, None, None, None, None, _ret(sw, dw), _ret(sb, db), None


def temporal_attention_packed(f, frame_off_i32, n_frames, B, T, A, w, bias, grad_scale=1.0):
    return TemporalAttnPackedFn.apply(f, frame_off_i32, n_frames, B, T, A, w, bias, grad_scale)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(weight, reduction = sum | mean) of eval/q_and_v_eval.py:124 as one HIP launch producing the loss
    and d logits; `row_perm` (int32) reads the targets through the batch sort of :113-116."""

    @staticmethod
    def forward(ctx, logits, ys, row_perm, weight, mean):
        loss, dlogits = K.ce_loss(logits.float().contiguous(), ys, row_perm, weight, mean)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss, None, None, None, None


def cross_entropy(logits, ys, row_perm=None, weight=None, reduction="sum"):
    assert reduction in ("sum", "mean"), reduction
    return CrossEntropyFn.apply(logits, ys, row_perm, weight, reduction == "mean")
