"""Shared machinery of the three FiLM video-question models (host side).

The reference walks frames one by one in Python (models/film_attn_pt_stem.py:201-251): per
frame a `ct_batch_size` prefix of the (length-sorted) batch goes through conv_init -> ReLU ->
train-mode BN -> FiLM residual blocks.  Here ALL valid (sample, frame) pairs of the minibatch
are packed into one image list — frame-major, so the images of one frame are contiguous — and
every conv runs once over the whole list on the MFMA igemm.  Everything that made the
per-frame loop necessary is re-expressed on the packed list:
  * per-frame BatchNorm statistics  -> segmented reduction over the image->frame map;
  * per-frame FiLM gamma/beta       -> a gather [frame, sample] per image;
  * the question LSTM re-run per frame with carried state (film_attn_pt_stem.py:160,213)
                                    -> ONE long sequence per sample (its tokens repeated
                                       n_frames times), outputs sampled at each repeat's end.
"""
import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import kernels as K
from .. import ops

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def compute_dtype(precision):
    if precision in ("bf16", torch.bfloat16):
        L.set_half("bf16")
        return torch.bfloat16
    if precision in ("fp16", "f16", torch.float16):      # the fp16-storage build of the library (csrc/vnqa_common.h)
        L.set_half("f16")
        return torch.float16
    if precision in ("fp32", "f32", torch.float32):
        return torch.float32
    if precision == "fp16h":
        # fp16 storage, arithmetic and loss-scaled backward exactly as 'fp16', with the FEW roundings that dominate the logits error
        # removed where that is cheap (profiles/r05_precision_budget.txt): [hi | lo] pair activations on the stem's last three
        # tensors, conv_init as three products on the pair features against split weights, the frozen 1x1 conv and fc_embed_attn
        # with split weights (two products).  The tolerance mode of round 5.
        L.set_half("f16")
        return torch.float16
    raise ValueError("precision must be 'bf16', 'fp16', 'fp16h' or 'fp32' (got %r)" % (precision,))


# Loss scale of the fp16-storage precision: the activation gradients that enter the conv trunk from the attention tail are
# 1e-5 .. 1e-7 here — at or below fp16's normal range (6.1e-5) — so the tail's backward kernel emits them multiplied by
# 2^10 (exact) and every fp32 result computed from them (weight / bias / gamma / beta gradients) is divided by it again.
# 2^10 is the INITIAL value: train.DynamicLossScale (owned by the Trainer) halves it when the fused clip+Adam kernel reports a
# non-finite gradient norm (that step's update is skipped on the device) and doubles it after a run of clean steps.
FP16_GRAD_SCALE = 1024.0
_LOSS_SCALE = {"fp16": FP16_GRAD_SCALE}


def grad_scale_of(dtype):
    return _LOSS_SCALE["fp16"] if dtype == torch.float16 else 1.0


def set_fp16_loss_scale(value):
    """The loss scale the NEXT forward/backward of the fp16-storage models uses (a power of two: exact scaling)."""
    _LOSS_SCALE["fp16"] = float(value)


def reference_init_(module):
    """The reference's `weights_init` rule applied to one module (film_attn_pt_stem.py:111-127, repeated in
    every model file upstream): Xavier-uniform weights + zero bias for Linear/Conv2d; for nn.LSTM
    Xavier input weights, orthogonal recurrent weights, forget-gate bias 1 in both bias vectors and then
    bias_ih zeroed.  Conv3d / BatchNorm / LayerNorm / Embedding keep PyTorch's defaults (as upstream)."""
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        nn.init.xavier_uniform_(module.weight)
        nn.init.zeros_(module.bias)
    elif isinstance(module, nn.LSTM):
        with torch.no_grad():
            nn.init.xavier_uniform_(module.weight_ih_l0)
            nn.init.orthogonal_(module.weight_hh_l0)
            h = module.hidden_size
            for b in (module.bias_ih_l0, module.bias_hh_l0):
                b[h:2 * h] = 1.0
            module.bias_ih_l0.zero_()


class FrameLayout(object):
    """Packed image list for one minibatch: image n <-> (frame t, sample b), frame-major.
    cts[t] = #videos with v_len >= t+1 (film_attn_pt_stem.py:201-208); v_lens sorted descending."""

    def __init__(self, v_lens, num_frames, device, perm=None, on_device=True):
        """perm (optional): sorted position s holds ORIGINAL sample perm[s] (the batch sort of
        eval/q_and_v_eval.py:113-116); img_of is then indexed by the original sample order, so the
        clip tensor itself never needs to be permuted."""
        vl = [int(v) for v in (v_lens.tolist() if torch.is_tensor(v_lens) else v_lens)]
        pm = list(range(len(vl))) if perm is None else [int(i) for i in perm]
        assert all(vl[i] >= vl[i + 1] for i in range(len(vl) - 1)), "v_lens must be sorted descending"
        B = len(vl)
        self.B, self.T = B, int(num_frames)
        self.cts = []
        for i in range(self.T):
            ct = sum(1 for v in vl if v >= i + 1)
            if ct == 0:
                break
            self.cts.append(ct)
        self.n_frames = len(self.cts)
        self.offsets = [0]
        for ct in self.cts:
            self.offsets.append(self.offsets[-1] + ct)
        self.n_img = self.offsets[-1]
        n_img, nf = self.n_img, self.n_frames
        dev = torch.device(device)
        if dev.type == "cuda" and B <= L.LAYOUT_MAX_BATCH and self.T <= 1024 and on_device:
            # the tables are written ON THE DEVICE by one small kernel whose inputs (sorted lengths, sort permutation) travel as kernel
            # arguments (vnqa_frame_layout): no host-to-device copy on the stem's stream — with the clips arriving over PCIe a 4-KB
            # pinned-memory copy there queued behind the 3-ms clip transfer and held the stem's first kernel (profiles/r04_h2d.txt)
            import ctypes
            packed = torch.empty(B * self.T + 2 * n_img + nf + 1, dtype=torch.int32, device=dev)
            o1, o2, o3 = B * self.T, B * self.T + n_img, B * self.T + 2 * n_img
            base = packed.data_ptr()
            varr = (ctypes.c_int32 * B)(*vl)
            parr = (ctypes.c_int32 * B)(*pm)
            L.check(L.lib().vnqa_frame_layout(varr, parr, B, self.T, ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * o1),
                                              ctypes.c_void_p(base + 4 * o2), ctypes.c_void_p(base + 4 * o3), L.stream()),
                    "vnqa_frame_layout")
        else:
            img_of = np.full((B * self.T,), -1, np.int32)
            frame_of, sample_of = [], []
            for t, ct in enumerate(self.cts):
                for b in range(ct):
                    img_of[pm[b] * self.T + t] = self.offsets[t] + b
                    frame_of.append(t)
                    sample_of.append(b)
            # ONE pinned staging buffer and ONE asynchronous H2D copy for all index tables: a pageable-memory copy would
            # block the launch thread until the stream it is issued on has drained (that stream holds a whole stem pass)
            packed = torch.from_numpy(np.concatenate([img_of, np.asarray(frame_of, np.int32), np.asarray(sample_of, np.int32),
                                                      np.asarray(self.offsets, np.int32)]))
            if dev.type == "cuda":
                packed = packed.pin_memory().to(dev, non_blocking=True)
        o1, o2, o3 = B * self.T, B * self.T + n_img, B * self.T + 2 * n_img
        self.img_of = packed[:o1]
        self.frame_of_i32 = packed[o1:o2]
        self.frame_off_i32 = packed[o3:o3 + nf + 1]
        self.sample_of_i32 = packed[o2:o3]
        self.uniform = all(ct == B for ct in self.cts)
        self._device_tensors = [packed]
        self._lazy = {}

    # int64 / float forms of the tables: built on first use (the fused training path never needs them)
    def _lazily(self, name, make):
        t = self._lazy.get(name)
        if t is None:
            t = self._lazy[name] = make()
            self._device_tensors.append(t)
        return t

    @property
    def frame_of(self):
        return self._lazily("frame_of", lambda: self.frame_of_i32.long())

    @property
    def sample_of(self):
        return self._lazily("sample_of", lambda: self.sample_of_i32.long())

    @property
    def cts_t(self):
        return self._lazily("cts_t", lambda: (self.frame_off_i32[1:] - self.frame_off_i32[:-1]).float())

    def last_token_rows(self, q_lens_cpu, S):
        """Row of the flattened LSTM output [B*S, H] holding sample b's state at the LAST token of its (t+1)-th repeat,
        for every image n = (frame t, sample b) in packing order (film_attn_pt_stem.py:163-171 evaluated per frame)."""
        ql = [int(v) for v in q_lens_cpu]
        return np.asarray([b * S + (t + 1) * ql[b] - 1 for t, ct in enumerate(self.cts) for b in range(ct)], np.int32)

    def record_stream(self, stream):
        """The tables were allocated on the stream current at construction (the stem's side stream when prefetched);
        tell the caching allocator that `stream` reads them too, so their memory is not recycled under its kernels."""
        for t in self._device_tensors:
            if t.is_cuda:
                t.record_stream(stream)


class NativeFeatures(object):
    """Stem output kept in kernel-native form: padded NHWC [n_img, h+2, w+2, Cpad] packed by `layout`."""

    def __init__(self, data, layout, channels, h, w, segs=None, shift=None):
        self.data, self.layout, self.channels, self.h, self.w = data, layout, channels, h, w
        # shift (fp32 [c_pad] on the device, or None): MEAN-SHIFTED storage — `data` holds feature - shift[c] and -shift[c] in its halo
        # (stem.FrozenStem.feature_shift); the trunk's conv_init absorbs sum(W) shift in its bias
        self.shift = shift
        # 1: a plain tensor; 3: a SPLIT tensor [hi | lo | hi] (the stem's dual epilogue); 2: the mean-shifted features written TWICE,
        # [x' | x'] (precision 'fp16h': conv_init's two products against split weights).  An explicit marker
        # (consumers used to infer it from the channel count alone); None = stated by nobody, checked against the shape
        if segs is None:
            segs = 3 if (L.is_half(data.dtype) and data.shape[-1] == 3 * L.round_up(channels, 64)) else 1
        self.segs = int(segs)
        assert self.segs in (1, 2, 3) and data.shape[-1] % (64 * self.segs) == 0 and (self.segs == 1 or L.is_half(data.dtype)), \
            "features tensor [..., %d] (%s) cannot be a %d-segment tensor" % (data.shape[-1], data.dtype, self.segs)

    @property
    def shape(self):  # what the reference tensor [B, C, h, w, T] would report
        return (self.layout.B, self.channels, self.h, self.w, self.layout.T)


def interior_mask(hp, wp, device, dtype=torch.float32):
    m = torch.zeros(1, hp, wp, 1, device=device, dtype=dtype)
    m[:, 1:-1, 1:-1, :] = 1
    return m


def pad_channels(v, c_pad):
    """[..., C] -> [..., c_pad] zero-padded."""
    c = v.shape[-1]
    return v if c == c_pad else F.pad(v, (0, c_pad - c))


def frame_batchnorm(r, bn, layout, training, cdt):
    """BatchNorm2d applied per frame to relu(conv_init) (film_attn_pt_stem.py:211) on fused HIP kernels.
    Train mode: batch statistics over the ct_B*h*w values of each (frame, channel); running statistics
    advanced once per frame in frame order (closed form of the 35 sequential EMA updates).
    r: padded NHWC (zero halo), the output of a ReLU whose mask this op's backward applies."""
    N, hp, wp, c_pad = r.shape
    C = bn.num_features
    S = (hp - 2) * (wp - 2)
    gamma = pad_channels(bn.weight, c_pad)
    beta = pad_channels(bn.bias, c_pad)
    if training:
        y, mean, var = ops.frame_bn_train(r, gamma, beta, layout.frame_of_i32, layout.frame_off_i32,
                                          layout.n_frames, BN_EPS, True)
        with torch.no_grad():
            T = layout.n_frames
            cnt = (layout.cts_t * S).unsqueeze(1)
            decay = (1.0 - BN_MOMENTUM) ** torch.arange(T - 1, -1, -1, device=r.device, dtype=torch.float32)
            coef = (BN_MOMENTUM * decay).unsqueeze(1)                          # weight of frame t's statistic
            unbias = cnt / torch.clamp(cnt - 1, min=1.0)
            keep = (1.0 - BN_MOMENTUM) ** T
            bn.running_mean.mul_(keep).add_((coef * mean[:, :C]).sum(0))
            bn.running_var.mul_(keep).add_((coef * (var * unbias)[:, :C]).sum(0))
            bn.num_batches_tracked += T
        return y
    # eval: running statistics for every frame (one pseudo-frame holding all images)
    mean = pad_channels(bn.running_mean, c_pad).float().view(1, c_pad).contiguous()
    rstd = torch.rsqrt(pad_channels(bn.running_var, c_pad).float() + BN_EPS).view(1, c_pad).contiguous()
    zeros = torch.zeros(N, dtype=torch.int32, device=r.device)
    return K.frame_bn_apply(r, zeros, mean, rstd, gamma.detach().float().contiguous(),
                            beta.detach().float().contiguous())


def film_relu_residual(z, res, gamma, beta, cdt):
    """relu(gamma * z + beta) + res, gamma/beta per (image, channel) (film_attn_pt_stem.py:231-241)."""
    c_pad = z.shape[-1]
    return ops.film_relu_res(z, res, pad_channels(gamma, c_pad), pad_channels(beta, c_pad))


def repeated_question_lstm(lstm, emb, q_lens, n_frames, h0, c0, want_states=False, wgrad_dtype=torch.float32):
    """The question LSTM re-run once per frame with its state carried over
    (film_attn_pt_stem.py:146-171 called from :213) == one chain per sample made of its q_len
    tokens repeated n_frames times — ONE persistent HIP launch (ops.lstm_seq).
    emb [B,L,E]; h0,c0 [B,H] per-sample.  Returns h_last [B,n_frames,H] (output at the last
    token of each repeat), optional per-frame states [B,n_frames,Lmax,H] (zero past q_len),
    and the final (h,c) per sample."""
    B, Lq, E = emb.shape
    dev = emb.device
    ql_cpu = q_lens.detach().cpu().long()      # host-side lengths (a DataLoader delivers them on the host)
    ql = L.to_device_async(ql_cpu, dev)
    Lmax = int(ql_cpu.max())
    S = Lmax * n_frames
    H = lstm.hidden_size
    # input projection is identical at every repeat: once per token, both biases folded in
    xg = F.linear(emb, lstm.weight_ih_l0, lstm.bias_ih_l0 + lstm.bias_hh_l0)
    out, hn, cn = ops.lstm_seq(xg, lstm.weight_hh_l0, h0, c0, ql.to(torch.int32), n_frames, S, wgrad_dtype)   # [B,S,H]
    rep = torch.arange(n_frames, device=dev).unsqueeze(0)                                   # [1,F]
    last_idx = rep * ql.unsqueeze(1) + ql.unsqueeze(1) - 1                                  # [B,F]
    h_last = out.gather(1, last_idx.unsqueeze(2).expand(B, n_frames, H))
    states = None
    if want_states:
        w = torch.arange(Lmax, device=dev).view(1, 1, Lmax)
        idx = rep.unsqueeze(2) * ql.view(B, 1, 1) + w                                       # [B,F,Lmax]
        valid = (w < ql.view(B, 1, 1)).expand(B, n_frames, Lmax)
        idx = torch.where(valid, idx, torch.zeros_like(idx))
        states = out.gather(1, idx.reshape(B, -1, 1).expand(B, n_frames * Lmax, H)).view(B, n_frames, Lmax, H)
        states = states * valid.unsqueeze(3).to(states.dtype)
    return h_last, states, (hn, cn)


class FiLMTrunkBase(nn.Module):
    """Common constructor pieces / conv trunk of the three models.  Parameter containers are
    stock torch.nn modules so that names, shapes, init and state_dict match the reference
    (GPU flavour: film_layer registered, conv1x1_layers a plain list — SURVEY §0.5/0.6);
    their forward() is never called: compute goes through videonavqa_amd.ops."""

    def _build_trunk_head(self, num_input_channels, num_res_block_channels):
        """relu / conv_init / bn_init — registered first, as upstream (film_attn_pt_stem.py:39-42)."""
        self.relu = nn.ReLU(inplace=True)
        self.conv_init = nn.Conv2d(num_input_channels, num_res_block_channels, kernel_size=3, padding=1)
        self.bn_init = nn.BatchNorm2d(num_res_block_channels)
        self.conv1x1_layers = []  # plain list on purpose (film_attn_pt_stem.py:44)

    def _build_film_pipeline(self, num_res_block_channels, num_res_blocks):
        """film_pipeline (get_film_pipeline, film_attn_pt_stem.py:93-108).  Called AFTER the FiLM generator modules are
        registered: upstream creates film_layer (time_multi_hop: q_encoder .. decoder_norm) before film_pipeline, and
        torch.optim.Adam's state_dict indexes its state by position in model.parameters() — the registration order is
        part of the checkpoint contract (eval/q_and_v_eval.py:148-156,344-345)."""
        layers = []
        in_channels = num_res_block_channels
        for _ in range(num_res_blocks):
            layers.append(nn.Conv2d(in_channels, num_res_block_channels, kernel_size=3, padding=1))
            c1 = nn.Conv2d(in_channels, num_res_block_channels, kernel_size=1)
            for p in c1.parameters():
                p.requires_grad_(False)   # never optimised upstream (not in parameters()); frozen here
            self.conv1x1_layers.append(c1)
        self.film_pipeline = nn.ModuleList(layers)
        self.num_res_blocks = num_res_blocks
        self.num_res_block_channels = num_res_block_channels

    def weights_init(self, m):
        reference_init_(m)

    # nn.Module._apply does not see the plain list: keep the frozen 1x1 convs on the model's device
    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        for c in self.conv1x1_layers:
            c._apply(fn)
        carried = self.__dict__.get("_carried")
        if carried is not None:          # the lazily kept state follows the module to its new device / dtype
            self.__dict__["_carried"] = (fn(carried[0]), fn(carried[1]), carried[2])
        elif self.__dict__.get("_film_hidden") is not None:
            self.__dict__["_film_hidden"] = tuple(fn(t) for t in self.__dict__["_film_hidden"])
        self.__dict__["_zero_state"] = None
        return self

    def extra_state_tensors(self):
        """Tensors the reference silently leaves out of state_dict() (conv1x1_layers)."""
        out = {}
        for i, c in enumerate(self.conv1x1_layers):
            out["conv1x1_layers.%d.weight" % i] = c.weight
            out["conv1x1_layers.%d.bias" % i] = c.bias
        return out

    def load_reference_tensors(self, tensors):
        """Load a {name: tensor} dict that may also carry conv1x1_layers.* (goldens, rank-0 broadcast)."""
        own = dict(self.state_dict())
        own.update(self.extra_state_tensors())
        with torch.no_grad():
            for k, v in tensors.items():
                if k in own:
                    own[k].copy_(torch.as_tensor(v).to(own[k].device).view_as(own[k]))

    # ---- input handling -------------------------------------------------------------------
    def _prepare_input(self, v_input, v_lens):
        cdt = self.compute_dtype
        if isinstance(v_input, NativeFeatures):
            lay = v_input.layout
            assert v_input.data.dtype == cdt
            # (precision 'fp16h': the stem's features may be a SPLIT tensor [hi | lo | hi] — three times the channels; conv_init recognises it)
            self.__dict__["_in_shift"] = getattr(v_input, "shift", None)      # mean-shifted features: conv_init's bias absorbs the mean
            self.__dict__["_in_twin"] = getattr(v_input, "segs", 1) == 2
            return v_input.data, lay, v_input.h, v_input.w
        self.__dict__["_in_shift"] = None
        self.__dict__["_in_twin"] = False
        assert v_input.is_cuda, "the HIP path needs device tensors (no CPU fallback)"
        B, C, h, w, T = v_input.shape
        lay = FrameLayout(v_lens, T, v_input.device)
        x = K.feat_to_nhwc(v_input, lay.img_of, lay.n_img, cdt)
        return x, lay, h, w

    # The carried question-LSTM state is kept lazily: the reference stores it in q_len-sorted order (:150,:160), which costs
    # a host sort + upload + two gathers per step — paid only when somebody actually reads `film_hidden` (a test, or a
    # caller that does NOT reset it with init_hidden() between minibatches).
    def _get_film_hidden(self):
        carried = self.__dict__.get("_carried")
        if carried is not None:
            hn, cn, ql_cpu = carried
            perm = L.to_device_async(torch.sort(ql_cpu, dim=0, descending=True, stable=True)[1], hn.device)
            self.__dict__["_film_hidden"] = (hn[perm].unsqueeze(0), cn[perm].unsqueeze(0))
            self.__dict__["_carried"] = None
        return self.__dict__.get("_film_hidden")

    def _set_film_hidden(self, value):
        self.__dict__["_film_hidden"] = value
        self.__dict__["_carried"] = None

    film_hidden = property(_get_film_hidden, _set_film_hidden)

    def _zero_hidden(self, B, H, device):
        """Cached zero state (never written in place): init_hidden() every minibatch costs no fill kernels."""
        z = self.__dict__.get("_zero_state")
        if z is None or z[0].shape != (1, B, H) or z[0].device != torch.device(device):
            z = (torch.zeros(1, B, H, device=device), torch.zeros(1, B, H, device=device))
            self.__dict__["_zero_state"] = z
        return z

    def _question_state(self, B, H, q_lens, device):
        """Per-sample initial (h, c); the reference stores it in q_len-sorted order (:150,:160)."""
        carried = self.__dict__.get("_carried")
        if carried is not None:
            # State of the previous forward, kept in SAMPLE order.  Upstream stores it in q_len-sorted order and hands sorted
            # slot i of the old batch to sorted slot i of the new one (:150,:160): identical to "use as is" when the two
            # batches sort the same way (always, when the caller resets with init_hidden() or repeats a batch); otherwise
            # sample perm_new[i] inherits the state of sample perm_old[i]
            hn, cn, ql_old = carried
            ql_new = q_lens.detach().cpu().long()
            if ql_old.shape == ql_new.shape and not torch.equal(ql_old, ql_new):
                perm_old = torch.sort(ql_old, dim=0, descending=True, stable=True)[1]
                perm_new = torch.sort(ql_new, dim=0, descending=True, stable=True)[1]
                if not torch.equal(perm_old, perm_new):
                    src = torch.empty_like(perm_old)
                    src[perm_new] = perm_old
                    src = L.to_device_async(src, hn.device)
                    return hn.index_select(0, src), cn.index_select(0, src)
            return hn, cn
        fh = self.__dict__.get("_film_hidden")
        z = self.__dict__.get("_zero_state")
        if fh is None or (z is not None and fh[0] is z[0]):
            z = self._zero_hidden(B, H, device)
            return z[0][0], z[1][0]
        perm = L.to_device_async(torch.sort(q_lens.cpu(), dim=0, descending=True, stable=True)[1], device)
        h0 = torch.empty(B, H, device=device)
        c0 = torch.empty(B, H, device=device)
        h0[perm] = fh[0][0].to(device)
        c0[perm] = fh[1][0].to(device)
        return h0, c0

    def _store_question_state(self, hn, cn, q_lens):
        self.__dict__["_carried"] = (hn.detach(), cn.detach(), q_lens.detach().cpu().long())

    def question_film_values(self, lstm, proj, q_input, q_lens, lay, padding_idx=None):
        """FiLM generator for all frames of the minibatch (film_attn_pt_stem.py:144-181 called once per frame from :213):
        embedding + input projection (one HIP launch), the question LSTM re-run per frame with carried state as ONE
        persistent chain, then Linear + ReLU on the state at the last token of every repeat, written per packed image.
        Returns film [n_img, 2*C*blocks] fp32."""
        B = q_input.shape[0]
        H = lstm.hidden_size
        dev = q_input.device
        ql_cpu = q_lens.detach().cpu().long()
        Lmax = int(ql_cpu.max())
        S = Lmax * lay.n_frames
        tbl = np.concatenate([ql_cpu.numpy().astype(np.int32), lay.last_token_rows(ql_cpu, S)])
        tbl = L.to_device_async(torch.from_numpy(tbl), dev)          # ONE pinned upload for both tables
        ql_i32, rows = tbl[:B], tbl[B:]
        xg = ops.embed_proj(q_input, self.embed.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, padding_idx)
        h0, c0 = self._question_state(B, H, q_lens, dev)
        hs, hn, cn = ops.lstm_seq(xg, lstm.weight_hh_l0, h0, c0, ql_i32, lay.n_frames, S, self._lstm_wgrad_dtype())
        self._store_question_state(hn, cn, q_lens)
        return ops.linear(hs.view(B * S, H), proj.weight, proj.bias, relu=True, rows=rows)

    def bow_film_values(self, enc, proj, q_input, lay):
        """FiLM generator with q_encoder='bow' (film_attn_pt_stem.py:75-77,171-181 / film_global_pooling_pt_stem.py:164-174):
        nn.Linear over every position of the padded question, summed over positions (padding tokens included — upstream's
        division by the question length at :175-176 discards its result), then Linear + ReLU.  No carried state, the same
        values for every frame; both products on the fp32 HIP GEMM.  Returns film [n_img, 2*C*blocks] fp32."""
        B, Lq = q_input.shape
        emb = self.embed(q_input)
        pooled = ops.linear(emb.reshape(B * Lq, -1), enc.weight, enc.bias).view(B, Lq, -1).sum(1)
        film = ops.linear(pooled, proj.weight, proj.bias, relu=True)
        return film.index_select(0, lay.sample_of)

    # ---- conv trunk on the packed image list -------------------------------------------------
    def _trunk_fused(self, x, lay, film_specs, join=None):
        """TRAIN-mode trunk on the fused conv epilogues as two autograd nodes (ops.FilmTrunkHeadFn: conv_init + BatchNorm, which
        needs nothing from the question; ops.FilmTrunkBlocksFn: the FiLM residual blocks).  film_specs[k] = (FiLM matrix
        [n_img, ld] fp32, column of block k's gamma) — beta follows at + C.  `join` (from _fork_generator) is called between the
        two nodes: the FiLM matrices may still be in flight on the generator's side stream until then."""
        if not self.training:
            return self._trunk_infer(x, lay, film_specs, join)
        C = self.num_res_block_channels
        meta = ops.TrunkMeta(lay, C, self.num_res_blocks, 0, [], BN_EPS, grad_scale=getattr(self, "_trunk_grad_scale", 1.0))
        meta.in_shift = self.__dict__.get("_in_shift")
        meta.in_twin = bool(self.__dict__.get("_in_twin", False))
        meta.hybrid = bool(self.__dict__.get("hyb", False))
        bn = self.bn_init
        S = (x.shape[1] - 2) * (x.shape[2] - 2)
        h, mean, var = ops.FilmTrunkHeadFn.apply(x, self.conv_init.weight, self.conv_init.bias, bn.weight, bn.bias, meta)
        self._advance_running_stats(bn, lay, mean, var, S)
        if join is not None:
            join()
        uniq, film_map = [], []
        for t, col in film_specs:
            for i, u in enumerate(uniq):
                if u is t:
                    break
            else:
                i = len(uniq)
                uniq.append(t)
            film_map.append((i, int(col)))
        uniq = [u if (u.dtype == torch.float32 and u.stride(1) == 1) else u.float().contiguous() for u in uniq]
        meta.n_film, meta.film_map = len(uniq), film_map
        blocks = []
        for k in range(self.num_res_blocks):
            c1, c3 = self.conv1x1_layers[k], self.film_pipeline[k]
            blocks += [c1.weight, c1.bias, c3.weight, c3.bias]
        meta.c1_packs = self._frozen_c1_packs(h.dtype, L.round_up(C, 64))
        if self.__dict__.get("hyb", False):
            meta.hybrid = True
            packs32 = self._frozen_c1_packs(torch.float32, L.round_up(C, 64))
            meta.c1_packs32 = [p[0] for p in packs32] if packs32 else None
        return ops.film_trunk_blocks(h, meta, uniq, blocks)

    def _frozen_c1_packs(self, cdt, c_pad):
        """K-major packs (forward, flipped for dgrad) of the frozen 1x1 conv weights, re-made only when a weight was modified
        in place (tensor._version: load_reference_tensors, checkpoint restore, .to()).  Trainable weights are never cached —
        the fused Adam kernel updates them through the flat buffer without touching their version counter."""
        if any(c.weight.requires_grad for c in self.conv1x1_layers):
            return None
        key = (cdt, c_pad, tuple((c.weight._version, c.weight.data_ptr()) for c in self.conv1x1_layers))
        cache = self.__dict__.setdefault("_c1_pack_cache", {})
        cached = cache.get(cdt)
        if cached is None or cached[0] != key:
            packs = [(K.pack_conv_weight(c.weight, cdt, c_out_pad=c_pad, c_in_pad=c_pad),
                      K.pack_conv_weight(c.weight, cdt, transpose_flip=True, c_out_pad=c_pad, c_in_pad=c_pad))
                     for c in self.conv1x1_layers]
            cached = cache[cdt] = (key, packs)
        return cached[1]

    @staticmethod
    def _advance_running_stats(bn, lay, mean, var, S):
        """Running statistics advanced once per processed frame in frame order (film_attn_pt_stem.py:211 calls bn_init once
        per frame: momentum 0.1, unbiased variance) — one HIP launch."""
        with torch.no_grad():
            K.bn_running_update(mean, var, lay.frame_off_i32, lay.n_frames, S, bn.running_mean, bn.running_var, BN_MOMENTUM)
            bn.num_batches_tracked += lay.n_frames

    def _lstm_wgrad_dtype(self):
        """Operand type of the LSTMs' dW_hh GEMM: the compute dtype, except fp16 (gate gradients underflow it: exact f32)."""
        return torch.float32 if self.compute_dtype == torch.float16 else self.compute_dtype

    def _use_fused_trunk(self):
        """The fused conv-trunk path: train mode (two autograd nodes on the fused epilogues), and — round 4 — eval mode under
        torch.no_grad() (val_epoch / test, eval/q_and_v_eval.py:159-224, eval/q_and_v_test.py:64-142): the forward-only
        form of the same kernels (_trunk_infer).  Eval mode WITH autograd enabled keeps the op-by-op graph."""
        import os
        if self.training:
            return os.environ.get("VNQA_FUSED_TRUNK", "1") != "0"
        return self._use_fused_eval()

    def _use_fused_eval(self):
        import os
        return (not self.training) and (not torch.is_grad_enabled()) and os.environ.get("VNQA_FUSED_EVAL", "1") != "0"

    def _trunk_infer(self, x, lay, film_specs, join=None):
        """Forward-only conv trunk for inference (model.eval() under no_grad): three launches per FiLM block-model instead of
        the op-by-op graph's seven —
          conv_init -> ReLU -> eval-mode BatchNorm (film_attn_pt_stem.py:211 with running statistics): ONE conv launch, the
            BatchNorm folded into the epilogue's per-channel affine (scale = gamma * rsqrt(running_var + eps), shift = beta -
            running_mean * scale, applied to the fp32 accumulator after the ReLU, rounded once);
          per block: the frozen 1x1 conv + ReLU, then conv3x3 -> FiLM -> ReLU -> + residual as one launch WITHOUT the z
            output (VNQA_EPI_FILM_RES, y = NULL: only the backward reads z).
        No autograd nodes, no statistics, no saved activations."""
        C = self.num_res_block_channels
        cdt = x.dtype
        c_pad = L.round_up(C, 64)
        bn = self.bn_init
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + BN_EPS)
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        hyb = self.__dict__.get("hyb", False) and L.is_half(cdt)
        twin = bool(self.__dict__.get("_in_twin", False))                        # mean-shifted features written twice: two products, split weights
        split = ops.is_split(x, self.conv_init.weight.shape[1]) and not twin     # [hi | lo | hi] features, three products
        wt0 = K.pack_conv_weight(self.conv_init.weight, torch.float32 if (split or twin) else cdt, c_out_pad=c_pad,
                                 c_in_pad=x.shape[-1] // (3 if split else (2 if twin else 1)))
        if twin:
            wt0 = K.split_weight2(wt0)
        ps = (split or twin) and K.conv_ps_supported(x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[-1], c_pad)
        b0 = K.pad_vec(self.conv_init.bias, c_pad)
        in_shift = self.__dict__.get("_in_shift")
        if in_shift is not None and not split:      # mean-shifted features: W * (x' + mu) = W * x' + sum_taps(W) mu, in fp32 with the exact weights
            b0 = b0 + ops.shift_bias_correction(self.conv_init.weight.detach(), in_shift, c_pad)
        h = K.conv2d_igemm(x, wt0, bias=b0, relu=True, post_scale=K.pad_vec(scale, c_pad),
                           post_shift=K.pad_vec(shift, c_pad), split_in=split, tile=L.TILE_PS_224x256 if ps else L.TILE_AUTO)
        if join is not None:
            join()
        packs = self._frozen_c1_packs(torch.float32 if hyb else cdt, c_pad)
        for k in range(self.num_res_blocks):
            c1, c3 = self.conv1x1_layers[k], self.film_pipeline[k]
            if hyb:          # the frozen 1x1 conv against split weights (two products), as the training graph runs it
                res = K.conv2d_igemm(h, packs[k][0] if packs else K.pack_conv_weight(c1.weight, torch.float32, c_out_pad=c_pad, c_in_pad=c_pad),
                                     bias=K.pad_vec(c1.bias, c_pad), relu=True, split_weights=True)
            else:
                wt1 = packs[k][0] if packs else K.pack_conv_weight(c1.weight, cdt, c_out_pad=c_pad, c_in_pad=c_pad)
                res = K.conv2d_igemm(h, wt1, bias=K.pad_vec(c1.bias, c_pad), relu=True)
            film, col = film_specs[k]
            if not (film.dtype == torch.float32 and film.stride(1) == 1):
                film = film.float().contiguous()
            _, h = K.conv2d_igemm_film_res(res, K.pack_conv_weight(c3.weight, cdt, c_out_pad=c_pad, c_in_pad=c_pad),
                                           K.pad_vec(c3.bias, c_pad), film[:, col:col + C], film[:, col + C:col + 2 * C], C, res,
                                           tile=K.ps_fused_tile(res), keep_z=False)
        return h

    def _trunk(self, x, lay, film_fn):
        """conv_init -> ReLU -> per-frame BN -> FiLM residual blocks.
        film_fn(k) -> (gamma [n_img,C], beta [n_img,C]) for block k."""
        return self._trunk_blocks(self._trunk_head(x, lay), lay, film_fn)

    def _trunk_head(self, x, lay):
        """conv_init -> ReLU -> per-frame BN (film_attn_pt_stem.py:211): the part that does not need the question."""
        # in train mode the BN backward applies conv_init's ReLU mask itself (fused)
        in_shift = self.__dict__.get("_in_shift")
        if in_shift is not None and (self.__dict__.get("_in_twin", False) or not ops.is_split(x, self.conv_init.weight.shape[1])):
            c_in_pad = L.round_up(self.conv_init.weight.shape[1], 64)
            x = ops.unshift_features(x[..., :c_in_pad].contiguous(), in_shift)      # the op-by-op graph reads plain features (an API path)
        r = ops.conv(x, self.conv_init.weight, self.conv_init.bias, relu=True, mask_in_backward=not self.training)
        return frame_batchnorm(r, self.bn_init, lay, self.training, self.compute_dtype)

    def _trunk_blocks(self, x, lay, film_fn):
        cdt = self.compute_dtype
        for k in range(self.num_res_blocks):
            c1 = self.conv1x1_layers[k]
            # (precision 'fp16h': split weights on the frozen 1x1 conv, as the fused graphs run it)
            res = ops.conv(x, c1.weight, c1.bias, relu=True, split_weights=self.__dict__.get("hyb", False) and L.is_half(x.dtype))
            z = ops.conv(res, self.film_pipeline[k].weight, self.film_pipeline[k].bias, relu=False)
            gamma, beta = film_fn(k)
            x = film_relu_residual(z, res, gamma, beta, cdt)
        return x

    # ---- FiLM generator on a side stream ---------------------------------------------------------
    # The question LSTM chain (~800 dependent cells, 0.8 ms forward / 0.95 ms BPTT on 8 CUs) needs nothing from the video: forked
    # onto its own high-priority stream it runs beside conv_init / BN in forward, and — autograd replays every backward op on
    # its forward's stream — its BPTT runs beside BN-backward / conv_init's wgrad in backward.  Default ON since round 3
    # (VNQA_SIDE_LSTM=0: same stream): in the pipelined step the stem of minibatch i+2 may not start before the trunk of
    # minibatch i has released its feature slot, so the trunk's dependent-chain LATENCY under contention — not only the summed
    # kernel work — sets the step; same-box A/B +6.5 % (877-897 -> 940-956 clips/s) even on the un-fused graph.
    def _fork_generator(self, fn, n_img=None):
        """Run fn() (returns a tensor or tuple of tensors) on the side stream; returns (result, join) where join()
        makes the current stream wait for it and registers the cross-stream use with the caching allocator.
        VNQA_SIDE_LSTM: 1 always, 0 never, default `auto` = for minibatches of up to 420 frames (n_img; without it: up to 16
        clips).  Same-box A/B, on / off: 8 x 35 frames +2.9 .. +6.5 %, 16 x 35 -0.5 %, 8 x 70 (multi-hop) -0.9 %, 32 x 35 -5.5 % —
        with more frames the trunk's kernels are longer, its chain latency no longer sets the step, and the chain's persistent
        high-priority workgroups only take CUs from the convs."""
        import os
        mode = os.environ.get("VNQA_SIDE_LSTM", "auto")
        small = (n_img <= 420) if n_img is not None else getattr(self, "batch_size", 8) <= 16
        on = mode == "1" or (mode != "0" and small)
        # (under no_grad only the fused inference path forks: the op-by-op eval graph keeps everything on one stream)
        if not torch.cuda.is_available() or not on or not (torch.is_grad_enabled() or self._use_fused_eval()):
            return fn(), (lambda: None)
        main = torch.cuda.current_stream()
        side = getattr(self, "_gen_stream", None)
        if side is None:
            side = self._gen_stream = torch.cuda.Stream(priority=-1)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn()

        def join():
            main.wait_stream(side)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if torch.is_tensor(t):
                    t.record_stream(main)
        return out, join

    def _gp_tail(self, x, lay, h, w):
        """relu(c1x1_tail) -> zero-padded stack over frames -> max over frames -> out_linear
        (film_global_pooling_pt_stem.py:228-238)."""
        gs = getattr(self, "_trunk_grad_scale", 1.0)      # fp16 storage: the tail conv and the trunk see scaled gradients
        # (precision 'fp16h': split weights — the tail conv's input is positive, so its weight rounding is a per-channel OFFSET of every
        # pooled feature, which the max over frames hands straight to out_linear)
        t = ops.conv(x, self.c1x1_tail.weight, self.c1x1_tail.bias, relu=True, grad_scale=gs,
                     split_weights=self.__dict__.get("hyb", False) and L.is_half(x.dtype))     # [n_img,hp,wp,tail_pad]
        tail = self.c1x1_tail.out_channels
        # HIP tail: segmented max over each sample's frames straight from the packed image list, written in the reference's
        # NCHW-flattened order, then out_linear on the fp32 GEMM (no dense [T, B, ...] stack, no weight re-layout).
        # (The dense torch form of this tail is test infrastructure: tests/torch_partners.py.)
        pooled, self._gp_argmax = ops.frame_max(t, lay, tail, gs, getattr(self, "_gp_route", None))
        return ops.linear(pooled, self.out_linear.weight, self.out_linear.bias)
