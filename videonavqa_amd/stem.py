"""Frozen per-frame conv stem on the HIP igemm: VGG-16 features[0:10] + ObjDetectCNN.

Reference: eval/q_and_v_eval.py:102-110 runs, per frame and under no_grad,
`feature_extractor(v[:, :, :, :, j])` (external faster-rcnn.pytorch VGG front; shape-derived
definition, SURVEY §0.1 — parity unpinned) then `obj_detector(features)`
(models/obj_detector.py:69-86, eval mode).  Here all valid frames of the minibatch go through
each layer in ONE launch:

  conv1_1 (+ReLU)            conv_first kernel, reads the [B,3,H,W,T] clip directly
  conv1_2 +ReLU +pool        igemm, fused epilogue
  conv2_1 +ReLU
  conv2_2 +ReLU +pool  +bn_input affine (applied by the PRODUCER so conv11's zero padding stays exact)
  conv11 (bias only)         (no ReLU between conv11 and conv12, obj_detector.py:72)
  conv12 (bn1 folded) +ReLU +pool
  conv21 ; conv22 (bn2 folded) +ReLU +pool
  conv31 ; conv32 (bn3 folded) +ReLU          -> [n_img, h+2, w+2, 512] padded NHWC

Eval-mode BatchNorm after a conv folds exactly into that conv's weights and bias.
Activation buffers are allocated once per geometry (zero halo written once, never touched again).
"""
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import kernels as K

BN_EPS = 1e-5


class VGGFront(nn.Module):
    """VGG-16 'D' features[0:10] parameter container + drop-in callable
    (`feature_extractor(x)` of eval/q_and_v_eval.py:106): [N,3,H,W] -> [N,128,H/4,W/4]."""

    def __init__(self, precision='bf16'):
        super(VGGFront, self).__init__()
        self.features = nn.ModuleDict({
            "0": nn.Conv2d(3, 64, 3, padding=1), "2": nn.Conv2d(64, 64, 3, padding=1),
            "5": nn.Conv2d(64, 128, 3, padding=1), "7": nn.Conv2d(128, 128, 3, padding=1)})
        self.precision = precision
        self._plan = None
        for p in self.parameters():
            p.requires_grad_(False)

    def load_state_dict(self, *a, **k):
        out = super(VGGFront, self).load_state_dict(*a, **k)
        self._plan = None
        return out

    def forward(self, x):
        if self._plan is None:      # (the per-module drop-in path of fp16x runs on the exact-f32 kernels: an API path, not the fast one)
            self._plan = FrozenStem(self, None, {"fp16x": "fp32", "fp16w": "fp16", "fp16h": "fp16"}.get(self.precision, self.precision))
        return self._plan.vgg_nchw(x)


def get_frcnn_feature_extractor(path=None, precision='bf16'):
    """Counterpart of `demo.get_frcnn_feature_extractor(path)` (eval/q_and_v_eval.py:17,308).
    Loads `features.{0,2,5,7}.*` from a VGG-16 state dict (e.g. vgg16_caffe.pth) when given."""
    m = VGGFront(precision)
    if path is not None:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        m.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.") and
                           k.split(".")[1] in ("0", "2", "5", "7")})
    return m.eval()


def coherent_round(w, m, dtype):
    """Weights w [c_out, c_in, kh, kw] (fp32) rounded to `dtype` (a 16-bit format) so that the rounding errors of each OUTPUT channel
    cancel against the mean input:  sum_{c_in, taps} m[c_in] * (w16 - w) ~ 0  instead of a random walk of c_in * taps half-ulps.
    Round-to-nearest everywhere, then per output channel the few weights nearest to a rounding midpoint (cheapest to move) whose
    flip to the OTHER neighbour moves the weighted residual towards zero are flipped, as many as minimise |residual|.
    The coherent part of a frozen layer's weight-rounding error — a constant offset per output channel wherever the input has its
    usual mean, which no later spatial / temporal averaging removes — goes; the incoherent part (zero-mean over pixels) grows by the
    few flipped weights only.  Returns fp32 values that are exactly representable in `dtype`."""
    co = w.shape[0]
    wf = w.detach().float().reshape(co, w.shape[1], -1)
    r = wf.to(dtype)
    bits = r.view(torch.int16).to(torch.int32)
    rf = r.float()
    up = rf < wf                                   # the other neighbour lies above (towards +inf) / below the nearest one
    mag_up = (rf > 0) | ((rf == 0) & up)           # moving towards +inf grows the magnitude of a positive value (bits + 1) ...
    step = torch.where(up == mag_up, torch.ones_like(bits), -torch.ones_like(bits))
    ob = bits + step
    # (crossing zero: the neighbour of +-0 is the smallest subnormal of the other sign's direction)
    zero = rf == 0
    ob = torch.where(zero, torch.where(up, torch.ones_like(bits), torch.full_like(bits, -32767)), ob)      # 0x0001 / 0x8001
    other = ob.to(torch.int16).view(dtype).float()
    exact = rf == wf
    d0, d1 = rf - wf, other - wf
    mm = m.detach().float().view(1, -1, 1).to(wf.device).expand_as(wf)
    R = (mm * d0).sum((1, 2))                                            # [co] weighted residual of round-to-nearest
    delta = (mm * (d1 - d0)).reshape(co, -1)                             # change of R when element j flips
    cost = (d1.abs() - d0.abs()).reshape(co, -1)
    ok = (delta * R.view(-1, 1) < 0) & ~exact.reshape(co, -1) & torch.isfinite(other).reshape(co, -1)
    cost = torch.where(ok, cost, torch.full_like(cost, float("inf")))
    order = cost.argsort(dim=1)
    dsort = torch.gather(torch.where(ok, delta, torch.zeros_like(delta)), 1, order)
    csum = torch.cumsum(dsort, 1)
    resid = torch.cat([R.view(-1, 1), R.view(-1, 1) + csum], 1).abs()    # residual after flipping the k cheapest candidates
    k = resid.argmin(1)                                                  # [co]
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(order.shape[1], device=order.device).view(1, -1).expand_as(order))
    flip = (rank < k.view(-1, 1)) & ok
    out = torch.where(flip.view_as(rf), other, rf)
    return out.view_as(w)


@torch.no_grad()
def calibration_means(vgg, od, frames=None, n_frames=4, height=224, width=224, seed=4242):
    """Per-input-channel mean activation at every stem layer's input — what coherent_round cancels against — from ONE pass of
    calibration frames [N, 3, H, W] (values in [0, 1]) through the library's own exact-f32 stem (a FrozenStem of precision 'fp32'
    with its layer outputs tapped).  Default frames: seeded uniform noise, the benchmark's kind of data; a deployment passes
    frames of its own videos.  Keys: first = conv1_1's input, vgg0 / vgg1 / vgg2 = conv1_2 / conv2_1 / conv2_2, od0 = conv11 (and
    the composed 5x5 pair), od1 = conv12, od2 .. od5 = conv21 .. conv32."""
    dev = vgg.features["0"].weight.device
    if frames is None:
        frames = torch.rand(n_frames, 3, height, width, generator=torch.Generator().manual_seed(seed))
    frames = frames.float().to(dev)
    N, _, H, W = frames.shape
    ref = FrozenStem(vgg, od, "fp32")
    ref._tap = {}
    clip = frames.permute(1, 2, 3, 0).unsqueeze(0).contiguous()                  # [1, 3, H, W, T]
    ref.forward_clip(clip, torch.arange(N, dtype=torch.int32, device=dev), N)

    def mean(key, c):
        t = ref._tap[key]                                                        # padded NHWC fp32, zero halo
        halo = 2 if key == ("vgg", 2) and ref.composed is not None else 1
        n, hp, wp, _ = t.shape
        return (t.double().sum((0, 1, 2)) / (n * (hp - 2 * halo) * (wp - 2 * halo)))[:c].float().cpu()
    m = {"first": frames.double().mean((0, 2, 3)).float().cpu(),
         "vgg0": mean("first", 64), "vgg1": mean(("vgg", 0), 64), "vgg2": mean(("vgg", 1), 128), "od0": mean(("vgg", 2), 128)}
    # conv12's input is conv11's output, a linear function of od0 (no nonlinearity inside a pair): its mean follows analytically
    w11, b11 = od.conv11.weight.detach().double().cpu(), od.conv11.bias.detach().double().cpu()
    m["od1"] = (w11.sum((2, 3)) @ m["od0"].double() + b11).float()
    c = od.conv12.out_channels
    m["od2"] = mean(("od", "c") if ref.composed is not None else ("od", 1), c)
    m["od3"], m["od4"], m["od5"] = mean(("od", 2), od.conv21.out_channels), mean(("od", 3), c), mean(("od", 4), od.conv31.out_channels)
    return m


def _fold_bn(bn):
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + BN_EPS)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


class FrozenStem(object):
    """Execution plan (packed weights + persistent activation buffers) for the frozen stem."""

    def __init__(self, vgg, objdet, precision='bf16', out_half=False, calibration="auto", pair_features=True):
        from .models.common import compute_dtype
        self.cdt = compute_dtype(precision)
        # precision 'fp16h' (round 5, the tolerance mode): the fp16 precision's stem — same kernels, coherently rounded weights — except
        # that the LAST THREE stored activations (conv22's pooled output, conv31's, conv32's = the features) are [hi | lo] PAIRS
        # (hi = fp16(v), lo = fp16(v - hi), written by the patch-stationary kernel's dual epilogue) and conv31 / conv32 contract
        # both halves against split weights (a plain conv over 3 C input channels, [hi | lo | hi] . [w_hi | w_hi | w_lo]): the three
        # stem activation roundings and the two weight roundings that weigh most in the logits error (profiles/
        # r05_precision_budget*.txt: 0.047 / 0.047 / 0.073 and 0.022 / 0.041 of the fp16 precision's 0.70e-6 squared error) are gone for
        # two extra products on the two CHEAPEST layers (14 x 14 maps).  pair_features=False keeps the
        # features a plain fp16 tensor (consumers that do not read pairs: MACNetwork, the per-module drop-in path).
        self.hyb = precision == "fp16h"
        self.pair_features = bool(pair_features) and self.hyb
        self.x3 = precision == "fp16x"       # fp32 storage, contractions as three fp16-half products (kernels.f32_conv_mode)
        # fp16x with a 16-bit-storage trunk behind it (precision 'fp16' / 'fp16w' models): the LAST layer's output rounded once to
        # fp16 instead of written as fp32
        self.out_half = bool(out_half) and self.x3
        self.w2 = precision == "fp16w"       # fp16 storage, every layer after the fused conv1 with split weights (two products)
        self.vgg, self.objdet = vgg, objdet
        self.layers_vgg, self.layers_od = [], []
        self.composed = None
        self.first = None
        self._bufs = {}
        self._pair_ok = {}
        # CUs the persistent one-workgroup-per-CU kernels (fused conv1, C_in = 64 direct conv, weights-in-registers conv) leave to
        # other streams, passed with every call (vnqa_conv_desc.flags): the Trainer sets it to its stem stream's CU reservation;
        # VNQA_PERSISTENT_RESERVE_CUS is the stand-alone A/B knob (8: -12 % on one GPU; multi-GPU investigation)
        self.reserve_cus = int(os.environ.get("VNQA_PERSISTENT_RESERVE_CUS", "0"))
        self.timing = None   # bench hook: list collecting (start event, end event, FLOPs, kernel) of the C_out = 512 stem launches
        # calibration: how the frozen 16-bit weights are rounded.  None = round-to-nearest; a dict = the means of an earlier
        # calibration_means() pass (a checkpoint's `extra_state['_stem_calibration']`: the test-time stem gets the weights the model
        # was trained behind); "noise" or a tensor of frames
        # [N, 3, H, W] = coherent_round against the mean input activations measured on those frames (calibration_means): each output
        # channel's rounding errors cancel against the mean input, the part of the weight-rounding error that is a constant offset
        # per channel and survives every later pooling.  Same kernels, same bytes; measured at the headline size on 12 minibatches
        # (profiles/r04_x3_error_budget_*.txt): whole-fp16 stem + x3 trunk 1.06e-3 -> 0.62e-3 rms logits error, the fp16 precision
        # 1.21e-3 -> 0.88e-3, bf16 9.0e-3 -> 7.1e-3 on the parity batches.  "auto" = "noise" for every 16-bit precision (the pass
        # costs one exact-f32 stem plan and 4 frames at construction); VNQA_COHERENT_ROUND=0 turns it off (round-to-nearest).
        self.calib = None
        self._tap = None             # calibration hook: {layer key: output tensor} filled by _run / _run_composed
        if isinstance(calibration, str) and calibration == "auto":
            env = os.environ.get("VNQA_COHERENT_ROUND")
            calibration = "noise" if env != "0" else None
        if isinstance(calibration, dict):      # calibration means computed earlier (a checkpoint's: eval/q_and_v_test.py)
            self.calib = {k: torch.as_tensor(v).float().cpu() for k, v in calibration.items()} if precision != "fp32" else None
        elif calibration is not None and vgg is not None and objdet is not None and precision != "fp32" and \
                vgg.features["0"].weight.is_cuda:
            self.calib = calibration_means(vgg, objdet, None if isinstance(calibration, str) else calibration)
        cm = lambda k: None if self.calib is None else self.calib[k]
        if vgg is not None:
            f = vgg.features
            dev = f["0"].weight.device
            w0 = f["0"].weight.detach().float().contiguous()
            if self.calib is not None and (precision != "fp16x" or os.environ.get("VNQA_X3_PLAIN_FIRST", "1") != "0"):
                w0 = coherent_round(w0, cm("first"), L.half_dtype()).contiguous()
            self.first = (w0, f["0"].bias.detach().float().contiguous())
            # fp16x: conv1_1 + conv1_2 — 3.6 GB of fp32 activations each at 280 frames, HBM-bound as x3 products (8.4 of the all-x3
            # stem's 32 ms) — run as the plain fp16 fused kernel by default (1.05 ms): five fp16 roundings (clip, two weight sets,
            # two activations) stay in the forward pass, ~0.5e-3 of logits error instead of ~1e-5 (VNQA_X3_PLAIN_FIRST=0: all x3)
            self.x3_plain_first = self.x3 and os.environ.get("VNQA_X3_PLAIN_FIRST", "1") != "0"
            # VNQA_X3_ROUND=n (default 6): the INPUT of the n heaviest x3 layers (composed 5x5, conv22, conv21, conv2_2, conv31, conv32 —
            # in that order) is kept as ONE rounded fp16 tensor: two products instead of three on that layer (a third of its matrix
            # work) for one more fp16 rounding (those layers run as fused two-product launches).  Speed / tolerance curve of the
            # mode at the headline size: profiles/r04_fp16x_curve.txt
            self.x3_round = set(("composed", "od3", "od2", "vgg2", "od4", "od5")[:int(os.environ.get("VNQA_X3_ROUND", "6"))]) if self.x3 else set()
            # VNQA_X3_PLAIN_PREFIX=k (fp16x): the first k stem layers — in the order fused conv1, conv2_1, conv2_2, the composed pair,
            # conv21, conv22, conv31 — run EXACTLY as in precision 'fp16' (plain storage, ONE product, the fast kernels: weights in
            # registers, composed 5x5 with its fp16 border GEMMs, patch-stationary); the layers after the prefix are x3 products (the
            # first of them with two products: its input is the prefix's rounded fp16 output).  Default 7 — every layer but conv32:
            # with the prefix's weights rounded coherently (`calibration`, above) the whole plain prefix costs 0.6e-3 rms of logits
            # error (12 minibatches, profiles/r04_x3_error_budget_coherent.txt; 1.06e-3 with round-to-nearest weights, where the
            # default had to be 4) at a third to a half of the layers' x3 cost.
            self.x3_prefix = max(1 if self.x3_plain_first else 0, int(os.environ.get("VNQA_X3_PLAIN_PREFIX", "7"))) if self.x3 else 0
            hp = lambda i: L.half_dtype() if (i < self.x3_prefix or (self.w2 and i == 0)) else None
            self.layers_vgg = [self._layer(f["2"], relu=True, pool=True, cdt=hp(0), m=cm("vgg0")),
                               self._layer(f["5"], relu=True, pool=False, cdt=hp(1), m=cm("vgg1")),
                               self._layer(f["7"], relu=True, pool=True, cdt=hp(2), m=cm("vgg2"))]
        if objdet is not None:
            od = objdet
            self.bn_input = _fold_bn(od.bn_input)
            pre = getattr(self, "x3_prefix", 0)
            hq = lambda i: L.half_dtype() if i < pre else None       # (stem layer index: 3 = the conv11 / conv12 pair, 4 = conv21, ...)
            self.layers_od = [self._layer(od.conv11, cdt=hq(3), m=cm("od0")),
                              self._layer(od.conv12, bn=od.bn1, relu=True, pool=True, cdt=hq(3), m=cm("od1")),
                              self._layer(od.conv21, cdt=hq(4), m=cm("od2")),
                              self._layer(od.conv22, bn=od.bn2, relu=True, pool=True, cdt=hq(5), m=cm("od3")),
                              self._layer(od.conv31, cdt=hq(6), m=cm("od4")),
                              self._layer(od.conv32, bn=od.bn3, relu=True, pool=False, m=cm("od5"))]
            if self.hyb:
                # conv22 writes [hi | lo | hi], conv31 reads it and writes the same, conv32 reads it and writes the split features
                # [hi | lo | hi] for conv_init — or a plain tensor for consumers that read no split tensors.  A
                # triple-reading layer is a plain conv over 3 C input channels against the SPLIT exact weights [w_hi | w_hi | w_lo]
                # (BatchNorm folded in fp32 first): x_hi w_hi + x_lo w_hi + x_hi w_lo — neither the layer's input rounding nor its
                # weight rounding is left (coherently rounded, conv31's / conv32's weights still cost 0.022 / 0.041e-6 of squared
                # logits error: profiles/r05_precision_budget_stem_weights.txt)
                for ly, rd, wr in zip(self.layers_od[3:], (False, True, True), (3, 3, 3 if self.pair_features else 0)):
                    if "wt_ps" in ly:
                        ly["pair_out"] = wr
                        if rd:
                            ly["wt_ps3"] = K._split_weight(ly.pop("wt32ps"), "hhl")
                    ly.pop("wt32ps", None)
            # conv12 is applied straight to conv11's output (obj_detector.py:72: no nonlinearity between the two convs of
            # a pair) and both are frozen: when the pair's 3x3 (c_in -> c_mid) . 3x3 (c_mid -> c_out) costs more than one
            # 5x5 (c_in -> c_out) — 9*c_in + 9*c_mid > 25*c_in, true for 128 -> 512 -> 512 only — it is evaluated as the
            # composed 5x5 conv plus an exact correction on the image border (see _compose_pair).
            self.composed = None
            ci, cm = od.conv11.in_channels, od.conv11.out_channels
            if os.environ.get("VNQA_STEM_COMPOSE", "1") != "0" and 9 * ci + 9 * cm > 25 * ci:
                if pre >= 4:         # the pair inside the plain fp16 prefix: built exactly as precision 'fp16' builds it
                    keep = (self.cdt, self.x3)
                    self.cdt, self.x3 = L.half_dtype(), False
                    try:
                        self.composed = self._compose_pair(od.conv11, od.conv12, od.bn1)
                    finally:
                        self.cdt, self.x3 = keep
                    self.composed["cdt"] = L.half_dtype()
                else:
                    self.composed = self._compose_pair(od.conv11, od.conv12, od.bn1)
            self.out_channels = od.conv32.out_channels
            if vgg is not None:
                # bn_input becomes the post-affine of the last VGG layer's epilogue
                s, t = self.bn_input
                self.layers_vgg[-1]["post"] = (K.pad_vec(s, 128), K.pad_vec(t, 128))
                if self.composed is not None:
                    self.layers_vgg[-1]["y_halo"] = 2        # the composed 5x5 conv reads a halo-2 image
                    # (x3 conv2_2: its output also as fp32, for the exact-f32 border-correction GEMMs)
                    self.layers_vgg[-1]["dual"] = self.x3 and self.layers_vgg[-1].get("cdt") is None

    def _layer(self, conv, bn=None, relu=False, pool=False, cdt=None, m=None):
        if cdt is not None:          # a layer in another storage dtype than the stem's (fp16x: the plain fp16 first layer)
            keep, keep_x3 = self.cdt, self.x3
            self.cdt, self.x3 = cdt, False
            try:
                ly = self._layer(conv, bn, relu, pool, m=m)
            finally:
                self.cdt, self.x3 = keep, keep_x3
            ly["cdt"] = cdt
            return ly
        w = conv.weight.detach().float()
        b = conv.bias.detach().float()
        c_out, c_in = w.shape[0], w.shape[1]
        c_out_pad, c_in_pad = L.round_up(c_out, 64), L.round_up(c_in, 64)
        scale = None
        if bn is not None:
            scale, shift = _fold_bn(bn)
            b = b * scale + shift
        bf16 = L.is_half(self.cdt)      # 16-bit storage (bf16 or, in the fp16 build, fp16): the MFMA fast path
        w32, scale32 = w, scale
        if m is not None and bf16 and not self.x3:
            # (the BN scale folded first: the values the kernel multiplies with are the ones rounded)
            w = coherent_round(w if scale is None else w * scale.view(-1, 1, 1, 1), m, self.cdt)
            scale = None
        if bf16 and c_in_pad == 64:
            tile = None                      # conv_c64 direct kernel (row layout, LDS-resident weights)
        elif bf16:
            # conv2_2 (C_out = 128): the 512x128 tile (VNQA_TILE_512x128 = 15) has the 256x256 kernel's 128x64 wave tiles and
            # MFMA work per K-step; 1.45 -> 1.08 ms alone, +2 % end to end against the 256x128 tile.
            # 128x128 (2 workgroups/CU) is 20 % faster for conv2_2 ALONE but costs 10 % end to end when the trunk co-runs on
            # the other stream (same-box A/B): finer interleaving of the two streams' workgroups hurts both
            t128 = L.TILE_128x128 if os.environ.get("VNQA_STEM_T128", "0") == "1" else int(os.environ.get("VNQA_STEM_C128_TILE", "15"))
            t512 = int(os.environ.get("VNQA_STEM_C512_TILE", str(L.TILE_STEM_256x256)))     # end-to-end A/B hook
            tile = t512 if c_out_pad >= 256 else (t128 if c_out_pad > 64 else L.TILE_256x64)
            # conv11 (C_in = 128: only 18 K-steps, and a 964 MB output to store): the 16-wave shape of the same tile
            # keeps more store / DMA issue slots busy around its short main loop (+10 % on this layer, -2..4 % on the
            # 72-K-step layers, which therefore keep the 8-wave staggered kernel)
            if c_out_pad >= 256 and c_in_pad == 128 and os.environ.get("VNQA_STEM_W16", "1") != "0":
                tile = L.TILE_256x256_W16
        else:
            tile = L.TILE_128x64 if c_out_pad <= 64 else L.TILE_128x128
        if tile is None or tile in (L.TILE_256x256_W16, L.TILE_PS_224x256, L.TILE_STEM_PS_224x256) or self.x3 or \
                os.environ.get("VNQA_STEM_TILED", "1") == "0":      # (x3 products read the K-major pack)
            wt = K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        else:   # frozen weights: pre-tiled once into the exact LDS images the igemm DMA consumes
            wt = K.pack_conv_weight_tiled(w, self.cdt, tile, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        ly = dict(wt=wt, bias=K.pad_vec(b, c_out_pad), relu=relu, pool=pool, post=None,
                  c_out=c_out, c_in=c_in, c_out_pad=c_out_pad, tile=tile)
        if self.w2:      # the fp32 K-major pack: the conv wrapper splits it into [w_hi | w_lo] once (cached on the tensor)
            ly["wt32"] = K.pack_conv_weight(w32, torch.float32, out_scale=scale32, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        # short-K layers of the VGG front (conv1_2 / conv2_1 / conv2_2 shapes): weights-stationary-in-registers direct conv
        # (csrc/conv_wreg.hip) when the run-time geometry has whole tiles; it reads the K-major row pack
        if bf16 and relu and (c_in_pad, c_out_pad, bool(pool)) in ((64, 64, True), (64, 128, False), (128, 128, True)) \
                and os.environ.get("VNQA_STEM_WREG", "1") != "0":
            ly["wt_rows"] = wt if tile is None else K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad,
                                                                       c_in_pad=c_in_pad)
        # wide 3x3 layers (conv21 .. conv32): patch-stationary kernel (csrc/conv_ps.hip, K-major weights) when the run-time
        # geometry qualifies (vnqa_conv_ps_supported); the implicit-GEMM tile above stays as the fallback
        if bf16 and tile == L.TILE_STEM_256x256 and w.shape[2] == 3 and os.environ.get("VNQA_STEM_PS", "1") != "0":
            ly["wt_ps"] = K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            if getattr(self, "hyb", False):      # the exact (BatchNorm-folded) weights, for the layers that run with split weights
                ly["wt32ps"] = K.pack_conv_weight(w32, torch.float32, out_scale=scale32, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        return ly

    def _compose_pair(self, c1, c2, bn):
        """Two stacked linear convs with frozen weights as ONE conv.
          y2 = s*(W2 * (W1 * x + b1) + b2) + t        (* = 3x3 'same' conv, s/t = folded eval BatchNorm)
             = Wc * x + bc  - R(x)                    with Wc = (s W2) (*) W1 (5x5), bc = s b2 + t + sum_taps(s W2) b1
        R is non-zero on the 1-pixel image border only: conv2 must see ZEROS outside the image, not conv1 evaluated
        there.  With Y1[q] = b1 + (W1 * x)[q] at the outside-ring positions q of the (H+2)x(W+2) grid,
          R[p] = sum_{taps d: p+d outside} (s W2)[d] Y1[p+d]
        i.e. one small GEMM for Y1 (ring im2col x W1) and four edge GEMMs (top / bottom / left / right, K = 3 c_mid);
        the composed kernel subtracts R from the border pixels' sums before ReLU and pooling."""
        dev = c1.weight.device
        # composed once, in fp64, on the HOST (a one-off 2-GFLOP product: keeps fp64 rocBLAS / im2col kernels out of the device
        # traces and costs ~0.3 s at construction)
        w1, b1 = c1.weight.detach().double().cpu(), c1.bias.detach().double().cpu()
        w2, b2 = c2.weight.detach().double().cpu(), c2.bias.detach().double().cpu()
        scale, shift = _fold_bn(bn)
        scale, shift = scale.double().cpu(), shift.double().cpu()
        w2 = w2 * scale.view(-1, 1, 1, 1)
        b2 = b2 * scale + shift
        wc = torch.nn.functional.conv2d(w1.permute(1, 0, 2, 3), w2.flip(2, 3), padding=2).permute(1, 0, 2, 3)   # [co,ci,5,5]
        bc = b2 + w2.sum((2, 3)) @ b1
        co, ci, cm = wc.shape[0], wc.shape[1], w1.shape[0]
        co_pad, ci_pad, cm_pad = L.round_up(co, 64), L.round_up(ci, 64), L.round_up(cm, 64)
        bf16 = L.is_half(self.cdt)
        tile = L.TILE_STEM_256x256 if (bf16 and co_pad >= 256) else (L.TILE_AUTO if bf16 else L.TILE_128x128)
        if bf16 and os.environ.get("VNQA_STEM_COMPOSE_TILE"):
            tile = int(os.environ["VNQA_STEM_COMPOSE_TILE"])      # A/B hook
        wcf = wc.float().contiguous().to(dev)
        wcf32 = wcf
        if getattr(self, "calib", None) is not None and bf16 and not self.x3:
            wcf = coherent_round(wcf, self.calib["od0"], self.cdt).contiguous()
        if tile in (L.TILE_STEM_256x256, L.TILE_STEM_I5_256x256) and os.environ.get("VNQA_STEM_TILED", "1") != "0" and not self.x3:
            wt = K.pack_conv_weight_tiled(wcf, self.cdt, tile, c_out_pad=co_pad, c_in_pad=ci_pad)
        else:
            wt = K.pack_conv_weight(wcf, self.cdt, c_out_pad=co_pad, c_in_pad=ci_pad)
        # ring GEMM operand: W1 K-major [cm_pad][9*ci_pad]; edge operands: (s W2) slices [co_pad][3*cm_pad]
        w1m = K.pack_conv_weight(w1.float().contiguous().to(dev), self.cdt, c_out_pad=cm_pad, c_in_pad=ci_pad).view(cm_pad, -1)

        def edge(sel):      # sel: [co,cm,3] -> [co_pad, 3*cm_pad] (slot-major, channels fastest)
            e = torch.zeros(co_pad, 3, cm_pad, dtype=torch.float64)
            e[:co, :, :cm] = sel.permute(0, 2, 1)
            return e.view(co_pad, -1).to(dev).to(self.cdt).contiguous()
        edges = dict(top=edge(w2[:, :, 0, :]), bottom=edge(w2[:, :, 2, :]), left=edge(w2[:, :, :, 0]), right=edge(w2[:, :, :, 2]))
        w1m32 = edges32 = None
        if self.x3:                  # fp32 operands of the two-product border GEMMs (split into [w_hi | w_lo] on first use, cached)
            w1m32 = K.pack_conv_weight(w1.float().contiguous().to(dev), torch.float32, c_out_pad=cm_pad, c_in_pad=ci_pad).view(cm_pad, -1)

            def edge32(sel):
                e = torch.zeros(co_pad, 3, cm_pad, dtype=torch.float64)
                e[:co, :, :cm] = sel.permute(0, 2, 1)
                return e.view(co_pad, -1).float().to(dev).contiguous()
            edges32 = dict(top=edge32(w2[:, :, 0, :]), bottom=edge32(w2[:, :, 2, :]), left=edge32(w2[:, :, :, 0]), right=edge32(w2[:, :, :, 2]))
        edges_all = torch.stack([edges[k] for k in ("top", "bottom", "left", "right")]).contiguous()   # [4, co_pad, 3*cm_pad]
        return dict(wt=wt, wt32=K.pack_conv_weight(wcf32, torch.float32, c_out_pad=co_pad, c_in_pad=ci_pad) if self.w2 else None,
                    bias=K.pad_vec(bc.float().to(dev), co_pad), b1=K.pad_vec(b1.float().to(dev), cm_pad), w1m=w1m, edges=edges,
                    w1m32=w1m32, edges32=edges32,
                    edges_all=edges_all,
                    c_in=ci, c_out=co, c_out_pad=co_pad, c_mid_pad=cm_pad, tile=tile, taps=25)

    def _run_composed(self, x, key, slot=0, use_slot=False):
        """x: halo-2 padded NHWC [n, H+4, W+4, ci_pad] -> relu/pool'ed output of the composed pair (halo 1)."""
        cp = self.composed
        xc = x                       # the composed conv's input
        # fp16x with a ROUNDED composed input (a plain fp16 tensor, the default): the border-correction GEMMs as two-product
        # launches too (fp16w keeps them on the plain fp16 kernels: measured 0.3 ms faster there at the same 0.95e-3) — conv11 at the ring positions from the fp16 input against [w_hi | w_lo], the four edge products
        # likewise from its fp16 output (that intermediate touches border pixels only) — instead of the exact-f32 matrix path
        # (1.5 of the fp16x stem's 15.5 ms)
        ring_w2 = cp.get("cdt") is None and cp.get("w1m32") is not None and L.is_half(x.dtype) and x.shape[-1] == cp["w1m32"].shape[1] // 9 and \
            self.x3 and K.x3_mode() == "x3" and os.environ.get("VNQA_RING_W2", "1") != "0"
        if self.x3 and cp.get("cdt") is None and x.dtype != torch.float32 and not ring_w2:
            x = self._x3_side        # (fp16x: the border-correction GEMMs read the fp32 copy, the 5x5 conv the x3 operand)
        n, hp, wp, ci_pad = x.shape
        H, W = hp - 4, wp - 4
        cm = cp["c_mid_pad"]
        # conv1 (+ b1) at the outside-ring positions, then the four edge GEMMs of conv2's outside taps -> ring of R[p]
        mode = os.environ.get("VNQA_RING_MODE", "implicit")
        if ring_w2:
            R = 2 * (W + 2) + 2 * H
            y1p = self._buf(key + ("y1p", H, W, "w2"), (n, R + 4, cm), dtype=L.half_dtype())
            with K.f32_conv_mode("w2"):
                K.conv2d_ring(x, cp["w1m32"].view(cm, 9, ci_pad), cp["b1"], H, W, out_padded=y1p)
                part = [K.ring_edge_conv(y1p, cp["edges32"][name], H, W, e)
                        for e, name in enumerate(("top", "bottom", "left", "right"))]
        elif mode == "implicit":
            # both correction operands as implicit GEMMs: conv11 at the ring positions straight from the halo-2 image, written
            # into a zero-separated ring layout; the four edge products as 1x3 convs along its rows (no im2col matrix, no
            # gathered edge operands: 147 + 4 x 48 MB less written and read back per 280-frame pass)
            R = 2 * (W + 2) + 2 * H
            y1p = self._buf(key + ("y1p", H, W), (n, R + 4, cm), dtype=cp.get("cdt"))
            K.conv2d_ring(x, cp["w1m"].view(cm, 9, ci_pad), cp["b1"], H, W, out_padded=y1p)
            part = [K.ring_edge_conv(y1p, cp["edges"][name], H, W, e)
                    for e, name in enumerate(("top", "bottom", "left", "right"))]
        else:
            if mode == "im2col":      # A/B: materialise the [n*ring, 9*ci] matrix, then a plain GEMM
                y1 = K.gemm_nt(K.ring_im2col(x, H, W), cp["w1m"], bias=cp["b1"], split_k=False)    # [n*ring, cm_pad]
            else:                     # "gather": implicit ring GEMM, gathered edge operands
                y1 = K.conv2d_ring(x, cp["w1m"].view(cm, 9, ci_pad), cp["b1"], H, W)
            if os.environ.get("VNQA_RING_GROUPED", "0") != "0":
                # the four edge products as ONE grouped GEMM on 256x256 tiles: 164 -> 110 us alone and the stem alone 1 % faster,
                # but END TO END the four small launches on 128x128 tiles (two workgroups per CU) interleave better with the
                # co-running trunk: same-box A/B 836 vs 830 clips/s at 224x224, 1013 vs 982 at 160x208 — so this is opt-in
                res = K.gemm_nt_grouped(K.ring_edge_gather_all(y1, n, H, W), cp["edges_all"])
                part = [res[0, :n * W], res[1, :n * W], res[2, :n * H], res[3, :n * H]]
            else:
                part = [K.gemm_nt(K.ring_edge_gather(y1, n, H, W, e), cp["edges"][name], split_k=False)
                        for e, name in enumerate(("top", "bottom", "left", "right"))]
        ring = K.ring_assemble(part[0], part[1], part[2], part[3], n, H, W)
        ho, wo = H // 2, W // 2
        plain = cp.get("cdt") is not None                     # the pair inside the plain fp16 prefix of an fp16x stem
        x3o = ((2 if "od2" in self.x3_round else 1) if self.x3 else 0) if not plain else 0
        if plain:
            out = self._buf(key + (ho, wo, "h16"), (n, ho + 2, wo + 2, cp["c_out_pad"]), dtype=cp["cdt"])
        elif self.x3:
            out = self._buf(key + (ho, wo, "x3", x3o), (n, ho + 2, wo + 2, (3 if x3o == 1 else 1) * cp["c_out_pad"]), dtype=L.half_dtype())
        else:
            out = self._buf(key + (ho, wo) + ((slot,) if use_slot else ()), (n, ho + 2, wo + 2, cp["c_out_pad"]))
        timed = self.timing is not None and ((self.x3 and not plain) or self.w2 or cp["tile"] in (L.TILE_STEM_256x256, L.TILE_STEM_I5_256x256, L.TILE_STEM_PS_224x256))
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        # every XCD computes ONE cout half of the composed conv (its L2 then holds 1.65 instead of 3.3 MB of weights): fabric-side reads
        # 1 022 -> 831 MB per launch (profiles/r04_pmc_traffic*.json), time unchanged (2.166 vs 2.161 ms; end to end 941-943 either way)
        xcd = L.CONV_XCD_SPLIT_N if ((not self.x3 or plain) and os.environ.get("VNQA_STEM_XCD_SPLIT", "1") == "1") else 0
        w2 = self.w2 and K.x3_mode() == "w2"
        y = K.conv2d_igemm(xc, cp["wt32"] if w2 else cp["wt"], bias=cp["bias"], relu=True, pool2=True, x_halo=2, y_halo=1, out=out,
                           tile=L.TILE_256x256 if w2 else cp["tile"], border_sub=ring, x3_out=x3o, desc_flags=0 if w2 else xcd)
        if timed:
            ev1.record()
            self.timing.append((ev0, ev1, 2.0 * n * H * W * cp["c_in"] * cp["c_out"] * 25,
                                "x3 product (split + conv_igemm_kernel raw + post)" if (self.x3 and not plain) else
                                "conv_igemm_kernel<..., TAG 4> (two products, x read twice along K)" if self.w2 else
                                ("conv_ps_kernel" if cp["tile"] == L.TILE_STEM_PS_224x256 else "conv_igemm_kernel")))
        if self._tap is not None:
            self._tap[key] = y
        return y

    def _c64_sched(self):
        """The two schedule words of the fused conv1 kernel's dynamic tile schedule: one pair per stream this plan runs on (a launch
        leaves them zero; two launches of one plan never overlap on different streams — the Trainer orders its inline and side-stream
        stem passes).  VNQA_C64_DYNAMIC=0: static stride."""
        if os.environ.get("VNQA_C64_DYNAMIC", "1") == "0":
            return None
        key = ("c64sched", torch.cuda.current_stream().cuda_stream)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.zeros(2, dtype=torch.int32, device="cuda")
        return t

    def _buf(self, key, shape, dtype=None):
        """Persistent zero-halo activation buffer, grown (never shrunk) along the image axis."""
        dtype = self.cdt if dtype is None else dtype
        key = key + (str(dtype),) if dtype != self.cdt else key
        cap = self._bufs.get(key)
        if cap is None or cap.shape[0] < shape[0] or tuple(cap.shape[1:]) != tuple(shape[1:]):
            cap = torch.zeros(shape, dtype=dtype, device="cuda")
            self._bufs[key] = cap
        return cap[:shape[0]]

    def _pair_geometry_ok(self, n, h, w, ly):
        """The pair path of precision 'fp16h' needs the patch-stationary kernel on conv22 (h x w maps, pooled) AND on conv31 / conv32
        (h/2 x w/2 maps, 2 C input channels): asked of the library once per geometry.  Where it does not serve them (the 10 x 13 maps
        of the reference's 160 x 208 frames) the three layers run exactly as in precision 'fp16'."""
        key = (n, h, w)
        ok = self._pair_ok.get(key)
        if ok is None:
            c = ly["c_out_pad"]
            ok = self._pair_ok[key] = bool(ly["pool"] and K.conv_ps_supported(n, h, w, c, c, 9, True) and
                                        K.conv_ps_supported(n, h // 2, w // 2, 3 * c, c, 9, False))
        return ok

    def _run(self, x, layers, tag, last_slot=0, first_index=0, final=True):
        for i, ly in enumerate(layers, first_index):
            n, hp, wp, _ = x.shape
            h, w = hp - 2, wp - 2
            ho, wo = (h // 2, w // 2) if ly["pool"] else (h, w)
            yh = ly.get("y_halo", 1)
            last = not (i + 1 < len(layers) + first_index)
            key = (tag, i, ho, wo) if not last else (tag, i, ho, wo, last_slot)
            # fp16x: consecutive layers hand each other the 16-bit x3 operand [hi | lo | hi] (no fp32 round trip); the chain's
            # last layer (`final`) writes fp32
            plain = ly.get("cdt") is not None                 # a layer of the plain 16-bit prefix inside the fp16x stem
            x3_out = self.x3 and not plain and K._F32_CONV_MODE[0] == "x3" and (not (last and final) or self.out_half)
            # precision 'fp16h': [hi | lo] pair tensors between conv22, conv31, conv32 and the trunk (see __init__)
            pair_rd = "wt_ps3" in ly and x.shape[-1] == ly["wt_ps3"].shape[2]
            pair_wr = int(ly.get("pair_out", 0))
            if pair_wr and not (yh == 1 and (pair_rd if "wt_ps3" in ly else self._pair_geometry_ok(n, h, w, ly))):
                pair_wr = 0
            if pair_wr:
                out = self._buf(key + ("pair",), (n, ho + 2, wo + 2, pair_wr * ly["c_out_pad"]))
            elif plain:
                out = self._buf(key + ("h16",), (n, ho + 2 * yh, wo + 2 * yh, ly["c_out_pad"]), dtype=ly["cdt"])
            elif x3_out:
                nxt = ("%s%d" % (tag, i + 1)) if not last else ("composed" if (tag == "vgg" and self.composed is not None) else "od0")
                x3_out = 2 if (nxt in self.x3_round or (last and final)) else 1
                out = self._buf(key + ("x3", x3_out), (n, ho + 2 * yh, wo + 2 * yh, (3 if x3_out == 1 else 1) * ly["c_out_pad"]),
                                dtype=L.half_dtype())
            else:
                out = self._buf(key, (n, ho + 2 * yh, wo + 2 * yh, ly["c_out_pad"]))
            post = ly["post"]
            tile = ly["tile"]
            timed = self.timing is not None and (tile in (L.TILE_STEM_256x256, L.TILE_STEM_I5_256x256, L.TILE_STEM_PS_224x256) or
                                                 ((self.x3 or self.w2) and ly["c_out_pad"] >= 256))      # (C_out = 512 layers, whichever kernel serves them)
            kname = "conv_igemm_kernel" if (not self.x3 or ly.get("cdt") is not None) else "x3 product (split + conv_igemm_kernel raw + post)"
            if timed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if pair_rd or pair_wr:
                kname = "conv_ps_kernel<%d>" % (28 if w % 28 == 0 else 14)
                x = K.conv2d_igemm(x, ly["wt_ps3"] if pair_rd else ly["wt_ps"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                   out=out, tile=L.TILE_STEM_PS_224x256, y_halo=yh, dual_out=pair_wr)
            elif self.w2 and "wt32" in ly and ly.get("cdt") is None and K.x3_mode() == "w2":
                # two products on the igemm's wrap variant: 512 x 128 tiles for the C_out = 128 layers, 256 x 256 for C_out = 512
                kname = "conv_igemm_kernel<..., TAG 4> (two products, x read twice along K)"
                x = K.conv2d_igemm(x, ly["wt32"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                   out=out, tile=15 if ly["c_out_pad"] == 128 else L.TILE_AUTO, y_halo=yh)
            elif "wt_rows" in ly and K.conv2d_wreg_supported(x, ly["wt_rows"], pool2=ly["pool"], y_halo=yh):
                x = K.conv2d_wreg(x, ly["wt_rows"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                  post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                  out=out, y_halo=yh, reserve_cus=self.reserve_cus)
            elif "wt_ps" in ly and yh == 1 and K.conv_ps_supported(n, h, w, x.shape[-1], ly["c_out_pad"], 9, ly["pool"]):
                kname = "conv_ps_kernel<%d>" % (28 if w % 28 == 0 else 14)      # (one entry per kernel SYMBOL, as rocprofv3 lists them)
                x = K.conv2d_igemm(x, ly["wt_ps"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                   out=out, tile=L.TILE_STEM_PS_224x256, y_halo=yh)
            elif tile is None:
                # C_in = 64 layers (conv1_2, conv2_1): persistent direct conv with LDS-resident weights
                x = K.conv2d_c64(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                 post_scale=post[0] if post else None, post_shift=post[1] if post else None, out=out,
                                 reserve_cus=self.reserve_cus)
            else:
                # (x3 mode: a plain 16-bit input with a rounded output runs as ONE fused two-product launch, no raw sums)
                fused_w2 = x3_out == 2 and L.is_half(x.dtype) and x.shape[-1] == ly["wt"].shape[2]
                x = K.conv2d_igemm(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                   out=out, tile=tile, y_halo=yh, x3_out=x3_out)
                if x3_out and ly.get("dual"):
                    # the composed pair's border correction runs on the exact-f32 GEMMs: this layer's output once more as fp32 —
                    # the rounded tensor itself when it is what the 5x5 conv reads, else the raw sums (still in this stream's
                    # x3 scratch) finished a second time
                    side = self._buf(key + ("f32side",), (n, ho + 2 * yh, wo + 2 * yh, ly["c_out_pad"]))
                    if x3_out == 2:
                        if os.environ.get("VNQA_RING_W2", "1") == "0":       # (else the border GEMMs read the fp16 tensor itself)
                            side.copy_(x)
                        self._x3_side = side
                    else:
                        self._x3_side = K.x3_post_again(side, n, h, w, ly["c_out_pad"], yh, bias=ly["bias"], relu=ly["relu"],
                                                        pool2=ly["pool"], post_scale=post[0] if post else None,
                                                        post_shift=post[1] if post else None)
            if timed:
                ev1.record()
                self.timing.append((ev0, ev1, 2.0 * n * h * w * ly["c_in"] * ly["c_out"] * 9 * (3 if pair_rd else 1), kname))
            if self._tap is not None:
                self._tap[(tag, i)] = x
        return x

    # ---- fused fast path: clip -> packed native features ------------------------------------
    @torch.no_grad()
    def forward_clip(self, clip, img_of, n_img, slot=0):
        """clip fp32 [B,3,H,W,T] on the GPU — or uint8 raw pixels k, meaning k / 255 exactly as the reference's loader forms it
        (eval/dataset.py:91; VNQADataset(uint8_video=True)) —; img_of int32 [B*T] (image index or -1).
        Returns padded NHWC [n_img, H/16+2, W/16+2, Cpad] in the compute dtype.
        `slot` selects one of several OUTPUT buffers (the intermediates are shared), so that the
        features of step i stay alive for its backward while step i+1's stem already runs."""
        assert self.vgg is not None and self.objdet is not None
        mode = "x3" if self.x3 else ("w2" if self.w2 else None)
        if mode is not None and K._F32_CONV_MODE[0] != mode:
            with K.f32_conv_mode(mode):
                return self.forward_clip(clip, img_of, n_img, slot)
        B, _, H, W, T = clip.shape
        ly = self.layers_vgg[0]
        if L.is_half(ly.get("cdt", self.cdt)) and ly["tile"] is None and os.environ.get("VNQA_FUSE_FIRST", "1") != "0":
            # conv1_1 evaluated inside the conv1_2 kernel from a 4-channel bf16 image list: its 64-channel output
            # (1.8 GB at 280 x 224 x 224) never goes to HBM
            hdt = ly.get("cdt", self.cdt)
            img4 = self._buf(("img4", H, W), (n_img, H + 4, W + 4, 4), dtype=hdt)
            K.clip_to_nhwc4(clip, img_of, n_img, out=img4)
            ho, wo = (H // 2, W // 2) if ly["pool"] else (H, W)
            out = self._buf(("vgg", 0, ho, wo), (n_img, ho + 2, wo + 2, ly["c_out_pad"]), dtype=hdt)
            post = ly["post"]
            split = int(os.environ.get("VNQA_C64_SPLIT", "1"))     # A/B hook: the persistent kernel as several shorter launches
            step = (n_img + split - 1) // split
            for n0 in range(0, n_img, step):
                K.conv_first_c64(img4[n0:n0 + step], self.first[0], self.first[1], ly["wt"], bias=ly["bias"], relu=ly["relu"],
                                 pool2=ly["pool"], post_scale=post[0] if post else None,
                                 post_shift=post[1] if post else None, out=out[n0:n0 + step], reserve_cus=self.reserve_cus,
                                 sched=self._c64_sched())
            x = out
            x = self._run(x, self.layers_vgg[1:], "vgg", first_index=1, final=False)
        else:
            if clip.dtype == torch.uint8:        # raw pixels: the un-fused first conv reads the fp32 clip
                clip = K.expand_u8_clip(clip)
            a = self._buf(("first", H, W), (n_img, H + 2, W + 2, 64))
            K.conv_first(clip, self.first[0], self.first[1], img_of, n_img, self.cdt, out=a)
            if self._tap is not None:
                self._tap["first"] = a
            x = self._run(a, self.layers_vgg, "vgg", final=False)
        return self._run_od(x, "od", slot)

    def _run_od(self, x, tag, slot=0):
        """ObjDetectCNN trunk on a padded NHWC map (halo 2 when the first pair is composed, else halo 1)."""
        if self.composed is None:
            return self._run(x, self.layers_od, tag, last_slot=slot)
        y = self._run_composed(x, (tag, "c"))
        return self._run(y, self.layers_od[2:], tag, last_slot=slot, first_index=2)

    # ---- drop-in per-module paths (reference tensor layouts in and out) -----------------------
    @torch.no_grad()
    def vgg_nchw(self, x):
        N, _, H, W = x.shape
        img_of = torch.arange(N, dtype=torch.int32, device=x.device)
        a = self._buf(("first", H, W), (N, H + 2, W + 2, 64))
        K.conv_first(x.float().contiguous().view(N, 3, H, W, 1), self.first[0], self.first[1], img_of, N,
                     self.cdt, out=a)
        layers = [dict(ly) for ly in self.layers_vgg]
        layers[-1]["post"] = None      # standalone VGG front: bn_input belongs to ObjDetectCNN
        y = self._run(a, layers, "vggs")
        return K.nhwc_to_nchw(y, 128)

    @torch.no_grad()
    def objdet_nchw(self, x):
        s, t = self.bn_input
        xin = x.float() * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)          # bn_input, eval (obj_detector.py:70)
        xn = K.nchw_to_nhwc(xin, self.cdt)
        if self.composed is not None:
            xn = torch.nn.functional.pad(xn, (0, 0, 1, 1, 1, 1))              # halo 1 -> halo 2
        y = self._run_od(xn, "ods")
        return K.nhwc_to_nchw(y, self.out_channels)
