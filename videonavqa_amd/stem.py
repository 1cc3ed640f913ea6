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

    def __init__(self, precision='fp16h'):
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
        if self._plan is None:      # (the per-module drop-in path: an API path, not the fast one; 'fp16h' runs it as plain fp16)
            self._plan = FrozenStem(self, None, "fp16" if self.precision == "fp16h" else self.precision)
        return self._plan.vgg_nchw(x)


def get_frcnn_feature_extractor(path=None, precision='fp16h'):
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


def default_calibration_frames(n, height=224, width=224, seed=4242):
    """The seeded SYNTHETIC calibration frames ("noise" calibration): the first half uniform noise (the benchmark's kind of clip), the
    second half smooth — 14 x 14 noise upsampled bilinearly, with a per-frame brightness.  Second moments measured on the mixture
    generalise: on a THIRD kind of clip (piecewise-constant 'blocks', tools/error_budget.py --data blocks) the nine stem weight roundings
    cost 0.022e-6 of squared logits error with this mixture against 0.082 calibrated on noise alone (0.117 mean-coherent, 0.67 nearest;
    profiles/r05_gptq_stem_weights.txt), with nothing lost on noise clips (0.008)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(n - n // 2, 3, height, width, generator=g)
    if n // 2 == 0:
        return a
    low = torch.rand(n // 2, 3, max(height // 16, 1), max(width // 16, 1), generator=g)
    b = torch.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)
    b = (b * (0.3 + 0.7 * torch.rand(n // 2, 1, 1, 1, generator=g))).clamp_(0, 1)
    return torch.cat([a, b])


@torch.no_grad()
def calibration_means(vgg, od, frames=None, n_frames=4, height=224, width=224, seed=4242, second_moments=False):
    """Per-input-channel mean activation at every stem layer's input — what coherent_round cancels against — from ONE pass of
    calibration frames [N, 3, H, W] (values in [0, 1]) through the library's own exact-f32 stem (a FrozenStem of precision 'fp32'
    with its layer outputs tapped).  Default frames: default_calibration_frames (seeded: half uniform noise, half smooth); a deployment
    passes frames of its own videos.  Keys: first = conv1_1's input, vgg0 / vgg1 / vgg2 = conv1_2 / conv2_1 / conv2_2, od0 = conv11 (and
    the composed 5x5 pair), od1 = conv12, od2 .. od5 = conv21 .. conv32."""
    dev = vgg.features["0"].weight.device
    if frames is None:
        frames = default_calibration_frames(n_frames, height, width, seed)
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
    m["feat"] = mean(("od", 5), od.conv32.out_channels)          # the features themselves (round 6: mean-shifted storage of the stem's output)
    if second_moments:
        # ... and the second moment of every layer's input PATCHES (what second_order_round minimises against), on the device; od1
        # (conv12's input when the pair runs layer by layer) is not stored by the composed pass: that layer keeps coherent_round
        def nchw(key, c_, halo=1):
            t = ref._tap[key]
            return t[:, halo:t.shape[1] - halo, halo:t.shape[2] - halo, :c_].permute(0, 3, 1, 2)
        halo2 = 2 if ref.composed is not None else 1
        H = {"first": patch_second_moment(frames, 3), "vgg0": patch_second_moment(nchw("first", 64), 3),
             "vgg1": patch_second_moment(nchw(("vgg", 0), 64), 3), "vgg2": patch_second_moment(nchw(("vgg", 1), 128), 3),
             "od0": patch_second_moment(nchw(("vgg", 2), 128, halo2), 3), "od3": patch_second_moment(nchw(("od", 2), od.conv21.out_channels), 3),
             "od4": patch_second_moment(nchw(("od", 3), c), 3), "od5": patch_second_moment(nchw(("od", 4), od.conv31.out_channels), 3)}
        if ref.composed is not None:
            H["od0_5x5"] = patch_second_moment(nchw(("vgg", 2), 128, halo2), 5)
            H["od2"] = patch_second_moment(nchw(("od", "c"), c), 3)
        else:
            H["od2"] = patch_second_moment(nchw(("od", 1), c), 3)
        m["_H"] = H
    return m


@torch.no_grad()
def second_order_round(w, H, dtype, damp=0.01, block=128):
    """Weights w [c_out, c_in, kh, kw] (fp32) rounded to `dtype` (a 16-bit format) against the SECOND MOMENT H = E[p p^T] (float64
    [K, K], K = c_in * kh * kw in w.reshape(c_out, -1)'s order) of the layer's input patches p: E[(p . dw)^2] = dw^T H dw, the mean
    squared error the rounding adds to the layer's output, is minimised greedily — column by column in order of decreasing H_jj, the
    rounding error of column j is pushed onto the not yet rounded columns along H^-1 before they are rounded (the GPTQ / OBQ
    sequential rounding, here onto the fp16 / bf16 grid).  coherent_round cancels the error against the MEAN patch only (first
    moment); on the frozen stem this form leaves 1/8 of its squared logits error on clips like the calibration frames and 0.43 of it
    on clips unlike them (tools/experiments/gptq_stem_weights.py, profiles/r05_gptq_stem_weights.txt) — little enough that
    precision 'fp16h' no longer spends a product on conv31's / conv32's weight roundings.  Returns fp32 values exactly
    representable in `dtype` (None if H could not be factorised).  Deterministic for given inputs (fp64 Cholesky + fixed-order updates)."""
    co = w.shape[0]
    W = w.detach().reshape(co, -1).double().clone()
    K_ = W.shape[1]
    H = H.to(W.device).double().clone()
    if not bool(torch.isfinite(H).all()) or float(torch.diagonal(H).sum()) <= 0.0:      # no signal on the calibration frames: nearest
        return w.detach().float().to(dtype).float()
    d = torch.diagonal(H)
    dead = d == 0                                   # inputs that are always zero (padded channels): any rounding is free
    d[dead] = 1.0
    perm = torch.argsort(torch.diagonal(H), descending=True)
    W = W[:, perm]
    H = H[perm][:, perm]
    # U = the upper Cholesky factor of H^-1 (H^-1 = U^T U), WITHOUT forming H^-1: with P the index reversal, P H P = L L^T gives
    # H^-1 = (P L^-1 P)^T (P L^-1 P) and P L^-1 P is upper triangular — one factorisation of the damped (positive definite by
    # construction) H and one triangular solve.  (The textbook route, cholesky(cholesky_inverse(cholesky(H))), factorises an explicitly
    # inverted matrix and was seen to fail — "not positive-definite" at a leading minor of order 4260 of 4608 — in one of two
    # processes calibrating concurrently on one GPU.)  Should the factorisation still fail, the damping grows; the last resort is
    # round-to-nearest (the caller's coherent rounding is then the better fallback: see FrozenStem._round).
    eye = torch.eye(K_, dtype=H.dtype, device=H.device)
    mean_diag = float(torch.diagonal(H).mean())
    U = None
    for factor in (1.0, 10.0, 100.0):
        try:
            Lr = torch.linalg.cholesky((H + eye * (damp * factor * mean_diag)).flip(0).flip(1))
            U = torch.linalg.solve_triangular(Lr, eye, upper=False).flip(0).flip(1).contiguous()
            if bool(torch.isfinite(U).all()):
                break
            U = None
        except RuntimeError:          # torch._C._LinAlgError
            U = None
    if U is None:
        return None
    if W.is_cuda:      # one launch per layer (csrc/round2.hip: a workgroup per output channel, the row in LDS as float64)
        assert dtype == L.half_dtype(), "the library rounds onto ITS 16-bit format (one format per process)"
        Qf = torch.empty(W.shape, dtype=torch.float32, device=W.device)
        Wc, Uc = W.contiguous(), U.contiguous()
        L.check(L.lib().vnqa_second_order_round(L.ptr(Wc), L.ptr(Uc), L.ptr(Qf), co, K_, L.stream()), "vnqa_second_order_round")
        Q = Qf.double()
    else:              # the same recursion in tensor arithmetic (host tensors: tests/test_coherent_round.py)
        Q = torch.zeros_like(W)
        for b0 in range(0, K_, block):
            b1 = min(b0 + block, K_)
            Wb = W[:, b0:b1].clone()
            Eb = torch.zeros_like(Wb)
            Ub = U[b0:b1, b0:b1]
            for j in range(b1 - b0):
                wj = Wb[:, j]
                q = wj.float().to(dtype).double()
                Q[:, b0 + j] = q
                e = (wj - q) / Ub[j, j]
                Wb[:, j:] -= e.unsqueeze(1) * Ub[j, j:].unsqueeze(0)
                Eb[:, j] = e
            W[:, b1:] -= Eb @ U[b0:b1, b1:]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(K_, device=perm.device)
    return Q[:, inv].float().view_as(w)


@torch.no_grad()
def patch_second_moment(x, k, max_rows=400000):
    """H = E[p p^T] (float64 [K, K], K = C k k, channel-major / taps minor) over the k x k 'same'-padded patches of x [N, C, H, W];
    positions of large maps are subsampled (seeded) to about max_rows."""
    N, C, Hh, Ww = x.shape
    K_ = C * k * k
    H = torch.zeros(K_, K_, dtype=torch.float64, device=x.device)
    per = Hh * Ww
    stride = max(1, (N * per + max_rows - 1) // max_rows)
    g = torch.Generator(device="cpu").manual_seed(5)
    rows = 0
    for n in range(N):
        p = torch.nn.functional.unfold(x[n:n + 1].float(), k, padding=k // 2)[0].t()
        if stride > 1:
            p = p[torch.randperm(per, generator=g)[:per // stride].to(x.device)]
        p = p.double()
        H += p.t() @ p
        rows += p.shape[0]
    return H / max(rows, 1)


FEATURE_TWIN = 1             # precision 'fp16h' with mean-shifted features: write them twice ([x' | x']) for conv_init's two products (split weights)
MEAN_SHIFT = 1               # 16-bit precisions with a calibration: store the stem's activations minus their calibration channel means (see FrozenStem)
SPLIT_DEPTH = 0              # precision 'fp16h': how many of the stem's LAST stored activations are split tensors (see FrozenStem);
                             # 0 = automatic: none needed with mean-shifted storage (1: the features, only where the consumer cannot
                             # take shifted ones), 3 without it (round 5's form)
RING_COMPOSED_EDGES = True   # the composed pair's border correction as four composed 1x5 / 5x1 edge convs + a corner term (round 6); False: the
                             # two-step form (conv11 on the outside ring, then conv12's edge taps: RING_EDGE_LAUNCHES) — the A/B partner
COMPOSED_PS = os.environ.get("VNQA_COMPOSED_PS", "1") != "0"   # the composed 5x5 on the patch-stationary 2-D tiles (8 x 28 pixels) where they serve the
                             # geometry (224 x 224 frames: 56 x 56 maps; not the reference's 160 x 208: 40 x 52) — 5 % less kernel time than the flat
                             # 256-pixel igemm tile (profiles/r06_composed_ps_ab.txt); False / VNQA_COMPOSED_PS=0: the igemm tile everywhere
COMPOSED_PS_XCD = os.environ.get("VNQA_COMPOSED_PS_XCD", "1") != "0"      # ... with one cout half per XCD (fabric-side reads 1 159 -> 953 MB per launch)
RING_EDGE_LAUNCHES = True    # conv11 on the outside ring as four 3-tap launches (False: one 9-tap launch, same bits: the A/B partner)
CALIBRATION_FRAMES = 40      # frames of the default ("noise") calibration pass: 40 x 196 patches > K = 4608 of the 14 x 14 layers


def _fold_bn(bn):
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + BN_EPS)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


class FrozenStem(object):
    """Execution plan (packed weights + persistent activation buffers) for the frozen stem.

    precision: 'fp16h' (default) | 'fp16' | 'bf16' (16-bit storage, fp32 accumulate: the MFMA fast path) | 'fp32' (the exact-f32 parity path).

    calibration: what the frozen 16-bit weights are rounded against and what the stored activations are shifted by.  None = round to
    nearest, un-shifted storage.  "noise" (= "auto") or a tensor of frames [N, 3, H, W] (values in [0, 1]; `--stem_calibration data`) = ONE
    exact-f32 stem pass over those frames (CALIBRATION_FRAMES seeded synthetic frames by default, half uniform noise, half smooth:
    default_calibration_frames) measures every layer's input-patch second moment H = E[p p^T] and every stored tensor's channel means:
      * each layer's weights (BatchNorm scale folded, the conv11.conv12 pair composed) are rounded by second_order_round — column by
        column, the error pushed onto the not yet rounded columns along H^-1: the nine weight roundings cost 0.009e-6 of squared logits
        error instead of 0.58 (nearest) / 0.074 (round 4's mean-cancelling coherent_round, still the form for means-only calibrations);
      * every stored activation — the image list, conv1_1's LDS-resident output, the fused first conv's, conv2_1's, conv2_2's, the
        composed pair's, conv21's, conv22's, conv31's and (for the FiLM trunk, `split_features=True`) the features — is stored
        MEAN-SHIFTED, value minus the channel mean (_setup_mean_shift: the consumer's bias absorbs the mean, the halo holds -mean, no
        extra product): the seven single-product stem roundings cost 0.08 instead of 0.27e-6 on noise clips, 0.17 instead of 0.75 on
        piecewise-constant ones.
    A dict = an earlier calibration (a checkpoint's `extra_state['_stem_calibration']`): its "frames" entry — "noise" or the frames — is
    what the pass is redone from; a means-only dict gets the first-order rounding and whatever shifts it has means for.
    VNQA_COHERENT_ROUND=0 / VNQA_MEAN_SHIFT=0 turn the two off (A/B partners).

    precision 'fp16h' and split_depth: with mean-shifted storage the stem needs NO split tensor (split_depth 0 = automatic); the features go
    out twice, [x' | x'], for conv_init's two products against split weights (FEATURE_TWIN).  split_depth = 3 additionally keeps conv22's and
    conv31's outputs as [hi | lo] tensors read by two-product layers (the pooling-head models ask for it: `stem_split_depth`), 4 / 5 add
    conv21's / the composed pair's (measured: nothing gained, -6 % / -13 %).  Without channel means (older calibrations) the round-5 form
    runs: split depth 3 with [hi | lo | hi] features into a three-product conv_init.  Dual outputs come from the patch-stationary kernel's
    fp32 epilogue and, where it does not serve the geometry (the 10 x 13 maps of 160 x 208 frames) or the layer (the composed 5x5), from the
    implicit-GEMM tile's.  split_features=False: plain un-shifted features (consumers that are not the FiLM trunk: MACNetwork)."""

    def __init__(self, vgg, objdet, precision='fp16h', calibration="auto", split_features=True, reserve_cus=0, split_depth=None):
        from .models.common import compute_dtype
        self.cdt = compute_dtype(precision)
        self.hyb = precision == "fp16h"
        self.split_features = bool(split_features) and self.hyb
        # split_features=True also says "the consumer is the FiLM trunk's conv_init": with a calibration that knows the features' channel
        # means the features go out as ONE plain tensor holding v - mean_c (feature_shift) — conv_init absorbs the mean in its bias, its
        # weight rounding then multiplies a zero-mean input, and its three products on [hi | lo | hi] features are not needed
        self.trunk_features = bool(split_features)
        self.feature_shift = None
        # 1: the features alone (conv_init's three products; everything before them one product on mean-shifted storage); 3 (round 5):
        # conv22's, conv31's and conv32's outputs; 4: + conv21's (conv22 then runs as two products); 5: + the composed conv11.conv12
        # pair's (conv21 as two products; the pair's dual output comes from the implicit-GEMM tile's fp32 epilogue)
        self._split_depth_arg = int(SPLIT_DEPTH if split_depth is None else split_depth)
        self.split_depth = 0          # (fixed below, once the calibration says whether mean-shifted storage is available)
        self.vgg, self.objdet = vgg, objdet
        self.layers_vgg, self.layers_od = [], []
        self.composed = None
        self.first = None
        self._bufs = {}
        self.split_segs = 0
        # CUs the persistent one-workgroup-per-CU kernels (fused conv1, C_in = 64 direct conv, weights-in-registers conv) leave to
        # other streams, passed with every call (vnqa_conv_desc.flags): the Trainer sets it to its stem stream's CU reservation
        self.reserve_cus = int(reserve_cus)
        self.timing = None   # bench hook: list collecting (start event, end event, FLOPs, kernel) of the C_out = 512 stem launches
        self.calib = None
        self._tap = None     # calibration hook: {layer key: output tensor} filled by _run / _run_composed
        if isinstance(calibration, str) and calibration == "auto":
            calibration = "noise" if os.environ.get("VNQA_COHERENT_ROUND") != "0" else None
        self._H = None       # second moments of every layer's input patches (device, float64): alive during construction only
        frames = None
        if isinstance(calibration, dict):      # an earlier calibration (a checkpoint's: eval/q_and_v_test.py)
            if precision != "fp32":
                self.calib = {k: (v if isinstance(v, str) else torch.as_tensor(v).float().cpu()) for k, v in calibration.items()}
                # ... made from these frames: the second-order rounding is redone from them (same frames + same frozen weights = the
                # same 16-bit weights the model was trained behind); a calibration without frames (written before round 5) has the
                # means only and gets the first-order rounding it was made for
                frames = self.calib.get("frames")
        elif calibration is not None and precision != "fp32":
            frames = calibration
        if frames is not None and vgg is not None and objdet is not None and vgg.features["0"].weight.is_cuda:
            noise = isinstance(frames, str)
            st = calibration_means(vgg, objdet, None if noise else frames, n_frames=CALIBRATION_FRAMES, second_moments=True)
            self._H = st.pop("_H")
            self.calib = st
            self.calib["frames"] = "noise" if noise else torch.as_tensor(frames).float().cpu()
        self.second_order = self._H is not None
        # mean-shifted storage needs the calibration's channel means of every stored tensor, and the whole stem in one plan
        self._shift_planned = bool(MEAN_SHIFT and os.environ.get("VNQA_MEAN_SHIFT", "1") != "0" and L.is_half(self.cdt) and vgg is not None
                                   and objdet is not None and self.calib is not None
                                   and all(k in self.calib for k in ("first", "vgg0", "vgg1", "vgg2", "od0", "od2", "od3")))
        if self.hyb:
            auto = 1 if self._shift_planned else 3
            self.split_depth = min(5, max(1, self._split_depth_arg if self._split_depth_arg > 0 else auto))
        cm = lambda k: k if self.calib is not None else None
        if vgg is not None:
            f = vgg.features
            w0 = self._round(f["0"].weight.detach().float().contiguous(), cm("first"), L.half_dtype()).contiguous()
            self.first = (w0, f["0"].bias.detach().float().contiguous())
            self.layers_vgg = [self._layer(f["2"], relu=True, pool=True, m=cm("vgg0")),
                               self._layer(f["5"], relu=True, pool=False, m=cm("vgg1")),
                               self._layer(f["7"], relu=True, pool=True, m=cm("vgg2"))]
        if objdet is not None:
            od = objdet
            self.bn_input = _fold_bn(od.bn_input)
            self.layers_od = [self._layer(od.conv11, m=cm("od0")),
                              self._layer(od.conv12, bn=od.bn1, relu=True, pool=True, m=cm("od1")),
                              self._layer(od.conv21, m=cm("od2")),
                              self._layer(od.conv22, bn=od.bn2, relu=True, pool=True, m=cm("od3")),
                              self._layer(od.conv31, m=cm("od4")),
                              self._layer(od.conv32, bn=od.bn3, relu=True, pool=False, m=cm("od5"))]
            if self.hyb:
                # conv22 writes a SPLIT tensor, conv31 reads it and writes one, conv32 reads it and writes the split features
                # [hi | lo | hi] for conv_init (or a plain tensor for consumers that read no split tensors).
                #  * second-order rounded weights (the default: calibration frames given): the 16-bit weights wq are within 1e-8 of
                #    exact in the squared logits error, so a split-reading layer is TWO products — a plain conv over 2 C input
                #    channels [hi | lo] against [wq | wq];
                #  * otherwise (calibration off / a means-only calibration): THREE products over [hi | lo | hi] against the split
                #    exact weights [w_hi | w_hi | w_lo] (BatchNorm folded in fp32 first).
                segs = 2 if self.second_order else 3
                first = 6 - self.split_depth            # first layer of layers_od that WRITES a split tensor (3 = conv22; 2 = conv21)
                for i, ly in enumerate(self.layers_od):
                    rd = i > max(first, 1) or (i == 2 and self.split_depth >= 5)      # (conv21 reads the composed pair's split output)
                    wr = 0 if i < first else ((3 if self.split_features else 0) if i == 5 else segs)
                    if not (rd or wr):
                        ly.pop("wt32ps", None)
                        continue
                    if "wt_ps" in ly:
                        ly["split_out"] = wr
                        if rd and segs == 3:
                            ly["wt_split"] = K.split_weight3(ly["wt32ps"])
                        elif rd:
                            ly["wt_split"] = torch.cat([ly["wt_ps"], ly["wt_ps"]], dim=2).contiguous()
                    ly.pop("wt32ps", None)
                self.split_segs = segs
            # conv12 is applied straight to conv11's output (obj_detector.py:72: no nonlinearity between the two convs of
            # a pair) and both are frozen: when the pair's 3x3 (c_in -> c_mid) . 3x3 (c_mid -> c_out) costs more than one
            # 5x5 (c_in -> c_out) — 9*c_in + 9*c_mid > 25*c_in, true for 128 -> 512 -> 512 only — it is evaluated as the
            # composed 5x5 conv plus an exact correction on the image border (see _compose_pair).  VNQA_STEM_COMPOSE=0: layer by layer
            ci, cmid = od.conv11.in_channels, od.conv11.out_channels
            if os.environ.get("VNQA_STEM_COMPOSE", "1") != "0" and 9 * ci + 9 * cmid > 25 * ci:
                self.composed = self._compose_pair(od.conv11, od.conv12, od.bn1)
            self.out_channels = od.conv32.out_channels
            if vgg is not None:
                # bn_input becomes the post-affine of the last VGG layer's epilogue
                s, t = self.bn_input
                self.layers_vgg[-1]["post"] = (K.pad_vec(s, 128), K.pad_vec(t, 128))
                if self.composed is not None:
                    self.layers_vgg[-1]["y_halo"] = 2        # the composed 5x5 conv reads a halo-2 image

        self.shift = {}
        if self._shift_planned:
            self._setup_mean_shift()
        for ly in self.layers_vgg + self.layers_od + ([self.composed] if self.composed is not None else []):
            ly.pop("_wsum", None)
        self._H = None       # (0.8 GB of float64 moments: construction only)

    @property
    def feature_segs(self):
        """Channel segments of forward_clip's output: 3 = a split tensor [hi | lo | hi] (precision 'fp16h' with split_features), 1 = a
        plain tensor.  `split_active`: the number of stored activations this plan keeps as split tensors (0: none)."""
        if self.layers_od and self.layers_od[-1].get("twin_out"):
            return 2          # [x' | x']: the mean-shifted features written twice (conv_init's two products against split weights)
        return 3 if (self.split_features and self.layers_od and int(self.layers_od[-1].get("split_out", 0)) == 3) else 1

    def plain_features(self, feats):
        """forward_clip's output as the plain fp32 feature tensor [n, h+2, w+2, C_pad] (zero halo) whatever its storage form: a split
        tensor's hi + lo, a mean-shifted tensor's value + shift.  For tests, tools and consumers that read plain tensors."""
        c = self.layers_od[-1]["c_out_pad"]
        f = feats[..., :c].float()
        if feats.shape[-1] >= 2 * c and not self.layers_od[-1].get("twin_out"):
            f = f + feats[..., c:2 * c].float()
        if self.feature_shift is not None:
            f = f + self.feature_shift.view(1, 1, 1, -1)
            f[:, 0] = 0
            f[:, -1] = 0
            f[:, :, 0] = 0
            f[:, :, -1] = 0
        return f

    @property
    def split_active(self):
        """How many of the stem's stored activations this plan keeps as split tensors."""
        return sum(1 for ly in self.layers_od if int(ly.get("split_out", 0)) > 0) + \
            (1 if (self.split_depth >= 5 and self.composed is not None and self.layers_od and "wt_split" in self.layers_od[2]) else 0)

    def packed_tensors(self):
        """Every device tensor of the execution plan (packed / tiled / split weights, biases, the ring operands): what a data-parallel
        run broadcasts from rank 0 (Trainer.sync_replicas) so that all replicas multiply with bit-identical 16-bit stem weights even if
        a rank's calibration pass rounded one tie the other way."""
        out, seen = [], set()

        def walk(v):
            if isinstance(v, K.TiledWeight):
                v = v.data
            if torch.is_tensor(v):
                if v.is_cuda and v.data_ptr() not in seen:
                    seen.add(v.data_ptr())
                    out.append(v)
            elif isinstance(v, dict):
                for x in v.values():
                    walk(x)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    walk(x)
        for part in (self.first, self.layers_vgg, self.layers_od, self.composed, getattr(self, "bn_input", None),
                     getattr(self, "shift", None), getattr(self, "_b1_pair", None)):
            walk(part)
        return out

    def _setup_mean_shift(self):
        """MEAN-SHIFTED STORAGE (round 6).  A 16-bit store rounds to a RELATIVE 2^-12 of the stored value, so a tensor with a large
        per-channel mean and a small spread around it (post-ReLU / post-pool activations, flat image regions) loses its information in
        the rounding of the mean: measured with tools/experiments/precision_budget.py, the seven single-product stem roundings cost
        0.75e-6 of squared logits error on piecewise-constant clips and 0.27e-6 on noise clips — and 0.17 / 0.08e-6 when the stored
        value is v - mu_c instead (mu_c: the calibration frames' channel mean, rounded to the storage format).  No extra product:
          * the PRODUCER subtracts mu_c in fp32 before its one storage rounding (post affine of the fused first conv / the
            weights-in-registers conv; VNQA_CONV_F32_EPILOGUE on the composed pair; folded into the bias where no ReLU / pool
            sits between conv and store: conv21; the image list by vnqa_clip_to_nhwc4_shifted; conv1_1's LDS-resident output by
            VNQA_CONV_FIRST_MID_SHIFT);
          * the tensor's HALO holds -mu_c — what the zero padding becomes — written once per buffer;
          * the CONSUMER's bias gets sum_taps(Wq) mu: exact, because (x' + mu) is the unshifted tensor EVERYWHERE incl. the padding,
            and Wq are the very 16-bit weights the kernel multiplies with (the border ring of the composed pair likewise).
        Split tensors (precision 'fp16h') shift their hi half; the features handed to the trainable trunk stay unshifted."""
        dev = self.first[0].device
        hd = self.cdt
        cal = self.calib

        def mu_of(key, c, c_pad):
            m = torch.zeros(c_pad, dtype=torch.float32)
            m[:c] = torch.as_tensor(cal[key]).float().reshape(-1)[:c]
            return m.to(hd).float().to(dev)                       # exactly representable in the storage format

        def corr(wsum, mu, c_out_pad):
            ci = wsum.shape[1]
            out = torch.zeros(c_out_pad, dtype=torch.float64, device=dev)
            out[:wsum.shape[0]] = wsum.to(dev) @ mu[:ci].double()
            return out.float()
        v0, v1, v2 = self.layers_vgg
        od = self.layers_od
        mu = {"clip": mu_of("first", 3, 4), "c11": mu_of("vgg0", 64, 64), "c12": mu_of("vgg1", v0["c_out"], v0["c_out_pad"]),
              "c21": mu_of("vgg2", v1["c_out"], v1["c_out_pad"]), "c22": mu_of("od0", v2["c_out"], v2["c_out_pad"]),
              "comp": mu_of("od2", od[1]["c_out"], od[1]["c_out_pad"]), "od21": mu_of("od3", od[2]["c_out"], od[2]["c_out_pad"])}
        if v0["tile"] is not None:           # (only the fused conv1 path carries the first two shifts)
            mu["clip"].zero_()
            mu["c11"].zero_()
        ones = lambda n: torch.ones(n, dtype=torch.float32, device=dev)
        # clip -> conv1_1
        w0, b0 = self.first
        self.first = (w0, b0 + corr(w0.to(hd).double().sum((2, 3)), mu["clip"], 64))
        # conv1_1's LDS-resident output -> conv1_2; conv1_2's output c12
        v0["bias"] = v0["bias"] + corr(v0["_wsum"], mu["c11"], v0["c_out_pad"])
        v0["post"] = (ones(v0["c_out_pad"]), -mu["c12"])
        v0["out_shift"] = mu["c12"]
        self._b1_pair = torch.cat([self.first[1].float().reshape(-1)[:64], mu["c11"][:64]]).contiguous()
        # conv2_1: reads c12, writes c21
        v1["bias"] = v1["bias"] + corr(v1["_wsum"], mu["c12"], v1["c_out_pad"])
        v1["post"] = (ones(v1["c_out_pad"]), -mu["c21"])
        v1["out_shift"] = mu["c21"]
        # conv2_2: reads c21, writes c22 through bn_input's affine (already its post)
        v2["bias"] = v2["bias"] + corr(v2["_wsum"], mu["c21"], v2["c_out_pad"])
        s_in, t_in = v2["post"] if v2["post"] is not None else (ones(v2["c_out_pad"]), torch.zeros(v2["c_out_pad"], device=dev))
        v2["post"] = (s_in, t_in - mu["c22"])
        v2["out_shift"] = mu["c22"]
        # the conv11.conv12 pair: reads c22, writes comp
        if self.composed is not None:
            cp = self.composed
            cp["bias"] = cp["bias"] + corr(cp["_wsum"], mu["c22"], cp["c_out_pad"])
            # the ring's conv11 multiplies its OWN 16-bit pack; the four edge launches compute three taps each (the other six see the
            # padding, whose unshifted value is 0): each edge's bias is corrected by exactly the taps it computes
            w1 = cp["w1m"].view(cp["c_mid_pad"], 9, -1).double()
            cp["b1_edges"] = [cp["b1"] + corr(w1[:, list(t), :].sum(1), mu["c22"], cp["c_mid_pad"]) for t in K.RING_EDGE_TAPS]
            cp["b1"] = cp["b1"] + corr(w1.sum(1), mu["c22"], cp["c_mid_pad"])            # (the one-launch nine-tap form)
            # the composed edge convs read the shifted tensor (halo -mu) through all five taps; the corner GEMM reads the corner pixels
            cp["edge5_bias"] = [b + corr(w.double().sum(1), mu["c22"], cp["c_out_pad"]) for w, b in zip(cp["edge5"], cp["edge5_bias"])]
            ci_pad = cp["edge5"][0].shape[2]
            cwd = cp["corner_w"].double()
            cp["corner_bias"] = cp["corner_bias"] + torch.cat(
                [cwd[k * cp["c_out_pad"]:(k + 1) * cp["c_out_pad"], k * ci_pad:(k + 1) * ci_pad] @ mu["c22"][:ci_pad].double() for k in range(4)]).float()
            cp["post"] = (ones(cp["c_out_pad"]), -mu["comp"])
            cp["out_shift"] = mu["comp"]
            if cp["tile"] == L.TILE_STEM_256x256:
                # relu(a) - mu = max(a - mu, -mu): -mu in the bias and as the ReLU's per-channel floor (VNQA_CONV_RELU_FLOOR) stores the
                # shifted output with one rounding through the tile's ordinary epilogue (the fp32 epilogue, TAG 6, costs +0.14 ms here)
                cp["bias_floor"] = (cp["bias"] - mu["comp"], (-mu["comp"]).contiguous())
        else:
            od[0]["bias"] = od[0]["bias"] + corr(od[0]["_wsum"], mu["c22"], od[0]["c_out_pad"])
            od[1]["post"] = (ones(od[1]["c_out_pad"]), -mu["comp"])
            od[1]["out_shift"] = mu["comp"]
        # conv21: reads comp; its own output (no ReLU / pool before the store) shifted through the bias
        od[2]["bias"] = od[2]["bias"] + corr(od[2]["_wsum"], mu["comp"], od[2]["c_out_pad"]) - mu["od21"]
        od[2]["out_shift"] = mu["od21"]
        # conv22: reads od21
        od[3]["bias"] = od[3]["bias"] + corr(od[3]["_wsum"], mu["od21"], od[3]["c_out_pad"])
        # conv22's and conv31's outputs where they are PLAIN tensors (precision 'fp16' / 'bf16', 'fp16h' at split depth < 3 / < 2; a split
        # tensor's lo half already carries what a shift would save): conv22 through its post affine, conv31 through its bias
        if not int(od[3].get("split_out", 0)) and "od4" in cal:
            mu["od22"] = mu_of("od4", od[3]["c_out"], od[3]["c_out_pad"])
            od[3]["post"] = (ones(od[3]["c_out_pad"]), -mu["od22"])
            od[3]["out_shift"] = mu["od22"]
            od[4]["bias"] = od[4]["bias"] + corr(od[4]["_wsum"], mu["od22"], od[4]["c_out_pad"])
        if not int(od[4].get("split_out", 0)) and "od5" in cal:
            mu["od31"] = mu_of("od5", od[4]["c_out"], od[4]["c_out_pad"])
            od[4]["bias"] = od[4]["bias"] - mu["od31"]
            od[4]["out_shift"] = mu["od31"]
            od[5]["bias"] = od[5]["bias"] + corr(od[5]["_wsum"], mu["od31"], od[5]["c_out_pad"])
        # the features, where the consumer is the trunk's conv_init (it absorbs the mean in its bias: ops.FilmTrunkHeadFn)
        if self.trunk_features and "feat" in cal and not od[5]["pool"]:
            mu["feat"] = mu_of("feat", od[5]["c_out"], od[5]["c_out_pad"])
            od[5]["post"] = (ones(od[5]["c_out_pad"]), -mu["feat"])
            od[5]["out_shift"] = mu["feat"]
            od[5]["split_out"] = 0
            # precision 'fp16h': written TWICE, [x' | x'] — conv_init then runs as x' w_hi + x' w_lo, a plain conv over 2 C channels against
            # split weights (its remaining weight rounding, ~0.1e-6 of the 0.22e-6 left, for one more product on a 14 x 14 layer)
            od[5]["twin_out"] = bool(self.hyb and FEATURE_TWIN)
            od[5]["out_shift2"] = torch.cat([mu["feat"], mu["feat"]])          # (the twin buffer's halo: -mean in both halves)
            self.feature_shift = mu["feat"]
        self.shift = mu
        self._fused_first_shift = v0["tile"] is None

    def packs_checksum(self):
        """sha256 over the bytes of packed_tensors() (in order): written into checkpoints (Trainer.extra_state_dict) so that a stem
        rebuilt from the checkpoint's calibration — another GPU / ROCm / torch build may flip a rounding tie — can be
        CHECKED against the 16-bit weights the model was trained behind."""
        import hashlib
        h = hashlib.sha256()
        for t in self.packed_tensors():
            c = t.detach().contiguous()
            h.update(c.view(torch.uint8).cpu().numpy().tobytes())
        return h.hexdigest()

    def _round(self, w, key, dtype):
        """The frozen weights `w` (fp32, BatchNorm scale folded) as the values the 16-bit kernels multiply with: second-order rounded
        when the calibration pass left the layer's patch moments, else rounded coherently against the mean input, else unchanged
        (the pack kernel rounds to nearest)."""
        if key is None or self.calib is None:
            return w
        if self._H is not None and key in self._H and self._H[key].shape[0] == w[0].numel():
            q = second_order_round(w, self._H[key], dtype)
            if q is not None:
                return q
        mkey = "od0" if key == "od0_5x5" else key
        return coherent_round(w, self.calib[mkey], dtype) if mkey in self.calib else w

    def _layer(self, conv, bn=None, relu=False, pool=False, m=None):
        w = conv.weight.detach().float()
        b = conv.bias.detach().float()
        c_out, c_in = w.shape[0], w.shape[1]
        c_out_pad, c_in_pad = L.round_up(c_out, 64), L.round_up(c_in, 64)
        scale = None
        if bn is not None:
            scale, shift = _fold_bn(bn)
            b = b * scale + shift
        half = L.is_half(self.cdt)      # 16-bit storage (bf16 or, in the fp16 build, fp16): the MFMA fast path
        w32, scale32 = w, scale
        if m is not None and half:
            # (the BN scale folded first: the values the kernel multiplies with are the ones rounded)
            w = self._round(w if scale is None else w * scale.view(-1, 1, 1, 1), m, self.cdt)
            scale = None
        if half and c_in_pad == 64:
            tile = None                      # conv_c64 direct kernel (row layout, LDS-resident weights)
        elif half:
            # conv2_2 (C_out = 128): the 512x128 tile (id 15) has the 256x256 kernel's 128x64 wave tiles and MFMA work per K-step;
            # 1.45 -> 1.08 ms alone, +2 % end to end against the 256x128 tile (128x128, two workgroups per CU, is 20 % faster for
            # conv2_2 ALONE and costs 10 % end to end when the trunk co-runs: finer interleaving of the two streams hurts both)
            tile = L.TILE_STEM_256x256 if c_out_pad >= 256 else (15 if c_out_pad > 64 else L.TILE_256x64)
            # conv11 layer by layer (C_in = 128: only 18 K-steps, and a 964 MB output to store): the 16-wave shape of the same
            # tile keeps more store / DMA issue slots busy around its short main loop (+10 % on this layer)
            if c_out_pad >= 256 and c_in_pad == 128:
                tile = L.TILE_256x256_W16
        else:
            tile = L.TILE_128x64 if c_out_pad <= 64 else L.TILE_128x128
        if tile is None or tile == L.TILE_256x256_W16:
            wt = K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        else:   # frozen weights: pre-tiled once into the exact LDS images the igemm DMA consumes
            wt = K.pack_conv_weight_tiled(w, self.cdt, tile, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        ly = dict(wt=wt, bias=K.pad_vec(b, c_out_pad), relu=relu, pool=pool, post=None,
                  c_out=c_out, c_in=c_in, c_out_pad=c_out_pad, tile=tile)
        if half and m is not None:
            # sum over the taps of the 16-bit weights the kernel multiplies with (float64 [c_out, c_in]): what a per-channel constant of
            # the INPUT contributes to every output — the bias correction of mean-shifted storage (_setup_mean_shift); construction only
            ly["_wsum"] = w.to(self.cdt).double().sum((2, 3))
        # short-K layers of the VGG front (conv1_2 / conv2_1 / conv2_2 shapes): weights-stationary-in-registers direct conv
        # (csrc/conv_wreg.hip) when the run-time geometry has whole tiles; it reads the K-major row pack
        if half and relu and (c_in_pad, c_out_pad, bool(pool)) in ((64, 64, True), (64, 128, False), (128, 128, True)):
            ly["wt_rows"] = wt if tile is None else K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad,
                                                                       c_in_pad=c_in_pad)
        # wide 3x3 layers (conv21 .. conv32): patch-stationary kernel (csrc/conv_ps.hip, K-major weights) when the run-time
        # geometry qualifies (vnqa_conv_ps_supported); the implicit-GEMM tile above stays as the fallback
        if half and tile == L.TILE_STEM_256x256 and w.shape[2] == 3:
            ly["wt_ps"] = K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
            if self.hyb:      # the exact (BatchNorm-folded) weights, for the layers that run with split weights
                ly["wt32ps"] = K.pack_conv_weight(w32, torch.float32, out_scale=scale32, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        return ly

    def _compose_pair(self, c1, c2, bn):
        """Two stacked linear convs with frozen weights as ONE conv.
          y2 = s*(W2 * (W1 * x + b1) + b2) + t        (* = 3x3 'same' conv, s/t = folded eval BatchNorm)
             = Wc * x + bc  - R(x)                    with Wc = (s W2) (*) W1 (5x5), bc = s b2 + t + sum_taps(s W2) b1
        R is non-zero on the 1-pixel image border only: conv2 must see ZEROS outside the image, not conv1 evaluated
        there.  With Y1[q] = b1 + (W1 * x)[q] at the outside-ring positions q of the (H+2)x(W+2) grid,
          R[p] = sum_{taps d: p+d outside} (s W2)[d] Y1[p+d]
        i.e. one small GEMM for Y1 (conv11 at the ring positions) and four edge products (top / bottom / left / right, K = 3 c_mid);
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
        half = L.is_half(self.cdt)
        tile = L.TILE_STEM_256x256 if (half and co_pad >= 256) else (L.TILE_AUTO if half else L.TILE_128x128)
        wcf = wc.float().contiguous().to(dev)
        if self.calib is not None and half:
            wcf = self._round(wcf, "od0_5x5" if (self._H is not None and "od0_5x5" in self._H) else "od0", self.cdt).contiguous()
        if tile == L.TILE_STEM_256x256:
            wt = K.pack_conv_weight_tiled(wcf, self.cdt, tile, c_out_pad=co_pad, c_in_pad=ci_pad)
        else:
            wt = K.pack_conv_weight(wcf, self.cdt, c_out_pad=co_pad, c_in_pad=ci_pad)
        # (K-major copy for the patch-stationary tiles: the same 16-bit values)
        wt_ps = K.pack_conv_weight(wcf, self.cdt, c_out_pad=co_pad, c_in_pad=ci_pad) if (COMPOSED_PS and tile == L.TILE_STEM_256x256) else None
        # ring operand: W1 K-major [cm_pad][9*ci_pad]; edge operands: (s W2) slices [co_pad][3*cm_pad]
        w1m = K.pack_conv_weight(w1.float().contiguous().to(dev), self.cdt, c_out_pad=cm_pad, c_in_pad=ci_pad).view(cm_pad, -1)

        def edge(sel):      # sel: [co,cm,3] -> [co_pad, 3*cm_pad] (slot-major, channels fastest)
            e = torch.zeros(co_pad, 3, cm_pad, dtype=torch.float64)
            e[:co, :, :cm] = sel.permute(0, 2, 1)
            return e.view(co_pad, -1).to(dev).to(self.cdt).contiguous()
        edges = dict(top=edge(w2[:, :, 0, :]), bottom=edge(w2[:, :, 2, :]), left=edge(w2[:, :, :, 0]), right=edge(w2[:, :, :, 2]))
        wsum = wcf.to(self.cdt).double().sum((2, 3)) if half else None
        # ... and the SAME border correction with conv11 composed into conv12's edge taps (round 6): a ring position one pixel outside the
        # image sees the image through ONE kernel row / column of conv11 only, so edge e's correction of border pixel j is a 1x5 / 5x1 conv
        # over the border row / column, W_e[t] = sum_{a + b = t} (s W2)[edge tap a] W1[facing tap b] (128 -> 512, K = 5 c_in), bias
        # sum_a (s W2)[a] b1 — a quarter of the two-step form's FLOPs.  The outside CORNER position is adjacent to the corner pixel only and
        # both of that pixel's edges count it: its term (a 1x1 conv of the corner pixel) is subtracted (ring_assemble(corner=...)).
        w2e = {0: [w2[:, :, 0, a] for a in range(3)], 1: [w2[:, :, 2, a] for a in range(3)],
               2: [w2[:, :, a, 0] for a in range(3)], 3: [w2[:, :, a, 2] for a in range(3)]}
        w1f = {0: [w1[:, :, 2, b] for b in range(3)], 1: [w1[:, :, 0, b] for b in range(3)],
               2: [w1[:, :, b, 2] for b in range(3)], 3: [w1[:, :, b, 0] for b in range(3)]}
        edge5, edge5_bias = [], []
        for e in range(4):
            we = torch.zeros(co_pad, 5, ci_pad, dtype=torch.float64)
            for a in range(3):
                for b in range(3):
                    we[:co, a + b, :ci] += w2e[e][a] @ w1f[e][b]
            edge5.append(we.to(dev).to(self.cdt).contiguous())
            edge5_bias.append(K.pad_vec(sum(w2e[e][a] @ b1 for a in range(3)).float().to(dev), co_pad))
        # corners (top-left, top-right, bottom-left, bottom-right): (s W2)[corner tap] (b1 + W1[facing corner tap] x[corner pixel]) as ONE
        # GEMM over the four corner pixels side by side against a block-diagonal weight matrix
        ctap = [((0, 0), (2, 2)), ((0, 2), (2, 0)), ((2, 0), (0, 2)), ((2, 2), (0, 0))]
        cw = torch.zeros(4 * co_pad, 4 * ci_pad, dtype=torch.float64)
        cb = torch.zeros(4 * co_pad, dtype=torch.float64)
        for k, ((r2, s2), (r1, s1)) in enumerate(ctap):
            cw[k * co_pad:k * co_pad + co, k * ci_pad:k * ci_pad + ci] = w2[:, :, r2, s2] @ w1[:, :, r1, s1]
            cb[k * co_pad:k * co_pad + co] = w2[:, :, r2, s2] @ b1
        return dict(wt=wt, wt_ps=wt_ps, bias=K.pad_vec(bc.float().to(dev), co_pad), b1=K.pad_vec(b1.float().to(dev), cm_pad), w1m=w1m, edges=edges, _wsum=wsum,
                    edge5=edge5, edge5_bias=edge5_bias, corner_w=cw.to(dev).to(self.cdt).contiguous(), corner_bias=cb.float().to(dev),
                    w1_edges=K.ring_edge_weights(w1m.view(cm_pad, 9, ci_pad)),
                    c_in=ci, c_out=co, c_out_pad=co_pad, c_mid_pad=cm_pad, tile=tile, taps=25)

    def _run_composed(self, x, key):
        """x: halo-2 padded NHWC [n, H+4, W+4, ci_pad] -> relu/pool'ed output of the composed pair (halo 1).
        Border correction: conv11 at the ring positions straight from the halo-2 image (four 3-tap launches), written into a zero-separated ring layout,
        and the four edge products as 1x3 convs along its rows — both implicit GEMMs (no im2col matrix, no gathered edge operands:
        147 + 4 x 48 MB less written and read back per 280-frame pass than the round-1 form)."""
        cp = self.composed
        n, hp, wp, ci_pad = x.shape
        H, W = hp - 4, wp - 4
        cm = cp["c_mid_pad"]
        R = 2 * (W + 2) + 2 * H
        if RING_COMPOSED_EDGES and "edge5" in cp and cp["edge5"][0].shape[2] == ci_pad:
            # round 6: four composed 1x5 / 5x1 edge convs straight from the halo-2 image + the corner term (see _compose_pair)
            part = [K.border_edge_conv(x, cp["edge5"][e], cp["edge5_bias"][e], H, W, e) for e in range(4)]
            corners = torch.cat([x[:, 2, 2], x[:, 2, W + 1], x[:, H + 1, 2], x[:, H + 1, W + 1]], dim=1)      # [n, 4 ci_pad]
            cfix = K.gemm_nt(corners, cp["corner_w"], bias=cp["corner_bias"], split_k=False)
            ring = K.ring_assemble(part[0], part[1], part[2], part[3], n, H, W, corner=cfix)
        else:
            # (edge by edge with the three taps that can see the image: a third of the one-launch form's K, whose other products are against
            # the zero halo; RING_EDGE_LAUNCHES = False runs that form: the same bits)
            y1p = self._buf(key + ("y1p", H, W), (n, R + 4, cm))
            if RING_EDGE_LAUNCHES:
                K.conv2d_ring_edges(x, cp["w1_edges"], cp.get("b1_edges", cp["b1"]), H, W, y1p)
            else:
                K.conv2d_ring(x, cp["w1m"].view(cm, 9, ci_pad), cp["b1"], H, W, out_padded=y1p)
            part = [K.ring_edge_conv(y1p, cp["edges"][name], H, W, e) for e, name in enumerate(("top", "bottom", "left", "right"))]
            ring = K.ring_assemble(part[0], part[1], part[2], part[3], n, H, W)
        ho, wo = H // 2, W // 2
        # precision 'fp16h' at split depth 5: the pair's output as a split tensor (the igemm tile's fp32 dual epilogue), conv21 reads it
        dual = self.split_segs if (self.split_depth >= 5 and cp["tile"] == L.TILE_STEM_256x256 and "wt_split" in self.layers_od[2]) else 0
        out = self._buf(key + (ho, wo) + (("split",) if dual else ()), (n, ho + 2, wo + 2, max(dual, 1) * cp["c_out_pad"]),
                        halo=(1, cp.get("out_shift")))
        post = cp.get("post")      # mean-shifted storage: -mu after ReLU / pool, in fp32 before the ONE rounding (the tile's fp32 epilogue)
        floor = cp.get("bias_floor") if not dual else None
        # 2-D pixel tiles (patch-stationary kernel, 5x5 instantiation) where they serve the geometry and the epilogue is the plain one
        ps = COMPOSED_PS and cp.get("wt_ps") is not None and not dual and (floor is not None or not post)
        if ps:
            ok = cp.setdefault("_ps_ok", {})
            if (n, H, W) not in ok:
                ok[(n, H, W)] = K.conv_ps_supported(n, H, W, ci_pad, cp["c_out_pad"], 25, True)
            ps = ok[(n, H, W)]
        timed = self.timing is not None and cp["tile"] == L.TILE_STEM_256x256
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        # every XCD computes ONE cout half of the composed conv (its L2 then holds 1.65 instead of 3.3 MB of weights): fabric-side reads
        # 1 022 -> 831 MB per launch on the igemm tile (profiles/r04_pmc_traffic*.json), 1 159 -> 953 MB on the 2-D tiles (r06_composed_ps_ab.txt)
        ps_flags = L.CONV_XCD_SPLIT_N if COMPOSED_PS_XCD else 0
        if floor is not None:
            y = K.conv2d_igemm(x, cp["wt_ps"] if ps else cp["wt"], bias=floor[0], relu=True, pool2=True, x_halo=2, y_halo=1, out=out,
                               tile=L.TILE_STEM_PS_224x256 if ps else cp["tile"], border_sub=ring, desc_flags=ps_flags if ps else L.CONV_XCD_SPLIT_N,
                               relu_floor=floor[1])
        elif ps:
            y = K.conv2d_igemm(x, cp["wt_ps"], bias=cp["bias"], relu=True, pool2=True, x_halo=2, y_halo=1, out=out, tile=L.TILE_STEM_PS_224x256,
                               border_sub=ring, desc_flags=ps_flags)
        else:
            y = K.conv2d_igemm(x, cp["wt"], bias=cp["bias"], relu=True, pool2=True, x_halo=2, y_halo=1, out=out, tile=cp["tile"],
                               border_sub=ring, desc_flags=L.CONV_XCD_SPLIT_N if L.is_half(self.cdt) else 0, dual_out=dual,
                               post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                               f32_epilogue=bool(post) and not dual and cp["tile"] == L.TILE_STEM_256x256)
        if timed:
            ev1.record()
            self.timing.append((ev0, ev1, 2.0 * n * H * W * cp["c_in"] * cp["c_out"] * 25, ("conv_ps_kernel<%d,5x5>" % (28 if W % 28 == 0 else 14)) if ps else "conv_igemm_kernel"))
        if self._tap is not None:
            self._tap[key] = y
        return y

    def _c64_sched(self):
        """The two schedule words of the fused conv1 kernel's dynamic tile schedule (its persistent workgroups draw tiles from a
        device counter: beside the trunk's forward pass the fixed-stride launch took 1.89 ms against 1.08 alone): one pair per stream
        this plan runs on (a launch leaves them zero; two launches of one plan never overlap on different streams)."""
        key = ("c64sched", torch.cuda.current_stream().cuda_stream)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.zeros(2, dtype=torch.int32, device="cuda")
        return t

    def _buf(self, key, shape, dtype=None, halo=None):
        """Persistent zero-halo activation buffer, grown (never shrunk) along the image axis.  halo = (width, mu): a MEAN-SHIFTED
        tensor's buffer — its halo ring holds -mu[c] in the first len(mu) channels (what the zero padding becomes; further channel
        segments — the lo half of a split tensor — stay zero), written once here: no kernel ever writes a stem buffer's halo."""
        dtype = self.cdt if dtype is None else dtype
        key = key + (str(dtype),) if dtype != self.cdt else key
        cap = self._bufs.get(key)
        if cap is None or cap.shape[0] < shape[0] or tuple(cap.shape[1:]) != tuple(shape[1:]):
            cap = torch.zeros(shape, dtype=dtype, device="cuda")
            if halo is not None and halo[1] is not None and float(halo[1].abs().max()) > 0:
                hw, mu = int(halo[0]), halo[1]
                v = (-mu).to(dtype)
                c = v.numel()
                cap[:, :hw, :, :c] = v
                cap[:, -hw:, :, :c] = v
                cap[:, :, :hw, :c] = v
                cap[:, :, -hw:, :c] = v
            self._bufs[key] = cap
        return cap[:shape[0]]

    def _run(self, x, layers, tag, last_slot=0, first_index=0):
        for i, ly in enumerate(layers, first_index):
            n, hp, wp, _ = x.shape
            h, w = hp - 2, wp - 2
            ho, wo = (h // 2, w // 2) if ly["pool"] else (h, w)
            yh = ly.get("y_halo", 1)
            last = not (i + 1 < len(layers) + first_index)
            key = (tag, i, ho, wo) if not last else (tag, i, ho, wo, last_slot)
            # split tensors between layers (split_depth >= 2) / mean-shifted outputs / twin features: see the class docstring
            split_rd = "wt_split" in ly and x.shape[-1] == ly["wt_split"].shape[2]
            split_wr = int(ly.get("split_out", 0))
            x_segs = x.shape[-1] // ly["c_out_pad"] if split_rd else 1      # (c_in == c_out on the split-reading layers)
            if split_wr and not (yh == 1 and (split_rd or "wt_split" not in ly)):      # (a split-reading layer handed a plain tensor: the chain is off)
                split_wr = 0
            twin = bool(ly.get("twin_out")) and not split_wr
            osh = ly.get("out_shift")
            out = self._buf(key + (("split",) if split_wr else ()) + (("twin",) if twin else ()),
                            (n, ho + 2 * yh, wo + 2 * yh, (2 if twin else max(split_wr, 1)) * ly["c_out_pad"]),
                            halo=(yh, ly.get("out_shift2") if twin else osh))
            post = ly["post"]
            f32e = (2 if twin else 1) if (post is not None and osh is not None) else 0
            ps, pt = (post[0], post[1]) if post else (None, None)
            tile = ly["tile"]
            timed = self.timing is not None and tile == L.TILE_STEM_256x256
            kname = "conv_igemm_kernel"
            if timed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if split_rd or split_wr:
                # the patch-stationary kernel where it serves the geometry, else (the 10 x 13 maps of the reference's 160 x 208 frames:
                # eval/utils.py:24-25) the 256x256 implicit-GEMM tile — plain or with ITS dual epilogue: the same values either way
                on_ps = yh == 1 and K.conv_ps_supported(n, h, w, x.shape[-1], ly["c_out_pad"], 9, ly["pool"])
                if on_ps:
                    kname = "conv_ps_kernel<%d>" % (28 if w % 28 == 0 else 14)
                x = K.conv2d_igemm(x, ly["wt_split"] if split_rd else ly["wt_ps"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=ps, post_shift=pt, out=out, tile=L.TILE_STEM_PS_224x256 if on_ps else L.TILE_STEM_256x256,
                                   y_halo=yh, dual_out=split_wr, f32_epilogue=0 if split_wr else f32e)
            elif "wt_rows" in ly and K.conv2d_wreg_supported(x, ly["wt_rows"], pool2=ly["pool"], y_halo=yh):
                x = K.conv2d_wreg(x, ly["wt_rows"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"], post_scale=ps, post_shift=pt,
                                  out=out, y_halo=yh, reserve_cus=self.reserve_cus)
            elif "wt_ps" in ly and yh == 1 and K.conv_ps_supported(n, h, w, x.shape[-1], ly["c_out_pad"], 9, ly["pool"]):
                kname = "conv_ps_kernel<%d>" % (28 if w % 28 == 0 else 14)      # (one entry per kernel SYMBOL, as rocprofv3 lists them)
                # (a mean-shifted output with ReLU / pool before the shift: the fp32 epilogue, ONE rounding after the affine)
                x = K.conv2d_igemm(x, ly["wt_ps"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"], post_scale=ps, post_shift=pt,
                                   out=out, tile=L.TILE_STEM_PS_224x256, y_halo=yh, f32_epilogue=f32e)
            elif tile is None:
                # C_in = 64 layers (conv1_2, conv2_1): persistent direct conv with LDS-resident weights
                x = K.conv2d_c64(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"], post_scale=ps, post_shift=pt, out=out,
                                 reserve_cus=self.reserve_cus)
            else:
                x = K.conv2d_igemm(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"], post_scale=ps, post_shift=pt,
                                   out=out, tile=tile, y_halo=yh,
                                   f32_epilogue=f32e if (yh == 1 and tile == L.TILE_STEM_256x256 and ly["c_out_pad"] % 8 == 0) else 0)
            if timed:
                ev1.record()
                self.timing.append((ev0, ev1, 2.0 * n * h * w * ly["c_in"] * ly["c_out"] * 9 * (x_segs if split_rd else 1), kname))
            if self._tap is not None:
                self._tap[(tag, i)] = x
        return x

    # ---- fused fast path: clip -> packed native features ------------------------------------
    @torch.no_grad()
    def forward_clip(self, clip, img_of, n_img, slot=0):
        """clip fp32 [B,3,H,W,T] on the GPU — or uint8 raw pixels k, meaning k / 255 exactly as the reference's loader forms it
        (eval/dataset.py:91; VNQADataset(uint8_video=True)) —; img_of int32 [B*T] (image index or -1).
        Returns padded NHWC [n_img, H/16+2, W/16+2, feature_segs * Cpad] in the compute dtype — plain features, or (the FiLM trunk's form,
        see the class docstring) the mean-shifted features `x' = x - feature_shift` once (feature_segs 1: 'fp16' / 'bf16') or twice
        ([x' | x'], feature_segs 2: 'fp16h'), or a split tensor [hi | lo | hi] (feature_segs 3: calibrations without feature means);
        plain_features() turns any of them into the plain fp32 tensor.
        `slot` selects one of several OUTPUT buffers (the intermediates are shared), so that the
        features of step i stay alive for its backward while step i+1's stem already runs."""
        assert self.vgg is not None and self.objdet is not None
        B, _, H, W, T = clip.shape
        ly = self.layers_vgg[0]
        if L.is_half(self.cdt) and ly["tile"] is None:
            # conv1_1 evaluated inside the conv1_2 kernel from a 4-channel 16-bit image list: its 64-channel output
            # (1.8 GB at 280 x 224 x 224) never goes to HBM
            mu_clip = self.shift.get("clip") if getattr(self, "_fused_first_shift", False) else None
            img4 = self._buf(("img4", H, W), (n_img, H + 4, W + 4, 4), halo=(2, mu_clip[:3] if mu_clip is not None else None))
            K.clip_to_nhwc4(clip, img_of, n_img, out=img4, shift=mu_clip)
            ho, wo = (H // 2, W // 2) if ly["pool"] else (H, W)
            out = self._buf(("vgg", 0, ho, wo), (n_img, ho + 2, wo + 2, ly["c_out_pad"]), halo=(1, ly.get("out_shift")))
            post = ly["post"]
            mid = self.shift.get("c11") if getattr(self, "_fused_first_shift", False) else None
            K.conv_first_c64(img4, self.first[0], self._b1_pair if mid is not None else self.first[1], ly["wt"], bias=ly["bias"],
                             relu=ly["relu"], pool2=ly["pool"],
                             post_scale=post[0] if post else None, post_shift=post[1] if post else None, out=out,
                             reserve_cus=self.reserve_cus, sched=self._c64_sched(), mid_shift=mid)
            x = self._run(out, self.layers_vgg[1:], "vgg", first_index=1)
        else:
            if clip.dtype == torch.uint8:        # raw pixels: the un-fused first conv reads the fp32 clip
                clip = K.expand_u8_clip(clip)
            a = self._buf(("first", H, W), (n_img, H + 2, W + 2, 64))
            K.conv_first(clip, self.first[0], self.first[1], img_of, n_img, self.cdt, out=a)
            if self._tap is not None:
                self._tap["first"] = a
            x = self._run(a, self.layers_vgg, "vgg")
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
