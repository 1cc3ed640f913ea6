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
        if self._plan is None:
            self._plan = FrozenStem(self, None, self.precision)
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


def _fold_bn(bn):
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + BN_EPS)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


class FrozenStem(object):
    """Execution plan (packed weights + persistent activation buffers) for the frozen stem."""

    def __init__(self, vgg, objdet, precision='bf16'):
        self.cdt = torch.bfloat16 if precision in ("bf16", torch.bfloat16) else torch.float32
        self.vgg, self.objdet = vgg, objdet
        self.layers_vgg, self.layers_od = [], []
        self.first = None
        self._bufs = {}
        self.timing = None   # bench hook: list collecting (start, end) events around stem-tagged launches
        if vgg is not None:
            f = vgg.features
            dev = f["0"].weight.device
            self.first = (f["0"].weight.detach().float().contiguous(), f["0"].bias.detach().float().contiguous())
            self.layers_vgg = [self._layer(f["2"], relu=True, pool=True),
                               self._layer(f["5"], relu=True, pool=False),
                               self._layer(f["7"], relu=True, pool=True)]
        if objdet is not None:
            od = objdet
            self.bn_input = _fold_bn(od.bn_input)
            self.layers_od = [self._layer(od.conv11), self._layer(od.conv12, bn=od.bn1, relu=True, pool=True),
                              self._layer(od.conv21), self._layer(od.conv22, bn=od.bn2, relu=True, pool=True),
                              self._layer(od.conv31), self._layer(od.conv32, bn=od.bn3, relu=True, pool=False)]
            self.out_channels = od.conv32.out_channels
            if vgg is not None:
                # bn_input becomes the post-affine of the last VGG layer's epilogue
                s, t = self.bn_input
                self.layers_vgg[-1]["post"] = (K.pad_vec(s, 128), K.pad_vec(t, 128))

    def _layer(self, conv, bn=None, relu=False, pool=False):
        w = conv.weight.detach().float()
        b = conv.bias.detach().float()
        c_out, c_in = w.shape[0], w.shape[1]
        c_out_pad, c_in_pad = L.round_up(c_out, 64), L.round_up(c_in, 64)
        scale = None
        if bn is not None:
            scale, shift = _fold_bn(bn)
            b = b * scale + shift
        bf16 = self.cdt == torch.bfloat16
        if bf16 and c_in_pad == 64:
            tile = None                      # conv_c64 direct kernel (row layout, LDS-resident weights)
        elif bf16:
            # 128x128 (2 workgroups/CU) is 20 % faster for conv2_2 ALONE but costs 10 % end to end when the trunk co-runs on
            # the other stream (same-box A/B): finer interleaving of the two streams' workgroups hurts both
            t128 = L.TILE_128x128 if os.environ.get("VNQA_STEM_T128", "0") == "1" else int(os.environ.get("VNQA_STEM_C128_TILE", str(L.TILE_256x128)))
            tile = L.TILE_STEM_256x256 if c_out_pad >= 256 else (t128 if c_out_pad > 64 else L.TILE_256x64)
            # conv11 (C_in = 128: only 18 K-steps, and a 964 MB output to store): the 16-wave shape of the same tile
            # keeps more store / DMA issue slots busy around its short main loop (+10 % on this layer, -2..4 % on the
            # 72-K-step layers, which therefore keep the 8-wave staggered kernel)
            if c_out_pad >= 256 and c_in_pad == 128 and os.environ.get("VNQA_STEM_W16", "1") != "0":
                tile = L.TILE_256x256_W16
        else:
            tile = L.TILE_128x64 if c_out_pad <= 64 else L.TILE_128x128
        if tile is None or tile == L.TILE_256x256_W16 or os.environ.get("VNQA_STEM_TILED", "1") == "0":
            wt = K.pack_conv_weight(w, self.cdt, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        else:   # frozen weights: pre-tiled once into the exact LDS images the igemm DMA consumes
            wt = K.pack_conv_weight_tiled(w, self.cdt, tile, out_scale=scale, c_out_pad=c_out_pad, c_in_pad=c_in_pad)
        return dict(wt=wt, bias=K.pad_vec(b, c_out_pad), relu=relu, pool=pool, post=None,
                    c_out=c_out, c_in=c_in, c_out_pad=c_out_pad, tile=tile)

    def _buf(self, key, shape):
        """Persistent zero-halo activation buffer, grown (never shrunk) along the image axis."""
        cap = self._bufs.get(key)
        if cap is None or cap.shape[0] < shape[0] or tuple(cap.shape[1:]) != tuple(shape[1:]):
            cap = torch.zeros(shape, dtype=self.cdt, device="cuda")
            self._bufs[key] = cap
        return cap[:shape[0]]

    def _run(self, x, layers, tag, last_slot=0, first_index=0):
        for i, ly in enumerate(layers, first_index):
            n, hp, wp, _ = x.shape
            h, w = hp - 2, wp - 2
            ho, wo = (h // 2, w // 2) if ly["pool"] else (h, w)
            key = (tag, i, ho, wo) if i + 1 < len(layers) + first_index else (tag, i, ho, wo, last_slot)
            out = self._buf(key, (n, ho + 2, wo + 2, ly["c_out_pad"]))
            post = ly["post"]
            tile = ly["tile"]
            timed = self.timing is not None and tile == L.TILE_STEM_256x256
            if timed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if tile is None:
                # C_in = 64 layers (conv1_2, conv2_1): persistent direct conv with LDS-resident weights
                x = K.conv2d_c64(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                 post_scale=post[0] if post else None, post_shift=post[1] if post else None, out=out)
            else:
                x = K.conv2d_igemm(x, ly["wt"], bias=ly["bias"], relu=ly["relu"], pool2=ly["pool"],
                                   post_scale=post[0] if post else None, post_shift=post[1] if post else None,
                                   out=out, tile=tile)
            if timed:
                ev1.record()
                self.timing.append((ev0, ev1, 2.0 * n * h * w * ly["c_in"] * ly["c_out"] * 9))
        return x

    # ---- fused fast path: clip -> packed native features ------------------------------------
    @torch.no_grad()
    def forward_clip(self, clip, img_of, n_img, slot=0):
        """clip fp32 [B,3,H,W,T] on the GPU; img_of int32 [B*T] (image index or -1).
        Returns padded NHWC [n_img, H/16+2, W/16+2, Cpad] in the compute dtype.
        `slot` selects one of several OUTPUT buffers (the intermediates are shared), so that the
        features of step i stay alive for its backward while step i+1's stem already runs."""
        assert self.vgg is not None and self.objdet is not None
        B, _, H, W, T = clip.shape
        ly = self.layers_vgg[0]
        if self.cdt == torch.bfloat16 and ly["tile"] is None and os.environ.get("VNQA_FUSE_FIRST", "1") != "0":
            # conv1_1 evaluated inside the conv1_2 kernel from a 4-channel bf16 image list: its 64-channel output
            # (1.8 GB at 280 x 224 x 224) never goes to HBM
            img4 = self._buf(("img4", H, W), (n_img, H + 4, W + 4, 4))
            K.clip_to_nhwc4(clip, img_of, n_img, out=img4)
            ho, wo = (H // 2, W // 2) if ly["pool"] else (H, W)
            out = self._buf(("vgg", 0, ho, wo), (n_img, ho + 2, wo + 2, ly["c_out_pad"]))
            post = ly["post"]
            x = K.conv_first_c64(img4, self.first[0], self.first[1], ly["wt"], bias=ly["bias"], relu=ly["relu"],
                                 pool2=ly["pool"], post_scale=post[0] if post else None,
                                 post_shift=post[1] if post else None, out=out)
            x = self._run(x, self.layers_vgg[1:], "vgg", first_index=1)
        else:
            a = self._buf(("first", H, W), (n_img, H + 2, W + 2, 64))
            K.conv_first(clip, self.first[0], self.first[1], img_of, n_img, self.cdt, out=a)
            x = self._run(a, self.layers_vgg, "vgg")
        return self._run(x, self.layers_od, "od", last_slot=slot)

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
        y = self._run(K.nchw_to_nhwc(xin, self.cdt), self.layers_od, "ods")
        return K.nhwc_to_nchw(y, self.out_channels)
