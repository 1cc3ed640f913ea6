"""VideoOnlyCNN3D — C3D-like video-only baseline (config 2: 3-D conv HIP kernel bring-up; drop-in for
models/v_only_cnn3d.py)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from .. import ops
from .common import compute_dtype, grad_scale_of, reference_init_

# conv stage = (conv attr, bn attr, pool attr, C_in, C_out, pool kernel == stride)   (v_only_cnn3d.py:16-26)
_STAGES = (("conv1", "bn1", "pool1", 3, 64, (1, 2, 2)),
           ("conv2", "bn2", "pool2", 64, 128, (4, 4, 4)),
           ("conv3a", "bn3", "pool3", 128, 128, (4, 4, 4)))
# classifier = (fc attr, bn attr or None, out features); fc6's input width is a constructor argument
_HEAD = (("fc6", "bn6", 2048), ("fc7", "bn7", 128), ("fc8", None, None))


class VideoOnlyCNN3D(nn.Module):
    """`VideoOnlyCNN3D(nb_classes)` as upstream; `fc6_in_features` parameterises the hard-coded 7680
    (= 128*10*6*1 for the reference's [B,3,160,208,35] clips, where Conv3d sees (D,H,W) = (H,W,T); 1152 for
    the 16x3x112x112 config-2 clips).  16-bit precisions with H, W multiples of 16 and pool-divisible extents (config 2) run the
    whole feature extractor as ONE autograd node on csrc/cnn3d.hip (ops.Cnn3dFeaturesFn: conv1 straight from the fp32 clip with
    bn_input / pool1 / bn1's statistics fused, pools with arg-max bytes, BatchNorm written into the next conv's padded input)
    plus the 27-tap igemm / small-channel wgrad kernels, and the classifier on vnqa_sgemm + the same BatchNorm kernels.
    precision='fp32' and other geometries (the reference's 160x208x35 clips: W = 35) take the generic path: convs on the
    igemm / wgrad kernels through padded NDHWC with BatchNorm3d / MaxPool3d on stock PyTorch-ROCm; the classifier is on library kernels in every precision."""

    def __init__(self, nb_classes, *, fc6_in_features=7680, precision='fp16h'):
        super(VideoOnlyCNN3D, self).__init__()
        self.compute_dtype = compute_dtype(precision)
        self.bn_input = nn.BatchNorm3d(3)
        for conv, bn, pool, cin, cout, pk in _STAGES:
            setattr(self, conv, nn.Conv3d(cin, cout, kernel_size=3, padding=1))
            setattr(self, pool, nn.MaxPool3d(kernel_size=pk, stride=pk))
            setattr(self, bn, nn.BatchNorm3d(cout))
        width = fc6_in_features
        for fc, bn, out in _HEAD:
            out = nb_classes if out is None else out
            setattr(self, fc, nn.Linear(width, out))
            if bn is not None:
                setattr(self, bn, nn.BatchNorm1d(out))
            width = out
        self.relu = nn.ReLU(inplace=True)
        self.apply(reference_init_)          # upstream's rule tests Conv2d, not Conv3d (:41): convs keep default init

    def _fast_ok(self, inputs):
        if self.compute_dtype == torch.float32 or getattr(self, "force_generic", False):     # (force_generic: the tests' cross-check switch)
            return False
        N, C, D, H, W = inputs.shape
        if C != 3 or not K.c3d_conv1_supported(N, D, H, W):
            return False
        return D % 4 == 0 and (H // 2) % 4 == 0 and (W // 2) % 4 == 0 and D // 4 >= 4 and H // 8 >= 4 and W // 8 >= 4

    def features(self, inputs):
        """bn_input -> 3 x [relu(conv3d) on the HIP kernels -> MaxPool3d -> BatchNorm3d]  (v_only_cnn3d.py:60-72).
        Returns NCDHW fp32."""
        assert inputs.is_cuda, "the HIP path needs device tensors (no CPU fallback)"
        if self._fast_ok(inputs):
            if self.training:
                with torch.no_grad():
                    for bn in (self.bn_input, self.bn1, self.bn2, self.bn3):
                        bn.num_batches_tracked += 1
            return ops.cnn3d_features(inputs, self.bn_input, self.conv1, self.bn1, self.conv2, self.bn2, self.conv3a, self.bn3,
                                      self.training, self.compute_dtype, grad_scale_of(self.compute_dtype))
        h = self.bn_input(inputs.float())
        for conv_name, bn_name, pool_name, _, cout, _ in _STAGES:
            conv = getattr(self, conv_name)
            x = ops.ncdhw_to_ndhwc_padded(h, self.compute_dtype)
            y = ops.conv3d(x, conv.weight, conv.bias, relu=True, need_dx=True)   # conv1 too: bn_input is trainable
            h = getattr(self, bn_name)(getattr(self, pool_name)(ops.ndhwc_padded_to_ncdhw(y, cout)))
        return h

    def forward(self, inputs):
        """inputs fp32 [B,3,D,H,W] -> logits [B,nb_classes] (v_only_cnn3d.py:59-81)."""
        h = self.features(inputs).flatten(1)
        # classifier on vnqa_sgemm (exact fp32) + the channel-last BatchNorm kernels, in every precision
        for fc, bn in ((self.fc6, self.bn6), (self.fc7, self.bn7)):
            if self.training:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
            h = ops.batch_norm_rows(ops.linear(h, fc.weight, fc.bias, relu=True), bn, self.training)
        return ops.linear(h, self.fc8.weight, self.fc8.bias)
