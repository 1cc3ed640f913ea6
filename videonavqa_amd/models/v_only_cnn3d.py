"""VideoOnlyCNN3D — C3D-like video-only baseline (config 2: 3-D conv HIP kernel bring-up; drop-in for
models/v_only_cnn3d.py)."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .common import compute_dtype, reference_init_

# conv stage = (conv attr, bn attr, pool attr, C_in, C_out, pool kernel == stride)   (v_only_cnn3d.py:16-26)
_STAGES = (("conv1", "bn1", "pool1", 3, 64, (1, 2, 2)),
           ("conv2", "bn2", "pool2", 64, 128, (4, 4, 4)),
           ("conv3a", "bn3", "pool3", 128, 128, (4, 4, 4)))
# classifier = (fc attr, bn attr or None, out features); fc6's input width is a constructor argument
_HEAD = (("fc6", "bn6", 2048), ("fc7", "bn7", 128), ("fc8", None, None))


class VideoOnlyCNN3D(nn.Module):
    """`VideoOnlyCNN3D(nb_classes)` as upstream; `fc6_in_features` parameterises the hard-coded 7680
    (= 128*10*6*1 for the reference's [B,3,160,208,35] clips, where Conv3d sees (D,H,W) = (H,W,T); 1152 for
    the 16x3x112x112 config-2 clips).  The three Conv3d layers (forward, dgrad, wgrad) run on the MFMA
    igemm / wgrad kernels through a 27-tap table over padded NDHWC; the BatchNorm3d / MaxPool3d / FC glue
    of this bring-up rung is stock PyTorch-ROCm."""

    def __init__(self, nb_classes, *, fc6_in_features=7680, precision='bf16'):
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

    def features(self, inputs):
        """bn_input -> 3 x [relu(conv3d) on the HIP kernels -> MaxPool3d -> BatchNorm3d]  (v_only_cnn3d.py:60-72)."""
        assert inputs.is_cuda, "the HIP path needs device tensors (no CPU fallback)"
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
        h = self.bn6(F.relu(self.fc6(h)))
        h = self.bn7(F.relu(self.fc7(h)))
        return self.fc8(h)
