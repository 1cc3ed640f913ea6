"""VideoOnlyCNN3D — drop-in for models/v_only_cnn3d.py (config 2: 3-D conv HIP kernel bring-up)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .common import compute_dtype


class VideoOnlyCNN3D(nn.Module):
    """C3D-like baseline (v_only_cnn3d.py:11-37).  Same constructor; `fc6_in_features` parameterises the
    hard-coded 7680 (= 128*10*6*1 for the reference's [B,3,160,208,35] clips, where Conv3d sees
    (D,H,W) = (H,W,T); 1152 for the 16x3x112x112 config-2 clips).  The three Conv3d layers (forward,
    dgrad, wgrad) run on the MFMA igemm / wgrad kernels through a 27-tap table over padded NDHWC; the
    BatchNorm3d / MaxPool3d / FC glue of this bring-up rung is stock PyTorch-ROCm."""

    def __init__(self, nb_classes, *, fc6_in_features=7680, precision='bf16'):
        super(VideoOnlyCNN3D, self).__init__()
        self.compute_dtype = compute_dtype(precision)
        self.bn_input = nn.BatchNorm3d(3)
        self.conv1 = nn.Conv3d(3, 64, kernel_size=3, padding=1)
        self.pool1 = nn.MaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2))
        self.bn1 = nn.BatchNorm3d(64)
        self.conv2 = nn.Conv3d(64, 128, kernel_size=3, padding=1)
        self.pool2 = nn.MaxPool3d(kernel_size=(4, 4, 4), stride=(4, 4, 4))
        self.bn2 = nn.BatchNorm3d(128)
        self.conv3a = nn.Conv3d(128, 128, kernel_size=3, padding=1)
        self.pool3 = nn.MaxPool3d(kernel_size=(4, 4, 4), stride=(4, 4, 4))
        self.bn3 = nn.BatchNorm3d(128)
        self.fc6 = nn.Linear(fc6_in_features, 2048)
        self.bn6 = nn.BatchNorm1d(2048)
        self.fc7 = nn.Linear(2048, 128)
        self.bn7 = nn.BatchNorm1d(128)
        self.fc8 = nn.Linear(128, nb_classes)
        self.relu = nn.ReLU(inplace=True)
        for m in self.modules():          # weights_init tests Conv2d, not Conv3d (:41): only Linear layers get xavier
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight.data)
                m.bias.data.fill_(0.0)

    def _conv_block(self, h, conv, pool, bn, need_dx):
        """relu(conv3d) on the HIP kernels -> MaxPool3d -> BatchNorm3d (v_only_cnn3d.py:62-72)."""
        x = ops.ncdhw_to_ndhwc_padded(h, self.compute_dtype)
        y = ops.conv3d(x, conv.weight, conv.bias, relu=True, need_dx=need_dx)
        return bn(pool(ops.ndhwc_padded_to_ncdhw(y, conv.out_channels)))

    def features(self, inputs):
        assert inputs.is_cuda, "the HIP path needs device tensors (no CPU fallback)"
        h = self.bn_input(inputs.float())                                       # :60
        h = self._conv_block(h, self.conv1, self.pool1, self.bn1, need_dx=True)   # bn_input is trainable: needs dgrad
        h = self._conv_block(h, self.conv2, self.pool2, self.bn2, need_dx=True)
        return self._conv_block(h, self.conv3a, self.pool3, self.bn3, need_dx=True)

    def forward(self, inputs):
        """inputs fp32 [B,3,D,H,W] -> logits [B,nb_classes] (v_only_cnn3d.py:59-81)."""
        h = self.features(inputs)
        h = h.reshape(h.shape[0], -1)
        h = self.bn6(F.relu(self.fc6(h)))
        h = self.bn7(F.relu(self.fc7(h)))
        return self.fc8(h)
