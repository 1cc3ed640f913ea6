"""ObjDetectCNN — drop-in for models/obj_detector.py (frozen stem back half), HIP inference path."""
import torch
import torch.nn as nn

from ..stem import FrozenStem, VGGFront  # noqa: F401  (re-exported for convenience)


class ObjDetectCNN(nn.Module):
    """Same constructor as the reference (models/obj_detector.py:11-17) and the same
    parameter/buffer names, so its checkpoints (`obj_detect.pt`, eval/utils.py:49) load
    unchanged.  forward() implements the path the video-QA models use: eval mode with
    pretrained_features=True (eval/utils.py:43-50) -> the conv stack of :69-86 on the MFMA
    igemm with eval-mode BatchNorm folded into the convolutions."""

    def __init__(self, nb_classes, num_filters=128, tail_hidden_dim=256, tail_dropout_p=0.5,
                 logits=False, pretrained_features=False, *, precision='bf16'):
        super(ObjDetectCNN, self).__init__()
        self.logits = logits
        self.pretrained_features = pretrained_features
        self.precision = precision
        self.bn_input = nn.BatchNorm2d(128)
        self.conv11 = nn.Conv2d(128, num_filters, kernel_size=3, padding=1)
        self.conv12 = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_filters)
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv21 = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1)
        self.conv22 = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(num_filters)
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv31 = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1)
        self.conv32 = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1)
        self.bn3 = nn.BatchNorm2d(num_filters)
        self.pool3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc_tail1 = nn.Linear(num_filters * 6 * 5, tail_hidden_dim)
        self.bn_tail1 = nn.BatchNorm1d(tail_hidden_dim)
        self.fc_tail2 = nn.Linear(tail_hidden_dim, nb_classes)
        self.dropout = nn.Dropout(p=tail_dropout_p)
        self.relu = nn.ReLU(inplace=True)
        for m in self.modules():                                   # obj_detector.py:46-47
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.xavier_uniform_(m.weight.data)
                m.bias.data.fill_(0.0)
        self._plan = None

    def invalidate(self):
        """Call after changing weights (load_state_dict does it automatically)."""
        self._plan = None

    def load_state_dict(self, *a, **k):
        out = super(ObjDetectCNN, self).load_state_dict(*a, **k)
        self._plan = None
        return out

    def forward(self, inputs):
        """inputs: fp32 [N,128,H,W] (output of the VGG front) -> fp32 [N,num_filters,H/4,W/4]."""
        if self.training or not self.pretrained_features:
            raise NotImplementedError(
                "the MI355X path implements ObjDetectCNN as the FROZEN stem (eval mode, "
                "pretrained_features=True; eval/utils.py:43-50); detector training is out of scope")
        if self._plan is None:
            self._plan = FrozenStem(None, self, self.precision)
        return self._plan.objdet_nchw(inputs)
