"""ObjDetectCNN — parameter container + HIP inference path for the frozen stem's back half
(drop-in for models/obj_detector.py of the reference)."""
import torch.nn as nn

from ..stem import FrozenStem, VGGFront  # noqa: F401  (re-exported for convenience)
from .common import reference_init_

# (attribute, kind, in, out): conv = 3x3 pad 1, bn = BatchNorm2d, pool = 2x2/2 max-pool; `F` = num_filters.
# Attribute names and shapes are the reference's (obj_detector.py:22-41) so that obj_detect.pt loads unchanged.
_TRUNK = (("bn_input", "bn", 128, None),
          ("conv11", "conv", 128, "F"), ("conv12", "conv", "F", "F"), ("bn1", "bn", "F", None), ("pool1", "pool", None, None),
          ("conv21", "conv", "F", "F"), ("conv22", "conv", "F", "F"), ("bn2", "bn", "F", None), ("pool2", "pool", None, None),
          ("conv31", "conv", "F", "F"), ("conv32", "conv", "F", "F"), ("bn3", "bn", "F", None), ("pool3", "pool", None, None))


class ObjDetectCNN(nn.Module):
    """Same constructor as the reference (models/obj_detector.py:11-17).  forward() implements the path the
    video-QA models use — eval mode with pretrained_features=True (eval/utils.py:43-50): the conv stack of
    obj_detector.py:69-86 on the MFMA igemm with eval-mode BatchNorm folded into the convolutions."""

    def __init__(self, nb_classes, num_filters=128, tail_hidden_dim=256, tail_dropout_p=0.5,
                 logits=False, pretrained_features=False, *, precision='fp16h'):
        super(ObjDetectCNN, self).__init__()
        self.logits, self.pretrained_features, self.precision = logits, pretrained_features, precision
        width = lambda v: num_filters if v == "F" else v
        for name, kind, cin, cout in _TRUNK:
            if kind == "conv":
                layer = nn.Conv2d(width(cin), width(cout), kernel_size=3, padding=1)
            elif kind == "bn":
                layer = nn.BatchNorm2d(width(cin))
            else:
                layer = nn.MaxPool2d(kernel_size=2, stride=2)
            setattr(self, name, layer)
        # classifier tail of the detector itself (unused on the video-QA path, kept for checkpoint parity)
        self.fc_tail1 = nn.Linear(num_filters * 6 * 5, tail_hidden_dim)
        self.bn_tail1 = nn.BatchNorm1d(tail_hidden_dim)
        self.fc_tail2 = nn.Linear(tail_hidden_dim, nb_classes)
        self.dropout = nn.Dropout(p=tail_dropout_p)
        self.relu = nn.ReLU(inplace=True)
        self.apply(reference_init_)
        self._plan = None

    def invalidate(self):
        """Drop the folded/packed weight plan (call after changing weights in place)."""
        self._plan = None

    def load_state_dict(self, *args, **kwargs):
        result = super(ObjDetectCNN, self).load_state_dict(*args, **kwargs)
        self._plan = None
        return result

    def forward(self, inputs):
        """inputs: fp32 [N,128,H,W] (output of the VGG front) -> fp32 [N,num_filters,H/4,W/4]."""
        if self.training or not self.pretrained_features:
            raise NotImplementedError(
                "the MI355X path implements ObjDetectCNN as the FROZEN stem (eval mode, "
                "pretrained_features=True; eval/utils.py:43-50); detector training is out of scope")
        if self._plan is None:
            self._plan = FrozenStem(None, self, "fp16" if self.precision == "fp16h" else self.precision)
        return self._plan.objdet_nchw(inputs)
