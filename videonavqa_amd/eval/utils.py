"""Configuration constants and small helpers of the evaluation drivers.

The NAMES and VALUES below are the reference's contract (eval/utils.py:6-25): other code — and user
scripts written against the reference — import them by name, so they are kept; everything else in this
module is this project's own code.
"""
import os

import numpy as np
import torch

# data root and the files/directories the drivers expect below it (eval/utils.py:6-16)
BASE_DIR = '../data/'
_LAYOUT = {
    'QUESTIONS_DIR': 'encoded_questions',
    'VIDEOS_DIR': 'videos',
    'LABELS_FILE': 'labels.json',
    'OBJ_DETECTOR_PATH': 'obj_detect.pt',
    'RAW_QUESTIONS_FILE': 'q_ids.json',
    'SPLIT_FILE': 'split.json',
}
globals().update({name: BASE_DIR + rel for name, rel in _LAYOUT.items()})

# clip / question geometry (eval/utils.py:19-25)
_GEOMETRY = dict(DROP_EVERY_N_FRAMES=4, MAX_ALLOWED_NUM_FRAMES_DROPPING=35, MAX_NUM_VIDEO_FRAMES=400,
                 MAX_Q_LEN=56, NUM_CLASSES=70, VID_HEIGHT=160, VID_WIDTH=208)
globals().update(_GEOMETRY)

use_cuda = torch.cuda.is_available()


def per_class_accuracies(y_target, y_pred, num_classes):
    """Fraction of correctly predicted examples per class; 0 for classes absent from y_target
    (same numbers as eval/utils.py:30-39, computed with two bincounts)."""
    t = np.asarray(y_target).astype(np.int64)
    p = np.asarray(y_pred).astype(np.int64)
    totals = np.bincount(t, minlength=num_classes)[:num_classes].astype(np.float64)
    hits = np.bincount(t[t == p], minlength=num_classes)[:num_classes].astype(np.float64)
    return np.divide(hits, totals, out=np.zeros(num_classes), where=totals > 0)


def get_object_detector(path=None, precision='fp16h', load=True):
    """The frozen ObjDetectCNN of the video-QA pipeline in eval mode — 27 classes, 512 filters,
    1024-wide tail, no dropout, logits, pretrained_features (eval/utils.py:42-51).  `load=False` keeps the
    random initialisation (synthetic benchmarking on a box without obj_detect.pt)."""
    from ..models.obj_detector import ObjDetectCNN
    net = ObjDetectCNN(27, num_filters=512, tail_hidden_dim=1024, tail_dropout_p=0, logits=True,
                       pretrained_features=True, precision=precision)
    if load:
        ckpt = torch.load(path or globals()['OBJ_DETECTOR_PATH'], map_location='cpu')
        net.load_state_dict(ckpt['state_dict'])
    return net.eval()
