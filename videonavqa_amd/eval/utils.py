"""Constants and helpers mirroring eval/utils.py of the reference (same names and values)."""
import numpy as np
import torch

BASE_DIR = '../data/'                                   # eval/utils.py:6

# Dir paths (eval/utils.py:9-10)
QUESTIONS_DIR = BASE_DIR + 'encoded_questions'
VIDEOS_DIR = BASE_DIR + 'videos'

# File paths (eval/utils.py:13-16)
LABELS_FILE = BASE_DIR + 'labels.json'
OBJ_DETECTOR_PATH = BASE_DIR + 'obj_detect.pt'
RAW_QUESTIONS_FILE = BASE_DIR + 'q_ids.json'
SPLIT_FILE = BASE_DIR + 'split.json'

# Numeric constants (eval/utils.py:19-25)
DROP_EVERY_N_FRAMES = 4
MAX_ALLOWED_NUM_FRAMES_DROPPING = 35
MAX_NUM_VIDEO_FRAMES = 400
MAX_Q_LEN = 56
NUM_CLASSES = 70
VID_HEIGHT = 160
VID_WIDTH = 208

use_cuda = torch.cuda.is_available()                   # eval/utils.py:27


def per_class_accuracies(y_target, y_pred, num_classes):
    """eval/utils.py:30-39."""
    accs = []
    for i in range(num_classes):
        idxs = np.where(y_target == i)[0]
        total = idxs.size
        hits = np.where(y_pred[idxs] == i)[0].size
        accs.append((float(hits) / float(total)) if total != 0 else 0.0)
    return np.array(accs)


def get_object_detector(path=OBJ_DETECTOR_PATH, precision='bf16', load=True):
    """eval/utils.py:42-51: ObjDetectCNN(27, 512, 1024, 0, logits, pretrained_features) in eval mode.
    `load=False` keeps the random initialisation (synthetic benchmarking without obj_detect.pt)."""
    from ..models.obj_detector import ObjDetectCNN
    model = ObjDetectCNN(nb_classes=27, num_filters=512, tail_hidden_dim=1024, tail_dropout_p=0, logits=True,
                         pretrained_features=True, precision=precision)
    if load:
        model.load_state_dict(torch.load(path, map_location='cpu')['state_dict'])
    model.eval()
    return model
