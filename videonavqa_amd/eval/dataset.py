"""Data sources with the item contract of the reference's VNQADataset (eval/dataset.py:57-106):

    ({'video': float [3, H, W, 35] in [0, 1], zero past v_len, 'v_len': int,
      'question': int64 [56], zero past q_len,                 'q_len': int [, 'q_id': int]}, label)

`VNQADataset` reads the real dataset (mp4 clips + encoded question .npy files); `SyntheticVNQADataset`
produces seeded random items of the same layout for boxes without the data.
"""
import json
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from . import utils as U


def _read_frames(path):
    """All frames of an mp4 as a list of uint8 [H, W, 3] arrays (OpenCV, BGR like the reference)."""
    import cv2  # optional dependency: only the real-data path needs it (eval/dataset.py:4)
    cap = cv2.VideoCapture(path)
    frames = []
    ok, img = cap.read()
    while ok and len(frames) < U.MAX_NUM_VIDEO_FRAMES:
        frames.append(img)
        ok, img = cap.read()
    cap.release()
    return frames


def subsample_indices(n_frames, every=None, rng=random):
    """One uniformly chosen frame out of every window of `every` consecutive frames
    (eval/dataset.py:81-89): ceil(n / every) indices, increasing."""
    every = every or U.DROP_EVERY_N_FRAMES
    return [rng.randint(lo, min(lo + every, n_frames) - 1) for lo in range(0, n_frames, every)]


def pad_question(tokens, max_len):
    q = torch.zeros(max_len, dtype=torch.long)
    q[:len(tokens)] = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    return q


class VNQADataset(Dataset):
    """Constructor arguments as in the reference (eval/dataset.py:18-27)."""

    def __init__(self, q_dir, v_dir, filenames, labels, q_only=False, v_only=False, max_q_len=None,
                 num_classes=None, q_metadata=False, uint8_video=False):
        """uint8_video (keyword, not upstream): item['video'] is the RAW 8-bit clip [3, H, W, 35] (zero padded) instead of
        float64 pixels / 255 — a quarter of the fp32 bytes over PCIe; the stem turns pixel k into float32(k / 255.0)
        (division in float64: kernels.pixel_lut), bit for bit what `clip / 255.0` + `.float()` gives (dataset.py:91,
        q_and_v_eval.py:92)."""
        self.uint8_video = uint8_video
        if q_only and v_only:
            raise AssertionError("Can't have both question- and video-only modes!")
        for d, what in ((q_dir, "question"), (v_dir, "video")):
            if not os.path.exists(d):
                raise AssertionError("Non-existent %s directory!" % what)
        self.q_dir, self.v_dir = q_dir, v_dir
        self.q_only, self.v_only = q_only, v_only
        self.max_q_len = U.MAX_Q_LEN if max_q_len is None else max_q_len
        self.num_classes = U.NUM_CLASSES if num_classes is None else num_classes
        self.filenames = np.array(filenames)
        self.labels = labels
        self.q_metadata = q_metadata
        self.q_ids = json.load(open(U.RAW_QUESTIONS_FILE)) if q_metadata else None

    def __len__(self):
        return len(self.filenames)

    def _video(self, name):
        frames = _read_frames(os.path.join(self.v_dir, name + '.mp4'))
        keep = subsample_indices(len(frames))[:U.MAX_ALLOWED_NUM_FRAMES_DROPPING]
        if self.uint8_video:
            clip = torch.zeros(3, U.VID_HEIGHT, U.VID_WIDTH, U.MAX_ALLOWED_NUM_FRAMES_DROPPING, dtype=torch.uint8)
            for slot, idx in enumerate(keep):
                clip[..., slot] = torch.from_numpy(frames[idx]).permute(2, 0, 1)
            return clip, len(keep)
        clip = torch.zeros(3, U.VID_HEIGHT, U.VID_WIDTH, U.MAX_ALLOWED_NUM_FRAMES_DROPPING, dtype=torch.float64)
        for slot, idx in enumerate(keep):
            clip[..., slot] = torch.from_numpy(frames[idx]).permute(2, 0, 1).double()
        return clip / 255.0, len(keep)

    def __getitem__(self, index):
        name = self.filenames[index]
        item = {}
        if not self.q_only:
            item['video'], item['v_len'] = self._video(name)
        if not self.v_only:
            tokens = np.load(os.path.join(self.q_dir, name + '.npy'))
            item['question'], item['q_len'] = pad_question(tokens, self.max_q_len), int(tokens.shape[0])
        if self.q_metadata:
            item['q_id'] = self.q_ids[name]
        return item, self.labels[name]

    def get_class_weights(self):
        """Inverse class frequency over this split (eval/dataset.py:112-120)."""
        counts = np.bincount([self.labels[f] for f in self.filenames], minlength=self.num_classes)
        with np.errstate(divide='ignore'):
            return 1.0 / counts[:self.num_classes].astype(np.float64)


class SyntheticVNQADataset(Dataset):
    """Seeded synthetic items with the VNQADataset layout (v_len 3..35 frames, q_len 5..25,
    tokens 1..vocab-1, frames in [0,1) zero-padded past v_len)."""

    def __init__(self, n_items, height=None, width=None, num_frames=None, vocab_size=134, num_classes=None,
                 max_q_len=None, seed=1234, full_length=False):
        self.n = n_items
        self.h = U.VID_HEIGHT if height is None else height
        self.w = U.VID_WIDTH if width is None else width
        self.t = U.MAX_ALLOWED_NUM_FRAMES_DROPPING if num_frames is None else num_frames
        self.vocab = vocab_size
        self.k = U.NUM_CLASSES if num_classes is None else num_classes
        self.lq = U.MAX_Q_LEN if max_q_len is None else max_q_len
        self.seed, self.full = seed, full_length

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        v_len = self.t if self.full else int(torch.randint(3, self.t + 1, (1,), generator=g))
        video = torch.rand(3, self.h, self.w, self.t, generator=g)
        video[:, :, :, v_len:] = 0
        q_len = int(torch.randint(5, 26, (1,), generator=g))
        q = pad_question(torch.randint(1, self.vocab, (q_len,), generator=g).numpy(), self.lq)
        y = int(torch.randint(0, self.k, (1,), generator=g))
        return {'video': video, 'v_len': v_len, 'question': q, 'q_len': q_len}, y
