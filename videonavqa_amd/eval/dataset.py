"""Dataset contract of eval/dataset.py (VNQADataset) + a synthetic source with the same item layout.

Item = ({'video': f64/f32 [3, H, W, 35] in [0,1] zero-padded past v_len, 'v_len': int,
         'question': int64 [56] zero-padded, 'q_len': int}, y)            (eval/dataset.py:57-106)
"""
import json
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import (DROP_EVERY_N_FRAMES, MAX_ALLOWED_NUM_FRAMES_DROPPING, MAX_NUM_VIDEO_FRAMES, MAX_Q_LEN,
                    NUM_CLASSES, RAW_QUESTIONS_FILE, VID_HEIGHT, VID_WIDTH)


class VNQADataset(Dataset):
    """Same constructor and item contract as the reference (eval/dataset.py:13-106).  mp4 decoding needs
    OpenCV exactly as upstream; it is imported lazily so that the synthetic path has no such dependency."""

    def __init__(self, q_dir, v_dir, filenames, labels, q_only=False, v_only=False, max_q_len=MAX_Q_LEN,
                 num_classes=NUM_CLASSES, q_metadata=False):
        assert not (q_only and v_only), "Can't have both question- and video-only modes!"
        self.q_only, self.v_only = q_only, v_only
        self.num_classes, self.max_q_len = num_classes, max_q_len
        assert os.path.exists(q_dir), "Non-existent question directory!"
        assert os.path.exists(v_dir), "Non-existent video directory!"
        self.q_dir, self.v_dir = q_dir, v_dir
        self.filenames = np.array(filenames)
        self.labels = labels
        self.q_metadata = q_metadata
        if self.q_metadata:
            self.q_ids = json.load(open(RAW_QUESTIONS_FILE, 'r'))

    def __len__(self):
        return self.filenames.shape[0]

    def __getitem__(self, index):
        filename = self.filenames[index]
        X = {}
        if not self.q_only:
            import cv2  # same dependency as the reference (eval/dataset.py:4)
            X_vid = np.empty(shape=(3, VID_HEIGHT, VID_WIDTH, MAX_NUM_VIDEO_FRAMES))
            vid = cv2.VideoCapture(os.path.join(self.v_dir, filename + '.mp4'))
            count = 0
            while True:
                ok, image = vid.read()
                if not ok:
                    break
                X_vid[:, :, :, count] = image.transpose(2, 0, 1)
                count += 1
            vid.release()
            X_vid = X_vid[:, :, :, :count]
            vid_len = count
            X_final = np.zeros(shape=(3, VID_HEIGHT, VID_WIDTH, MAX_ALLOWED_NUM_FRAMES_DROPPING))
            count = 0
            for i in range(0, vid_len, DROP_EVERY_N_FRAMES):          # 1-of-4 random frame subsample (:81-89)
                keep = random.randint(i, min(i + DROP_EVERY_N_FRAMES, vid_len) - 1)
                X_final[:, :, :, count] = X_vid[:, :, :, keep]
                count += 1
            X['video'] = torch.from_numpy(X_final) / 255.0
            X['v_len'] = count
        if not self.v_only:
            X_q = torch.from_numpy(np.load(os.path.join(self.q_dir, filename + '.npy')))
            q = torch.LongTensor(np.zeros((self.max_q_len,)))
            q[:X_q.shape[0]] = X_q
            X['question'] = q
            X['q_len'] = X_q.shape[0]
        if self.q_metadata:
            X['q_id'] = self.q_ids[filename]
        return X, self.labels[filename]

    def get_class_weights(self):
        """eval/dataset.py:112-120."""
        classes = np.array([self.labels[f] for f in self.filenames])
        return np.array([(1.0 / float((classes == i).sum())) for i in range(self.num_classes)])


class SyntheticVNQADataset(Dataset):
    """Seeded synthetic items with the VNQADataset layout (v_len 3..35 frames, q_len 5..25,
    tokens 1..vocab-1, frames in [0,1) zero-padded past v_len)."""

    def __init__(self, n_items, height=VID_HEIGHT, width=VID_WIDTH, num_frames=MAX_ALLOWED_NUM_FRAMES_DROPPING,
                 vocab_size=134, num_classes=NUM_CLASSES, max_q_len=MAX_Q_LEN, seed=1234, full_length=False):
        self.n, self.h, self.w, self.t = n_items, height, width, num_frames
        self.vocab, self.k, self.lq, self.seed, self.full = vocab_size, num_classes, max_q_len, seed, full_length

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        v_len = self.t if self.full else int(torch.randint(3, self.t + 1, (1,), generator=g))
        video = torch.rand(3, self.h, self.w, self.t, generator=g)
        video[:, :, :, v_len:] = 0
        q_len = int(torch.randint(5, 26, (1,), generator=g))
        q = torch.zeros(self.lq, dtype=torch.long)
        q[:q_len] = torch.randint(1, self.vocab, (q_len,), generator=g)
        y = int(torch.randint(0, self.k, (1,), generator=g))
        return {'video': video, 'v_len': v_len, 'question': q, 'q_len': q_len}, y
