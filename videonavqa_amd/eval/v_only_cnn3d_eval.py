#!/usr/bin/env python3
"""Counterpart of eval/v_only_cnn3d_eval.py (ladder config 2: VideoOnlyCNN3D, the 3-D conv bring-up): clips ->
VideoOnlyCNN3D (Conv3d layers on the HIP igemm / wgrad kernels) -> CE -> Adam, per-epoch checkpoint
`e{epoch}_{checkpoint_path}` and resume from `--checkpoint_path`.  Flags of v_only_cnn3d_eval.py:21-38;
`--synthetic N` uses seeded clips, `--clip D H W` sets their geometry (default 16 112 112, BASELINE.json config 2;
the reference feeds [B,3,160,208,35] so that Conv3d sees (D,H,W) = (H,W,T))."""
import argparse
import json
import os

import torch
import torch.nn as nn

from . import single_modality as S
from . import utils as U


def build_parser():
    ap = argparse.ArgumentParser()
    yes = lambda v: str(v).lower() in ('1', 'true', 'yes')
    ap.add_argument('--num_classes', type=int, default=70)
    ap.add_argument('--use_class_weights', type=yes, default=False)
    ap.add_argument('--batch_size', type=int, default=8)
    ap.add_argument('--l_rate', type=float, default=1e-4)
    ap.add_argument('--loss_reduction', type=str, choices=['sum', 'mean', 'elementwise_mean'])
    ap.add_argument('--num_epochs', type=int, default=1)
    ap.add_argument('--checkpoint_path', type=str)
    ap.add_argument('--num_workers', type=int, default=4)
    ap.add_argument('--stats_after_every', type=int, default=1000)
    ap.add_argument('--val_only', type=yes, default=False)
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--clip', type=int, nargs=3, default=[16, 112, 112], metavar=('D', 'H', 'W'))
    ap.add_argument('--precision', type=str, choices=['bf16', 'fp16', 'fp16h', 'fp32'], default='fp16h')
    return ap


class _SyntheticClips(torch.utils.data.Dataset):
    def __init__(self, n, dhw, classes, seed):
        self.n, self.dhw, self.k, self.seed = n, tuple(dhw), classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        return {'video': torch.rand((3,) + self.dhw, generator=g)}, int(torch.randint(0, self.k, (1,), generator=g))


def main(argv=None):
    args = build_parser().parse_args(argv)
    from torch.utils.data import DataLoader
    from ..models import VideoOnlyCNN3D
    from .dataset import VNQADataset
    assert torch.cuda.is_available(), "the MI355X path needs a GPU (there is no CPU fallback)"
    dev = torch.device('cuda', 0)
    if args.synthetic > 0:
        train = _SyntheticClips(args.synthetic, args.clip, args.num_classes, 21)
        val = _SyntheticClips(max(args.synthetic // 4, args.batch_size), args.clip, args.num_classes, 22)
        d, h, w = args.clip
    else:
        split, labels = json.load(open(U.SPLIT_FILE)), json.load(open(U.LABELS_FILE))
        train = VNQADataset(q_dir=U.QUESTIONS_DIR, v_dir=U.VIDEOS_DIR, v_only=True, num_classes=args.num_classes,
                            filenames=split['train'], labels=labels)
        val = VNQADataset(q_dir=U.QUESTIONS_DIR, v_dir=U.VIDEOS_DIR, v_only=True, num_classes=args.num_classes,
                          filenames=split['val'], labels=labels)
        d, h, w = U.VID_HEIGHT, U.VID_WIDTH, U.MAX_ALLOWED_NUM_FRAMES_DROPPING
    print('%d train examples, %d validation examples' % (len(train), len(val)))
    loaders = [DataLoader(dataset=ds, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
               for ds in (train, val)]
    fc6_in = 128 * (d // 16) * (h // 32) * (w // 32)           # pools (1,2,2), (4,4,4), (4,4,4): 7680 for 160x208x35
    model = VideoOnlyCNN3D(nb_classes=args.num_classes, fc6_in_features=fc6_in, precision=args.precision).to(dev)
    weights = None
    if args.use_class_weights and hasattr(train, 'get_class_weights'):
        weights = torch.as_tensor(train.get_class_weights(), dtype=torch.float32, device=dev)
    loss_fn = nn.CrossEntropyLoss(weight=weights)              # upstream ignores --loss_reduction here (:170)
    print(model)
    opt = torch.optim.Adam(model.parameters(), lr=args.l_rate)
    start = 0
    if args.checkpoint_path is not None:
        if not os.path.exists(args.checkpoint_path):
            print('=> No checkpoint existent - will save the model here')
        else:
            ck = torch.load(args.checkpoint_path, map_location=dev)
            start = ck['epoch'] + 1
            model.load_state_dict(ck['state_dict'])
            opt.load_state_dict(ck['optimizer'])
            print('==> Restored checkpoint %s (epoch %d)' % (args.checkpoint_path, start))

    def train_step(Xs, ys):
        x, ys = Xs['video'].float().to(dev), ys.to(dev)
        out = model(x)
        loss = loss_fn(out, ys)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss.detach(), out, ys

    def val_step(Xs, ys):
        with torch.no_grad():
            x, ys = Xs['video'].float().to(dev), ys.to(dev)
            out = model(x)
            return loss_fn(out, ys), out, ys

    for epoch in range(start, start + args.num_epochs):
        if not args.val_only:
            model.train()
            t = S.run_epoch(loaders[0], args.batch_size, train_step)
            f1w, f1m = t.f1() if t.n else (0.0, 0.0)
            print('Train Epoch: {}\tAverage loss: {:.6f}\tAccuracy: {}/{}\tF1: w{:.4f}, micro{:.4f}\n'.format(
                epoch, t.loss / max(t.n, 1), t.hit, t.n, f1w, f1m))
            S.save_if(args.checkpoint_path, {'epoch': epoch, 'state_dict': model.state_dict(), 'train_f1w': f1w,
                                             'train_f1micro': f1m, 'optimizer': opt.state_dict()}, 'e%d_' % epoch)
        model.eval()
        v = S.run_epoch(loaders[1], args.batch_size, val_step)
        f1w, f1m = v.f1() if v.n else (0.0, 0.0)
        print('Validation:\tAverage loss: {:.6f}, Accuracy: {}/{}, F1: w{:.4f}, micro{:.4f}\n'.format(
            v.loss / max(v.n, 1), v.hit, v.n, f1w, f1m))


if __name__ == '__main__':
    main()
