#!/usr/bin/env python3
"""Counterpart of eval/q_only_eval.py (ladder config 1: question-only plumbing): encoded questions -> sort by length
-> QOnlyLSTM -> weighted CE -> Adam.  Flags and defaults of q_only_eval.py:20-44; `--synthetic N` replaces ../data by
N seeded encoded questions.  The recurrence runs on the persistent HIP LSTM kernel (models/q_only_lstm.py), so a GPU is
required; the reference's 'bow' model (QOnlyBOW) is not part of the hot path and is not built.

Upstream quirks kept: a fresh N(0,1) LSTM state per TRAINING batch (`init_hidden()`, :80-82) but the state carried over
between validation batches; validation every `--stats_after_every` epochs, checkpoint on a new best micro-F1 (upstream
calls an undefined `test()` there, :208 — `val_epoch` is what it means)."""
import argparse
import json

import torch
import torch.nn as nn

from . import single_modality as S


def build_parser():
    ap = argparse.ArgumentParser()
    for flag, typ, dflt in (('--embed_size', int, 128), ('--hidden_size', int, 128), ('--num_classes', int, 70),
                            ('--vocab_size', int, 134), ('--batch_size', int, 1024), ('--l_rate', float, 1e-5),
                            ('--num_epochs', int, 1000), ('--stats_after_every', int, 50), ('--num_workers', int, 4),
                            ('--checkpoint_path', str, None), ('--labels_file', str, '../data/labels.json'),
                            ('--split_file', str, '../data/split.json'), ('--q_dir', str, '../data/encoded_questions/'),
                            ('--v_dir', str, '../data/videos/'), ('--synthetic', int, 0)):
        ap.add_argument(flag, type=typ, default=dflt)
    ap.add_argument('--model', type=str, choices=['lstm', 'bow'], default='lstm')
    ap.add_argument('--use_class_weights', type=lambda v: str(v).lower() in ('1', 'true', 'yes'), default=True)
    return ap


class _SyntheticQuestions(torch.utils.data.Dataset):
    def __init__(self, n, vocab, classes, seed):
        from .dataset import SyntheticVNQADataset
        self.src = SyntheticVNQADataset(n, 2, 2, num_frames=3, vocab_size=vocab, num_classes=classes, seed=seed)
        self.classes = classes

    def __len__(self):
        return len(self.src)

    def __getitem__(self, i):
        X, y = self.src[i]
        return {'question': X['question'], 'q_len': X['q_len']}, y

    def get_class_weights(self):
        counts = torch.bincount(torch.tensor([self[i][1] for i in range(len(self))]), minlength=self.classes).double()
        return (1.0 / counts.clamp(min=1)).numpy()


def main(argv=None):
    args = build_parser().parse_args(argv)
    from torch.utils.data import DataLoader
    from ..models import QOnlyLSTM
    from .dataset import VNQADataset
    assert args.model == 'lstm', "only --model lstm is built on the MI355X path"
    assert torch.cuda.is_available(), "the MI355X path needs a GPU (there is no CPU fallback)"
    dev = torch.device('cuda', 0)
    if args.synthetic > 0:
        train = _SyntheticQuestions(args.synthetic, args.vocab_size, args.num_classes, 11)
        val = _SyntheticQuestions(max(args.synthetic // 4, args.batch_size), args.vocab_size, args.num_classes, 12)
    else:
        split, labels = json.load(open(args.split_file)), json.load(open(args.labels_file))
        train = VNQADataset(q_dir=args.q_dir, v_dir=args.v_dir, q_only=True, filenames=split['train'], labels=labels,
                            num_classes=args.num_classes)
        val = VNQADataset(q_dir=args.q_dir, v_dir=args.v_dir, q_only=True, filenames=split['val'], labels=labels,
                          num_classes=args.num_classes)
    print('%d train examples, %d validation examples' % (len(train), len(val)))
    loaders = [DataLoader(dataset=d, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
               for d in (train, val)]
    model = QOnlyLSTM(batch_size=args.batch_size, embedding_size=args.embed_size, hidden_size=args.hidden_size,
                      nb_classes=args.num_classes, vocab_size=args.vocab_size).to(dev)
    weights = torch.as_tensor(train.get_class_weights(), dtype=torch.float32, device=dev) if args.use_class_weights else None
    loss_fn = nn.CrossEntropyLoss(weight=weights)
    print(model)
    opt = torch.optim.Adam(model.parameters(), lr=args.l_rate)

    def sorted_batch(Xs, ys):
        lens, perm = Xs['q_len'].sort(0, descending=True)                 # :76-77
        return Xs['question'][perm].to(dev), lens, ys[perm].to(dev)

    def train_step(Xs, ys):
        q, lens, ys = sorted_batch(Xs, ys)
        opt.zero_grad()
        model.init_hidden()
        out = model(q, lens)
        loss = loss_fn(out, ys)
        loss.backward()
        opt.step()
        return loss.detach(), out, ys

    def val_step(Xs, ys):
        q, lens, ys = sorted_batch(Xs, ys)
        with torch.no_grad():
            out = model(q, lens)
            return loss_fn(out, ys), out, ys

    best = 0.0
    for epoch in range(1, args.num_epochs + 1):
        model.train()
        t = S.run_epoch(loaders[0], args.batch_size, train_step)
        if epoch % args.stats_after_every == 0 and t.n:
            f1w, f1m = t.f1()
            print('Train Epoch: {}\tAverage loss: {:.6f}\tF1: w{:.4f}, micro{:.4f}'.format(epoch, t.loss / t.n, f1w, f1m))
            model.eval()
            v = S.run_epoch(loaders[1], args.batch_size, val_step)
            f1w, f1m = v.f1() if v.n else (0.0, 0.0)
            print('Validation:\tAverage loss: {:.6f}, F1: w{:.4f}, micro{:.4f}'.format(v.loss / max(v.n, 1), f1w, f1m))
            if f1m > best:
                best = f1m
                S.save_if(args.checkpoint_path, {'epoch': epoch - 1, 'model': args.model, 'state_dict': model.state_dict(),
                                                 'val_acc': best, 'optimizer': opt.state_dict()})


if __name__ == '__main__':
    main()
