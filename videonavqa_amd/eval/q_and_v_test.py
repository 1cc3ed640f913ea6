"""Counterpart of eval/q_and_v_test.py: test-split evaluation of a trained checkpoint on MI355X.

Same flags as q_and_v_eval (the reference duplicates its parser, q_and_v_test.py:29-60), same `test()` flow
(:64-142): the last short batch is PADDED to batch_size (questions 0 / q_len 1 / zero video / v_len 1 / q_id 35,
:80-87) and logits sliced back to the real examples (:123); writes `t_/p_/q_<checkpoint_path>.npy`
(targets, predictions, question ids; :268-271) for results_analysis.py.

usage: python -m videonavqa_amd.eval.q_and_v_test --model film_attn_pt --checkpoint_path at.pt [--synthetic N]
"""
import json
import os
import pprint as pp
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import utils as U
from .q_and_v_eval import build_model, build_parser, check_stem_against_checkpoint, stem_calibration


def _padded(args, Xs, ys):
    """The last short batch PADDED to batch_size (q_and_v_test.py:80-87): questions 0 / q_len 1 / zero video / v_len 1 /
    q_id 35 / label 0.  Returns (Xs, ys, q_ids, number of real examples)."""
    num_real = ys.size(0)
    q_ids = Xs.get('q_id', torch.zeros(num_real, dtype=torch.long))
    if num_real < args.batch_size:
        padded = args.batch_size - num_real
        Xs = dict(Xs)
        Xs['question'] = F.pad(Xs['question'], (0, 0, 0, padded), 'constant', 0)
        Xs['q_len'] = F.pad(Xs['q_len'], (0, padded), 'constant', 1)
        Xs['video'] = F.pad(Xs['video'], (0, 0, 0, 0, 0, 0, 0, 0, 0, padded), 'constant', 0)
        Xs['v_len'] = F.pad(Xs['v_len'], (0, padded), 'constant', 1)
        q_ids = F.pad(q_ids, (0, padded), 'constant', 35)
        ys = F.pad(ys, (0, padded), 'constant', 0)
    return Xs, ys, q_ids, num_real


def test(args, model, trainer, data_loader, loss_fn, device):
    """q_and_v_test.py:64-142 on the inference path (Trainer.eval_step: forward-only fused trunk; the stem and the H2D copy
    of the NEXT batch overlap this batch's trunk); loss, predictions, targets and question ids stay on the device and are
    read back once after the last batch."""
    from sklearn.metrics import f1_score
    model.eval()
    num_examples = 0
    loss_acc = torch.zeros((), dtype=torch.float64, device=device)
    preds, targets, qids, reals = [], [], [], []

    def stage(item):
        Xs, ys, q_ids, num_real = _padded(args, *item)
        clip = Xs['video'] if Xs['video'].dtype == torch.uint8 else Xs['video'].float()
        clip = trainer.upload(clip.pin_memory() if not clip.is_pinned() else clip)
        return (clip, Xs['question'].to(device, non_blocking=True), Xs['v_len'].long().cpu(), Xs['q_len'].long().cpu(),
                ys.to(device, non_blocking=True), q_ids.to(device, non_blocking=True), num_real)

    it = iter(data_loader)
    cur = next(it, None)
    cur = stage(cur) if cur is not None else None
    while cur is not None:
        nxt = next(it, None)
        nxt = stage(nxt) if nxt is not None else None
        clip, q, v_lens, q_lens, ys, q_ids, num_real = cur
        num_examples += num_real
        ahead = dict(next_clip=nxt[0], next_v_lens_cpu=nxt[2]) if nxt is not None else {}
        loss, output, perm_d = trainer.eval_step(clip, q, v_lens, q_lens, ys, n_real=num_real, **ahead)   # stem + sort + forward (:101-123)
        loss_acc += loss
        targets.append(ys.index_select(0, perm_d))                                          # :117-118 (sorted order)
        qids.append(q_ids.index_select(0, perm_d))
        preds.append(output.max(1)[1])
        reals.append(num_real)
        cur = nxt
    if preds:       # the ONE read-back
        P, T, Q = (torch.stack(t).cpu().numpy() for t in (preds, targets, qids))
        y_pred = np.concatenate([P[k][:n] for k, n in enumerate(reals)]).astype(np.float64)     # :123 the first num_real sorted rows
        y_target = np.concatenate([T[k][:n] for k, n in enumerate(reals)]).astype(np.float64)
        qs = np.concatenate([Q[k][:n] for k, n in enumerate(reals)]).astype(np.float64)
    else:
        y_pred, y_target, qs = np.array([]), np.array([]), np.array([])
    test_loss = float(loss_acc)
    hit = int((y_pred == y_target).sum())
    accs = U.per_class_accuracies(y_target, y_pred, args.num_classes)
    pp.pprint({i: accs[i] for i in np.nonzero(accs)[0].tolist()})
    f1_w = f1_score(y_target, y_pred, average='weighted')
    f1_micro = f1_score(y_target, y_pred, average='micro')
    print('Testing:\tAverage loss: {:.6f}, Accuracy: {}/{}, F1: w{:.4f}, micro{:.4f}\n'.format(
        test_loss / max(num_examples, 1), hit, num_examples, f1_w, f1_micro))
    return y_target, y_pred, qs


def main(argv=None):
    args = build_parser().parse_args(argv)
    from torch.utils.data import DataLoader
    from ..stem import FrozenStem, get_frcnn_feature_extractor
    from ..train import Trainer
    from .dataset import SyntheticVNQADataset, VNQADataset
    assert torch.cuda.is_available(), "the MI355X path needs a GPU (there is no CPU fallback)"
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.synthetic > 0:
        test_data = SyntheticVNQADataset(args.synthetic, args.height, args.width, vocab_size=args.vocab_size,
                                         num_classes=args.num_classes, seed=777)
    else:
        with open(U.SPLIT_FILE, 'r') as f:
            split = json.load(f)
        with open(U.LABELS_FILE, 'r') as f:
            labels = json.load(f)
        test_data = VNQADataset(q_dir=U.QUESTIONS_DIR, v_dir=U.VIDEOS_DIR, filenames=split['test'], labels=labels,
                                q_metadata=True)
    print('%d test examples' % len(test_data))
    test_loader = DataLoader(dataset=test_data, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers)
    spatial = (args.height // 16) * (args.width // 16)
    # the frozen conv1x1_layers are not in state_dict(): same seed as q_and_v_eval.main so that a checkpoint WITHOUT the
    # 'extra_state' key (written upstream / by an older build) still meets the 1x1 convs it was trained with
    torch.manual_seed(0)
    model = build_model(args, spatial).to(device)
    feature_extractor = get_frcnn_feature_extractor(args.frcnn_pretrained_path, args.precision).to(device)
    obj_detector = U.get_object_detector(precision=args.precision,
                                         load=args.synthetic == 0 or os.path.exists(U.OBJ_DETECTOR_PATH)).to(device)
    if args.checkpoint_path is None or not os.path.exists(args.checkpoint_path):            # :252-255
        print('=> No checkpoint existent! Aborting.')
        sys.exit(-1)
    print('=> Restoring from checkpoint path %s' % args.checkpoint_path)
    checkpoint = torch.load(args.checkpoint_path, map_location=device)
    # The frozen stem's 16-bit weights are rounded against calibration means (stem.coherent_round): test behind the SAME weights the
    # checkpoint was trained behind — its own means when it carries them (written by Trainer.extra_state_dict); without them 'noise'
    # / 'off' are reproducible from the flags alone, 'data' is not (it would calibrate on the TEST split) and is refused
    calib = (checkpoint.get('extra_state') or {}).get('_stem_calibration')
    if calib is None:
        if getattr(args, 'stem_calibration', 'noise') == 'data':
            print("=> --stem_calibration data: the checkpoint carries no calibration means (an older build wrote it) and the training "
                  "frames are not available here — pass 'noise' or 'off', whichever the training run used. Aborting.")
            sys.exit(-1)
        calib = stem_calibration(args, None)
    stem = FrozenStem(feature_extractor, obj_detector, args.precision, calibration=calib, split_features=args.model != 'mac',
                      split_depth=getattr(model, 'stem_split_depth', None))
    check_stem_against_checkpoint(stem, checkpoint)
    if args.precision == 'fp16h':
        print('=> stem: precision fp16h, %d split activation tensors' % stem.split_active)
    reduction = 'mean' if args.loss_reduction == 'elementwise_mean' else args.loss_reduction
    loss_fn = nn.CrossEntropyLoss(reduction=reduction)
    trainer = Trainer(model, stem, loss_reduction=reduction, feature_channels=args.num_input_channels)
    model.load_state_dict(checkpoint['state_dict'])
    if checkpoint.get('extra_state') and hasattr(model, 'load_reference_tensors'):
        model.load_reference_tensors(checkpoint['extra_state'])
    print('==> Restored checkpoint from epoch %d (validation accuracy %.4f)' %
          (checkpoint['epoch'] + 1, checkpoint.get('val_acc', -1.0)))
    t, p, q = test(args, model, trainer, test_loader, loss_fn, device)
    np.save('t_' + args.checkpoint_path, t)                                                 # :268-271
    np.save('p_' + args.checkpoint_path, p)
    np.save('q_' + args.checkpoint_path, q)


if __name__ == '__main__':
    main()
