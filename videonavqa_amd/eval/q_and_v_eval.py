"""Counterpart of eval/q_and_v_eval.py: train / validate the video+question FiLM models on MI355X.

Same command-line flags and defaults as the reference (eval/q_and_v_eval.py:29-64) for the three models
on the hot path (`--model film_attn_pt | film_gp_pt | time_multi_hop | mac`), same train_epoch / val_epoch
flow, printed line formats and checkpoint schema (`e{epoch}_{checkpoint_path}`, keys epoch / model /
state_dict / train_f1w / train_f1micro / optimizer).  Additions (all optional):
  --synthetic N        N seeded synthetic items per split instead of ../data (no dataset on the box)
  --precision bf16|fp32, --height/--width, and data-parallel launch through torchrun / torch.distributed.run
  (one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the environment; rank 0 prints and checkpoints).
Known upstream pitfalls kept or fixed deliberately: `type=bool` flags are parsed as real
booleans here; `--loss_reduction` defaults to 'sum' (upstream has no default and eval.sh always passes sum);
eval.sh's extra `--best_acc` flag is accepted and ignored.

usage: python -m videonavqa_amd.eval.q_and_v_eval --model film_attn_pt --synthetic 64 --num_epochs 1
"""
import argparse
import json
import os
import pprint as pp

import numpy as np
import torch
import torch.nn as nn

from . import utils as U


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes", "y", "t")


def build_parser():
    parser = argparse.ArgumentParser()
    # Model args (q_and_v_eval.py:32-38); the off-path raw-video baselines (concat2d/concat3d) are not built
    parser.add_argument('--model', type=str, choices=['film_gp_pt', 'film_attn_pt', 'mac', 'time_multi_hop'],
                        required=True)
    parser.add_argument('--num_classes', type=int, default=70)
    parser.add_argument('--q_encoder', type=str, choices=['lstm', 'bow'], default='lstm')
    parser.add_argument('--use_obj_detector', type=str2bool, default=True)
    parser.add_argument('--use_visual_features', type=str2bool, default=True)
    parser.add_argument('--vocab_size', type=int, default=134)
    # Model hyperparameters (:41-49)
    parser.add_argument('--embed_size', type=int, default=128)
    parser.add_argument('--hidden_size', type=int, default=128)
    parser.add_argument('--at_hidden_size', type=int, default=128)
    parser.add_argument('--num_res_blocks', type=int, default=1)
    parser.add_argument('--num_res_block_channels', type=int, default=512)
    parser.add_argument('--num_input_channels', type=int, default=512)
    parser.add_argument('--num_tail_channels', type=int, default=16)
    parser.add_argument('--mac_dim', type=int, default=512)
    parser.add_argument('--mac_max_step', type=int, default=12)
    # Optimization args (:52-57)
    parser.add_argument('--batch_size', type=int, default=8)
    parser.add_argument('--clip_value', type=float, default=1.0)
    parser.add_argument('--l_rate', type=float, default=1e-4)
    parser.add_argument('--loss_reduction', type=str, choices=['sum', 'mean', 'elementwise_mean'], default='sum')
    parser.add_argument('--num_epochs', type=int, default=1)
    parser.add_argument('--use_class_weights', type=str2bool, default=False)
    # Other args (:60-64)
    parser.add_argument('--checkpoint_path', type=str)
    parser.add_argument('--frcnn_pretrained_path', type=str)
    parser.add_argument('--num_workers', type=int, default=4)
    parser.add_argument('--stats_after_every', type=int, default=400)
    parser.add_argument('--val_only', type=str2bool, default=False)
    parser.add_argument('--best_acc', type=float, default=0)          # passed by eval.sh:57, unused upstream
    # additions
    parser.add_argument('--synthetic', type=int, default=0)
    parser.add_argument('--precision', type=str, choices=['bf16', 'fp16', 'fp16h', 'fp32'], default='fp16h')
    # how the frozen stem's 16-bit weights are rounded (stem.coherent_round): against the mean activations of seeded noise frames
    # ('noise'), of frames of the first training videos ('data'), or to nearest ('off')
    parser.add_argument('--stem_calibration', type=str, choices=['noise', 'data', 'off'], default='noise')
    parser.add_argument('--height', type=int, default=U.VID_HEIGHT)
    parser.add_argument('--width', type=int, default=U.VID_WIDTH)
    return parser


def build_model(args, spatial_size):
    """Model factory of q_and_v_eval.py:255-303."""
    from ..models import (FiLMAttnPretrainedStem, FiLMGlobalPoolingPretrainedStem, MACNetwork,
                          TimeMultiHopFiLMPretrainedStem)
    extra = dict(spatial_size=spatial_size, precision=args.precision)
    if args.model == 'mac':                                             # :288-293
        return MACNetwork(n_vocab=args.vocab_size, dim=args.mac_dim, embed_hidden=args.embed_size,
                          max_step=args.mac_max_step, classes=args.num_classes, precision=args.precision)
    if args.model == 'film_attn_pt':
        return FiLMAttnPretrainedStem(batch_size=args.batch_size, q_embedding_size=args.embed_size,
                                      nb_classes=args.num_classes, q_encoder=args.q_encoder,
                                      num_input_channels=args.num_input_channels,
                                      num_res_block_channels=args.num_res_block_channels,
                                      num_res_blocks=args.num_res_blocks, hidden_size=args.hidden_size,
                                      at_hidden_size=args.at_hidden_size,
                                      max_num_frames=U.MAX_ALLOWED_NUM_FRAMES_DROPPING,
                                      vocab_size=args.vocab_size, **extra)
    if args.model == 'film_gp_pt':
        return FiLMGlobalPoolingPretrainedStem(batch_size=args.batch_size, q_embedding_size=args.embed_size,
                                               nb_classes=args.num_classes,
                                               num_input_channels=args.num_input_channels,
                                               num_res_block_channels=args.num_res_block_channels,
                                               num_res_blocks=args.num_res_blocks, hidden_size=args.hidden_size,
                                               num_tail_channels=args.num_tail_channels, q_encoder=args.q_encoder,
                                               vocab_size=args.vocab_size, **extra)
    return TimeMultiHopFiLMPretrainedStem(batch_size=args.batch_size, q_embedding_size=args.embed_size,
                                          nb_classes=args.num_classes,
                                          num_input_channels=args.num_input_channels,
                                          num_res_block_channels=args.num_res_block_channels,
                                          num_res_blocks=args.num_res_blocks,
                                          num_tail_channels=args.num_tail_channels, hidden_size=args.hidden_size,
                                          vocab_size=args.vocab_size, **extra)


def _to_device(Xs, ys, device):
    clip = Xs['video'].float().to(device, non_blocking=True)
    q = Xs['question'].to(device, non_blocking=True)
    return clip, q, Xs['v_len'].long().cpu(), Xs['q_len'].long().cpu(), ys.to(device, non_blocking=True)


def _staged_batches(args, trainer, data_loader, device):
    """(index, batch, next_batch) with the clips already being uploaded: the H2D copy of the NEXT minibatch and its
    frozen stem overlap the current minibatch's trunk pass (Trainer.upload / Trainer.step(next_clip=...))."""
    def stage(item):
        i, (Xs, ys) = item
        clip = Xs['video']
        if clip.dtype != torch.uint8:          # (VNQADataset(uint8_video=True) hands out raw pixels: uploaded as they are)
            clip = clip.float()
        clip = trainer.upload(clip.pin_memory() if not clip.is_pinned() else clip)
        return i, (clip, Xs['question'].to(device, non_blocking=True), Xs['v_len'].long().cpu(),
                   Xs['q_len'].long().cpu(), ys.to(device, non_blocking=True))
    full = (it for it in enumerate(data_loader, 0) if len(it[1][1]) >= args.batch_size)     # :86-87 skips short batches
    cur = next(full, None)
    cur = stage(cur) if cur is not None else None
    while cur is not None:
        nxt = next(full, None)
        nxt = stage(nxt) if nxt is not None else None
        yield cur[0], cur[1], (nxt[1] if nxt is not None else None)
        cur = nxt


def _global_stats(loss_sum, hit, n, device):
    """Sum (loss, hits, examples) over data-parallel ranks for the per-epoch log line: the one scalar all-reduce per epoch
    besides the gradient all-reduce (the reference is single-process; F1 stays rank-local)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([loss_sum, float(hit), float(n)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0]), int(t[1]), int(t[2])
    return loss_sum, hit, n


def train_epoch(epoch, args, trainer, data_loader, device, rank=0):
    """q_and_v_eval.py:73-156 (stem, sort, forward, loss, clip, Adam are inside Trainer.step)."""
    from sklearn.metrics import f1_score
    # Loss, predictions and targets stay ON THE DEVICE during the epoch: a float(loss) / .cpu() per step would make the
    # launch thread wait for the whole step and serialise upload(i+2) / stem(i+1) behind trunk(i).  They are read back at
    # the --stats_after_every print and once at the end of the epoch.
    num_examples = 0
    loss_acc = torch.zeros((), dtype=torch.float64, device=device)
    preds, targets = [], []
    for i, batch, nxt in _staged_batches(args, trainer, data_loader, device):
        clip, q, v_lens, q_lens, ys = batch
        num_examples += len(ys)
        perm = torch.sort(v_lens, dim=0, descending=True, stable=True)[1]
        ahead = dict(next_clip=nxt[0], next_v_lens_cpu=nxt[2]) if nxt is not None else {}
        loss, logits = trainer.step(clip, q, v_lens, q_lens, ys, **ahead)
        targets.append(ys[trainer.to_device_async(perm)])              # :117 (sorted order, as the logits)
        loss_acc += loss
        preds.append(logits.max(1)[1])                                  # :127
        if rank == 0 and (i + 1) % args.stats_after_every == 0:
            print('Average loss after %d iterations in epoch %d: %.6f' % (i + 1, epoch + 1, float(loss_acc) / num_examples))
    if preds:
        pred_t, targ_t = torch.cat(preds), torch.cat(targets)
        hit = int((pred_t == targ_t).sum())
        y_pred, y_target = pred_t.cpu().numpy().astype(np.float64), targ_t.cpu().numpy().astype(np.float64)
    else:
        hit, y_pred, y_target = 0, np.array([]), np.array([])
    avg_loss = float(loss_acc)
    f1_w = f1_score(y_target, y_pred, average='weighted')
    f1_micro = f1_score(y_target, y_pred, average='micro')
    avg_loss, hit, num_examples = _global_stats(avg_loss, hit, num_examples, device)
    if rank == 0:
        print('Train Epoch: {}\tAverage loss: {:.6f}\tAccuracy: {}/{}\tF1: w{:.4f}, micro{:.4f}\n'.format(
            epoch, avg_loss / max(num_examples, 1), hit, num_examples, f1_w, f1_micro))
        if args.checkpoint_path is not None:
            torch.save({'epoch': epoch, 'model': args.model, 'state_dict': trainer.model.state_dict(),
                        'train_f1w': f1_w, 'train_f1micro': f1_micro,
                        'optimizer': trainer.optimizer_state_dict(),
                        # not in the reference schema (its loaders ignore unknown keys): the frozen conv1x1_layers, which
                        # state_dict() does not carry — without them a restored model evaluates with different weights
                        'extra_state': trainer.extra_state_dict()},
                       'e' + str(epoch) + '_' + args.checkpoint_path)   # :148-156
    return avg_loss / max(num_examples, 1)


class ShardedBatchSampler(object):
    """Batch sampler of the validation / test split for data-parallel evaluation: the split's FULL batches in order
    (val_epoch skips a short last batch, q_and_v_eval.py:188-189), batch i evaluated by rank i % world only — every rank
    decodes and evaluates 1 / world of the split instead of all of it.  keep_short=True also yields the short last batch
    (the test script pads it, q_and_v_test.py:80-87)."""

    def __init__(self, n_items, batch_size, rank=0, world=1, keep_short=False):
        n_batches = n_items // batch_size + (1 if (keep_short and n_items % batch_size) else 0)
        self.batches = [list(range(i * batch_size, min((i + 1) * batch_size, n_items)))
                        for i in range(n_batches) if i % world == rank]
        self.global_index = [i for i in range(n_batches) if i % world == rank]
        self.n_batches_total = n_batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def gather_eval_shards(per_batch, loss_sum, n_examples, world, device=None):
    """Merge the ranks' evaluation shards with ONE collective (all_gather_object).  per_batch: this rank's list of
    (global batch index, 1-D float64 arrays...) — e.g. (i, y_target, y_pred) — in any order.  Returns (list of merged
    arrays, concatenated in global batch order — exactly the single-process loop's order —, total loss, total examples);
    identical on every rank."""
    import torch.distributed as dist
    shards = [(per_batch, float(loss_sum), int(n_examples))]
    if world > 1 and dist.is_available() and dist.is_initialized():
        out = [None] * world
        dist.all_gather_object(out, shards[0])
        shards = out
    rows = sorted((item for sh in shards for item in sh[0]), key=lambda it: it[0])
    n_arrays = len(rows[0]) - 1 if rows else 0
    merged = [np.concatenate([np.asarray(r[1 + k], dtype=np.float64) for r in rows]) if rows else np.array([])
              for k in range(max(n_arrays, 2))]
    return merged, sum(sh[1] for sh in shards), sum(sh[2] for sh in shards)


def stem_calibration(args, dataset, n_frames=40):
    """FrozenStem's `calibration` argument from --stem_calibration: 'noise' (seeded noise frames), None ('off': round-to-nearest) or
    [n_frames, 3, H, W] frames in [0, 1] from the FIRST items of `dataset` (the same on every rank) — the first valid frames of as many
    videos as it takes."""
    mode = getattr(args, "stem_calibration", "noise")
    if mode == "off":
        return None
    if mode == "noise" or dataset is None or len(dataset) == 0:
        return "noise"
    n_items = min(len(dataset), n_frames)
    per_item = -(-n_frames // n_items)
    frames = []
    for i in range(n_items):
        item = dataset[i][0]
        v = torch.as_tensor(item['video'])
        v = v.float() / 255.0 if v.dtype == torch.uint8 else v.float()                # (uint8_video datasets: raw pixels k -> k / 255)
        frames.append(v[:, :, :, :max(1, min(int(item['v_len']), per_item))].permute(3, 0, 1, 2))
    return torch.cat(frames)[:n_frames].contiguous()


def check_stem_against_checkpoint(stem, ckpt, rank=0):
    """The rebuilt stem's rounded 16-bit packs against the checksum the checkpoint carries (Trainer.extra_state_dict): re-rounding from
    the calibration frames is bit-reproducible on the same GPU / ROCm / torch build only.  Returns True / False / None (nothing to check)."""
    want = ((ckpt or {}).get('extra_state') or {}).get('_stem_packs_sha256')
    if want is None or not hasattr(stem, 'packs_checksum'):
        return None
    ok = stem.packs_checksum() == want
    if rank == 0:
        print('=> stem weights %s' % ('reproduced bit for bit from the checkpoint\'s calibration' if ok else
                                      'DIFFER from the ones this checkpoint was trained behind (another GPU / ROCm / torch build re-rounded a tie '
                                      'differently, or another precision / split depth): expect logits differences of the order of one 16-bit weight rounding'))
    return ok


def val_epoch(args, trainer, data_loader, device, rank=0, world=1):
    """q_and_v_eval.py:159-224 on the inference path: Trainer.eval_step (forward-only fused trunk, the stem of the NEXT
    minibatch and its H2D copy overlapping this minibatch's trunk, as in training), loss / predictions / targets kept on the
    device for the whole epoch and read back ONCE; with world > 1 the loader hands this rank only its share of the batches
    (ShardedBatchSampler) and the shards are merged by one gather."""
    from sklearn.metrics import f1_score
    trainer.model.eval()
    num_examples = 0
    loss_acc = torch.zeros((), dtype=torch.float64, device=device)
    preds, targets = [], []
    for i, batch, nxt in _staged_batches(args, trainer, data_loader, device):
        clip, q, v_lens, q_lens, ys = batch
        num_examples += len(ys)
        ahead = dict(next_clip=nxt[0], next_v_lens_cpu=nxt[2]) if nxt is not None else {}
        loss, output, perm_d = trainer.eval_step(clip, q, v_lens, q_lens, ys, **ahead)
        loss_acc += loss
        targets.append(ys.index_select(0, perm_d))                      # sorted order, as the logits (:195-199)
        preds.append(output.max(1)[1])
    sampler = getattr(data_loader, "batch_sampler", None)
    index = getattr(sampler, "global_index", None) or list(range(len(preds)))
    if preds:      # the epoch's ONE read-back
        P = torch.stack(preds).cpu().numpy().astype(np.float64)
        T = torch.stack(targets).cpu().numpy().astype(np.float64)
        per_batch = [(index[k], T[k], P[k]) for k in range(len(preds))]
    else:
        per_batch = []
    (y_target, y_pred), val_loss, num_examples = gather_eval_shards(per_batch, float(loss_acc), num_examples, world, device)
    hit = int((y_pred == y_target).sum())
    accs = U.per_class_accuracies(y_target, y_pred, args.num_classes)
    f1_w = f1_score(y_target, y_pred, average='weighted') if num_examples else 0.0
    f1_micro = f1_score(y_target, y_pred, average='micro') if num_examples else 0.0
    if rank == 0:
        pp.pprint({i: accs[i] for i in np.nonzero(accs)[0].tolist()})
        print('Validation:\tAverage loss: {:.6f}, Accuracy: {}/{}, F1: w{:.4f}, micro{:.4f}\n'.format(
            val_loss / max(num_examples, 1), hit, num_examples, f1_w, f1_micro))
    return val_loss / max(num_examples, 1)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch.distributed as dist
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from ..stem import FrozenStem, get_frcnn_feature_extractor
    from ..train import Trainer
    from .dataset import SyntheticVNQADataset, VNQADataset

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "the MI355X path needs a GPU (there is no CPU fallback)"
    if os.environ.get("VNQA_SINGLE_DEVICE") == "1":       # test hook: several ranks on ONE GPU (with VNQA_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VNQA_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            # RCCL's kernels on a high-priority stream: they are dispatched ahead of the co-running stem's workgroups, like the trunk's
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.synthetic > 0:
        train_data = SyntheticVNQADataset(args.synthetic, args.height, args.width, vocab_size=args.vocab_size,
                                          num_classes=args.num_classes, seed=1234)
        val_data = SyntheticVNQADataset(max(args.synthetic // 4, args.batch_size), args.height, args.width,
                                        vocab_size=args.vocab_size, num_classes=args.num_classes, seed=4321)
    else:
        with open(U.SPLIT_FILE, 'r') as f:                              # :234-240
            split = json.load(f)
        with open(U.LABELS_FILE, 'r') as f:
            labels = json.load(f)
        train_data = VNQADataset(q_dir=U.QUESTIONS_DIR, v_dir=U.VIDEOS_DIR, filenames=split['train'], labels=labels)
        val_data = VNQADataset(q_dir=U.QUESTIONS_DIR, v_dir=U.VIDEOS_DIR, filenames=split['val'], labels=labels)
    if rank == 0:
        print('%d train examples, %d validation examples' % (len(train_data), len(val_data)))
    tr_sampler = DistributedSampler(train_data, world, rank, shuffle=True) if world > 1 else None
    train_loader = DataLoader(dataset=train_data, batch_size=args.batch_size, shuffle=tr_sampler is None,
                              sampler=tr_sampler, num_workers=args.num_workers, drop_last=False)
    # validation is sharded over the ranks by BATCH (rank r evaluates batches r, r + world, ...) and merged by one gather
    val_loader = DataLoader(dataset=val_data, num_workers=args.num_workers,
                            batch_sampler=ShardedBatchSampler(len(val_data), args.batch_size, rank, world))

    spatial = (args.height // 16) * (args.width // 16)
    torch.manual_seed(0)
    model = build_model(args, spatial).to(device)
    assert args.use_visual_features and args.use_obj_detector, \
        "the MI355X path implements the --use_visual_features/--use_obj_detector configuration (the defaults)"
    feature_extractor = get_frcnn_feature_extractor(args.frcnn_pretrained_path, args.precision).to(device)
    obj_detector = U.get_object_detector(precision=args.precision,
                                         load=args.synthetic == 0 or os.path.exists(U.OBJ_DETECTOR_PATH)).to(device)
    if rank == 0:
        print(obj_detector)
        print(model)
    # A RESUMED run continues behind the stem weights it started with: the checkpoint's own calibration (frames / means), not the
    # flags' (the flags may disagree, and the default number of calibration frames changed between builds)
    ckpt = None
    calib = stem_calibration(args, train_data)
    if args.checkpoint_path is not None and os.path.exists(args.checkpoint_path):
        ckpt = torch.load(args.checkpoint_path, map_location=device)
        saved = (ckpt.get('extra_state') or {}).get('_stem_calibration')
        if saved is not None and args.precision != 'fp32':
            kind = saved.get('frames') if isinstance(saved.get('frames'), str) else 'data'
            if rank == 0 and kind != getattr(args, 'stem_calibration', 'noise'):
                print("=> WARNING: --stem_calibration %s, but the checkpoint was trained behind a stem calibrated on '%s': using the "
                      "checkpoint's" % (args.stem_calibration, kind))
            calib = saved
    stem = FrozenStem(feature_extractor, obj_detector, args.precision, calibration=calib, split_features=args.model != 'mac',
                      split_depth=getattr(model, 'stem_split_depth', None))
    check_stem_against_checkpoint(stem, ckpt, rank)
    if rank == 0 and args.precision == 'fp16h':
        print('=> stem: precision fp16h, %d split activation tensors, weights %s' %
              (stem.split_active, 'second-order rounded' if stem.second_order else 'rounded to nearest / coherently'))

    class_weights = None
    if args.use_class_weights and hasattr(train_data, "get_class_weights"):
        class_weights = torch.FloatTensor(train_data.get_class_weights()).to(device)
    reduction = 'mean' if args.loss_reduction == 'elementwise_mean' else args.loss_reduction
    trainer = Trainer(model, stem, lr=args.l_rate, clip=args.clip_value, loss_reduction=reduction,
                      class_weights=class_weights, world_size=world, rank=rank,
                      feature_channels=args.num_input_channels)

    start_epoch = 0
    if args.checkpoint_path is not None:                                # :337-346
        if not os.path.exists(args.checkpoint_path):
            if rank == 0:
                print('=> No checkpoint existent - will save the model here')
        else:
            if rank == 0:
                print('=> Restoring from checkpoint path %s' % args.checkpoint_path)
            start_epoch = ckpt['epoch'] + 1
            trainer.load_checkpoint(ckpt)
            if rank == 0:
                print('==> Restored checkpoint %s (epoch %d)' % (args.checkpoint_path, start_epoch))

    for epoch in range(start_epoch, start_epoch + args.num_epochs):     # :354-365
        if tr_sampler is not None:
            tr_sampler.set_epoch(epoch)
        if not args.val_only:
            train_epoch(epoch, args, trainer, train_loader, device, rank)
        if args.model == 'mac':                                         # :357-363, as upstream: AFTER epoch 0 the
            trainer.lr = args.l_rate / 10. if epoch == 0 else args.l_rate   # rate drops to l_rate/10 for one epoch
            if rank == 0:
                print('learning rate %.5f' % trainer.lr)
        val_epoch(args, trainer, val_loader, device, rank, world)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
