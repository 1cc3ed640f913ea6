"""Shared epoch loop of the single-modality entry points (eval/q_only_eval.py, eval/v_only_cnn3d_eval.py of the
reference): batches below --batch_size are skipped, metrics are loss / accuracy / weighted+micro F1."""
import numpy as np
import torch


class Tally(object):
    def __init__(self):
        self.loss, self.hit, self.n = 0.0, 0, 0
        self.pred, self.target = [], []

    def add(self, loss, logits, ys):
        p = logits.detach().max(1)[1]
        self.loss += float(loss)
        self.hit += int((p == ys).sum())
        self.n += len(ys)
        self.pred.append(p.cpu().numpy())
        self.target.append(ys.cpu().numpy())

    def f1(self):
        from sklearn.metrics import f1_score
        t, p = np.concatenate(self.target), np.concatenate(self.pred)
        return f1_score(t, p, average='weighted'), f1_score(t, p, average='micro')


def run_epoch(loader, batch_size, step):
    """step(Xs, ys) -> (loss tensor, logits, ys as scored).  Returns the Tally over all full batches."""
    tally = Tally()
    for Xs, ys in loader:
        if len(ys) < batch_size:
            continue
        loss, logits, ys_scored = step(Xs, ys)
        tally.add(loss, logits, ys_scored)
    return tally


def save_if(path, state, prefix=''):
    if path is not None:
        torch.save(state, prefix + path)
