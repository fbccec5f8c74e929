"""Training utilities of the driver (counterparts of the reference's `var_gp/train_utils.py`:
set_seeds :13-18, compute_accuracy :21-35, compute_acc_ent :38-56, compute_bwt :59-65, EarlyStopper :69-98)."""
import random

import numpy as np
import torch
from torch.utils.data import DataLoader


def set_seeds(seed=None):
    if seed:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def compute_accuracy(dataset, gp, batch_size=512, device=None):
    """Fraction of argmax(predict) == label; asserts on NaN predictions like the reference (:29)."""
    with torch.no_grad():
        count = 0
        for x, y in DataLoader(dataset, batch_size=batch_size):
            preds = gp.predict(x.to(device))
            assert not torch.isnan(preds).any(), 'Found NaNs'
            count += (preds.argmax(dim=-1) == y.to(device)).sum().item()
    return count / len(dataset)


def compute_acc_ent(dataset, gp, batch_size=512, device=None):
    with torch.no_grad():
        corr, ent = 0, 0.
        for x, y in DataLoader(dataset, batch_size=batch_size):
            preds = gp.predict(x.to(device))
            assert not torch.isnan(preds).any(), 'Found NaNs'
            corr += (preds.argmax(dim=-1) == y.to(device)).sum().item()
            ent += torch.distributions.Categorical(probs=preds).entropy().sum().item()
    return corr / len(dataset), ent / len(dataset)


def compute_bwt(acc_mat):
    assert acc_mat.ndim == 2 and acc_mat.shape[0] == acc_mat.shape[1]
    return (acc_mat[-1][:-1] - acc_mat.diagonal()[:-1]).mean()


class EarlyStopper:
    """Keeps the best-scoring `info` and counts evaluations without improvement (patience < 0: never
    stops).  As in the reference, `info` is stored by reference — a `state_dict()` passed in aliases
    the live parameters (SURVEY §5 'Checkpoint / resume')."""

    def __init__(self, patience=10, delta=1e-4):
        self.patience, self.delta = patience, delta
        self._counter, self._best_info, self._best_score = 0, None, None

    def is_done(self):
        return self.patience >= 0 and self._counter >= self.patience

    def info(self):
        return self._best_info

    def __call__(self, score, info):
        assert not self.is_done()
        if self._best_score is None or score >= self._best_score + self.delta:
            self._best_score, self._best_info, self._counter = score, info, 0
        else:
            self._counter += 1
