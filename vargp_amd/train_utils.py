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


def materialize(dataset, device=None):
    """The whole data set as two tensors (x (N, D), y (N)) on `device`: every data set of var_gp/datasets.py is
    tensor-indexable (VARGP.create_clf relies on that, vargp.py:207-209); a ConcatDataset is the concatenation of its parts."""
    from torch.utils.data import ConcatDataset
    if isinstance(dataset, ConcatDataset):
        xs, ys = zip(*[materialize(d, device) for d in dataset.datasets])
        return torch.cat(xs), torch.cat(ys)
    x, y = dataset[torch.arange(len(dataset))]
    return x.to(device).contiguous(), y.to(device).contiguous()


class DeviceBatches:
    """Device-resident replacement of `DataLoader(train_set, batch_size, shuffle=True)` (experiments/vargp.py:26): data and
    targets live in HBM, every epoch draws ONE on-device permutation, a minibatch is its index slice -- gathered by the
    consumer with index_select (one small kernel per tensor), no host copy and no host sync anywhere in the epoch.
    Iterating yields index tensors; `take(idx)` gathers (x, y)."""

    def __init__(self, dataset, batch_size, device, shuffle=True):
        self.data, self.targets = materialize(dataset, device)
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.n = self.targets.size(0)

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def epoch_order(self):
        """The epoch's order of the training points (device int64): one on-device permutation, or range(n) without shuffling."""
        return (torch.randperm(self.n, device=self.data.device) if self.shuffle
                else torch.arange(self.n, device=self.data.device))

    def __iter__(self):
        order = self.epoch_order()
        for i in range(0, self.n, self.batch_size):
            yield order[i:i + self.batch_size]

    def take(self, idx):
        return self.data.index_select(0, idx), self.targets.index_select(0, idx)


def _sweep(dataset, gp, batch_size, device, shared_hypers):
    """Class probabilities of `gp` over `dataset`, one (probs, labels) pair per chunk, everything left on the device.
    shared_hypers: the whole set goes through ONE predict(x, tile=batch_size) call -- one hyper-parameter draw and one
    factorisation of K(z_<=t) for the sweep instead of one per batch (the reference, train_utils.py:25-27, re-draws per
    batch; every batch's prediction has the same distribution either way)."""
    loader = DataLoader(dataset, batch_size=batch_size)
    import inspect
    if shared_hypers and 'tile' in inspect.signature(gp.predict).parameters:
        xs, ys = zip(*[(x, y) for x, y in loader])
        yield gp.predict(torch.cat(xs).to(device), tile=batch_size), torch.cat(ys).to(device)
        return
    for x, y in loader:
        yield gp.predict(x.to(device)), y.to(device)


def compute_accuracy(dataset, gp, batch_size=512, device=None, shared_hypers=False):
    """Fraction of argmax(predict) == label; asserts on NaN predictions like the reference (:29).  The hit count and the
    NaN flag are accumulated on the device: ONE host sync per data set, not one per batch."""
    with torch.no_grad():
        hits, bad = None, None
        for preds, y in _sweep(dataset, gp, batch_size, device, shared_hypers):
            h, b = (preds.argmax(dim=-1) == y).sum(), torch.isnan(preds).any()
            hits, bad = (h, b) if hits is None else (hits + h, bad | b)
        assert not bool(bad), 'Found NaNs'
    return int(hits) / len(dataset)


def compute_acc_ent(dataset, gp, batch_size=512, device=None, shared_hypers=False):
    """(accuracy, mean predictive entropy), accumulated on the device like compute_accuracy."""
    with torch.no_grad():
        hits, ent, bad = None, None, None
        for preds, y in _sweep(dataset, gp, batch_size, device, shared_hypers):
            h, b = (preds.argmax(dim=-1) == y).sum(), torch.isnan(preds).any()
            e = -(preds * preds.clamp_min(torch.finfo(preds.dtype).tiny).log()).sum()
            hits, ent, bad = (h, e, b) if hits is None else (hits + h, ent + e, bad | b)
        assert not bool(bad), 'Found NaNs'
    return int(hits) / len(dataset), float(ent) / len(dataset)


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
