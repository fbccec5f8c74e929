"""Datasets for the continual-learning driver: counterparts of the reference's `var_gp/datasets.py`
(ToyDataset :10-67, SplitMNIST :70-105, PermutedMNIST :108-138) without the torchvision dependency.

MNIST: if the four IDX files (`train-images-idx3-ubyte` ...) are present under `root` (optionally in
`MNIST/raw/`, optionally gzipped) they are parsed directly; otherwise — there is no network here —
`synthetic=True` builds an MNIST-shaped surrogate (784 features in [0,1], 10 classes) so that the whole
training loop can run.  Same attributes and methods as the reference classes: `.data`, `.targets`,
`.task_ids`, `filter_by_class`, `filter_by_idx`, `set_task`, tensor-indexable `__getitem__`.
"""
import gzip
import os
import warnings
import struct

import torch
from torch.utils.data import Dataset

from .synthetic import mnist_like


class ToyDataset(Dataset):
    """4-class 2-D toy problem, N_K points per class (reference datasets.py:10-67)."""

    def __init__(self, N_K=50, K=4, X=None, Y=None):
        super().__init__()
        if X is not None:
            self.data, self.targets = X, Y
        else:
            self.data, self.targets = self._init_data(N_K)
        self.task_ids = torch.arange(self.targets.size(0))

    @staticmethod
    def _init_data(n):
        def blob(mx, sx, my, sy):
            return torch.stack([mx + sx * torch.randn(n), my + sy * torch.randn(n)], dim=-1)
        xs = [blob(0.8, 0.4, 1.5, 0.4), blob(0.5, 0.6, -0.2, -0.1), blob(2.5, -0.1, 1.0, 0.6)]
        mvn = torch.distributions.MultivariateNormal(torch.tensor([-0.5, 1.5]),
                                                     covariance_matrix=torch.tensor([[0.2, 0.1], [0.1, 0.1]]))
        xs.append(mvn.sample(torch.Size([n])))
        X = torch.cat(xs, dim=0) - torch.tensor([0.5, 1.0])
        Y = torch.arange(4).repeat_interleave(n)
        return X, Y

    def filter_by_class(self, class_list=None):
        mask = torch.ones_like(self.targets).bool()
        if class_list:
            mask = torch.zeros_like(self.targets).bool()
            for c in class_list:
                mask |= self.targets == c
        self.task_ids = torch.arange(self.targets.size(0))[mask]

    def __getitem__(self, index):
        return self.data[self.task_ids[index]], self.targets[self.task_ids[index]]

    def __len__(self):
        return self.task_ids.size(0)


def _read_idx(path):
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rb') as f:
        magic, = struct.unpack('>I', f.read(4))
        nd = magic & 0xFF
        dims = struct.unpack('>' + 'I' * nd, f.read(4 * nd))
        return torch.frombuffer(bytearray(f.read()), dtype=torch.uint8).reshape(*dims)


def _find(root, stem):
    for d in (root, os.path.join(root, 'MNIST', 'raw')):
        for ext in ('', '.gz'):
            p = os.path.join(d, stem + ext)
            if os.path.exists(p):
                return p
    return None


kSyntheticScale = 0.15


def load_mnist(root, train=True, synthetic=None, n_synth=None, seed=0):
    """-> (data [N,784] float in [0,1], targets [N] int64)."""
    stem = 'train' if train else 't10k'
    pi, pl = _find(root, f'{stem}-images-idx3-ubyte'), _find(root, f'{stem}-labels-idx1-ubyte')
    if pi and pl and not synthetic:
        return _read_idx(pi).reshape(-1, 784).float() / 255., _read_idx(pl).long()
    if synthetic is False:
        raise FileNotFoundError(f'MNIST IDX files not found under {root} (no network to download them)')
    if synthetic is None:
        warnings.warn(f'MNIST IDX files not found under {root}: using the MNIST-shaped SYNTHETIC surrogate; accuracies '
                      'logged by this run are not MNIST accuracies (pass --synthetic to silence this)', stacklevel=2)
    n = n_synth or (60000 if train else 10000)
    # train and test share the class prototypes (seed) and differ in the per-sample randomness.  The surrogate is scaled so
    # that the reference's default kernel initialisation (lengthscale 0.5, kernels.py:15-16) sees informative distances:
    # at pixel range [0, 1] two samples of one class are ~72 apart in squared distance, i.e. k(x, x') = exp(-144) = 0 in
    # fp32 for every pair and nothing trains (the reference's algorithm on the CPU oracle behaves the same); at 0.15 the
    # within-class kernel values are ~0.05 and the model learns from the first steps.
    x, y = mnist_like(n, 784, 10, kind='mnist_classes', seed=seed, sample_seed=2 * seed + (1 if train else 2))
    return kSyntheticScale * x, y


class SplitMNIST(Dataset):
    def __init__(self, root='/tmp', train=True, synthetic=None, n_synth=None):
        self.data, self.targets = load_mnist(root, train, synthetic, n_synth)
        self.task_ids = torch.arange(self.targets.size(0))

    def filter_by_class(self, class_list=None):
        mask = torch.ones_like(self.targets).bool()
        if class_list:
            mask = torch.zeros_like(self.targets).bool()
            for c in class_list:
                mask |= self.targets == c
        self.task_ids = torch.arange(self.targets.size(0))[mask]

    def filter_by_idx(self, idx):
        self.data, self.targets = self.data[idx], self.targets[idx]
        self.task_ids = torch.arange(self.targets.size(0))

    def __getitem__(self, index):
        return self.data[self.task_ids[index]], self.targets[self.task_ids[index]]

    def __len__(self):
        return self.task_ids.size(0)


class PermutedMNIST(Dataset):
    @staticmethod
    def create_tasks(n=1):
        return [torch.randperm(784) for _ in range(n)]

    def __init__(self, root='/tmp', train=True, synthetic=None, n_synth=None):
        self.data, self.targets = load_mnist(root, train, synthetic, n_synth)
        self.perm = None

    def set_task(self, perm):
        assert self.perm is None, 'Cannot set task again.'
        self.data = self.data[:, perm]
        self.perm = perm

    def filter_by_idx(self, idx):
        self.data, self.targets = self.data[idx], self.targets[idx]

    def __getitem__(self, index):
        return self.data[index], self.targets[index]

    def __len__(self):
        return self.targets.size(0)
