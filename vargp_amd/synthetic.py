"""Synthetic stand-ins for the reference's datasets (var_gp/datasets.py needs torchvision + network).

`mnist_like`: D-dimensional points with 10 classes, either N(0, 0.25/D) features ("gauss": keeps
K_uf = O(1)) or 19 %-dense U[0,1] pixels ("mnist").  Class structure is added by shifting
each class mean so that the toy classifier has something to learn.
"""
import math

import torch


def mnist_like(n, d=784, n_classes=10, kind='gauss', seed=0):
    g = torch.Generator().manual_seed(seed)
    y = torch.arange(n) % n_classes
    if kind == 'mnist':
        x = torch.rand(n, d, generator=g) * (torch.rand(n, d, generator=g) < 0.19)
    else:
        x = torch.randn(n, d, generator=g) * math.sqrt(0.25 / d)
        centers = torch.randn(n_classes, d, generator=g) * math.sqrt(0.25 / d)
        x = x + centers[y]
    return x.float(), y.long()
