"""Synthetic stand-ins for the reference's datasets (var_gp/datasets.py needs torchvision + network).

`mnist_like`: D-dimensional points with 10 classes, either N(0, 0.25/D) features ("gauss": keeps
K_uf = O(1)) or 19 %-dense U[0,1] pixels ("mnist").  Class structure is added by shifting
each class mean so that the toy classifier has something to learn.
"""
import math

import torch


def mnist_like(n, d=784, n_classes=10, kind='gauss', seed=0, sample_seed=None):
    """sample_seed ('mnist_classes' only): seed of the per-sample randomness, so that several draws (train / test) share
    the class prototypes that `seed` fixes."""
    g = torch.Generator().manual_seed(seed)
    y = torch.arange(n) % n_classes
    if kind == 'mnist':
        x = torch.rand(n, d, generator=g) * (torch.rand(n, d, generator=g) < 0.19)
    elif kind == 'mnist_classes':
        # learnable surrogate with MNIST-like geometry: every class has 16 sparse "stroke" prototypes (19 % of
        # the pixels on); a sample is one prototype with half of its pixels dropped, a few spurious pixels and
        # random intensities, so same-class points are far from identical (no near-singular K_uu)
        n_proto = 16
        protos = (torch.rand(n_classes, n_proto, d, generator=g) < 0.19).float()
        if sample_seed is not None:
            g = torch.Generator().manual_seed(sample_seed)
        which = torch.randint(0, n_proto, (n,), generator=g)
        keep = (torch.rand(n, d, generator=g) < 0.5).float()
        extra = (torch.rand(n, d, generator=g) < 0.03).float()
        x = ((protos[y, which] * keep + extra).clamp(0, 1) * (0.5 + 0.5 * torch.rand(n, d, generator=g)))
    else:
        x = torch.randn(n, d, generator=g) * math.sqrt(0.25 / d)
        centers = torch.randn(n_classes, d, generator=g) * math.sqrt(0.25 / d)
        x = x + centers[y]
    return x.float(), y.long()
