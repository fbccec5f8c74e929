"""Where the Monte-Carlo noise of one ELBO step comes from.

The reference draws three tensors per `loss()` call from torch's global generator:
theta-noise (S, D+1) in RBFKernel.sample_hypers (var_gp/kernels.py:66-67), u-noise
(n_v, S, C, Mt-M) in VARGP.forward for t>0 (var_gp/vargp.py:137-138) and f-noise (S, F, C, B) in
MulticlassSoftmax.forward (var_gp/likelihoods.py:26).  Here they are requested by name so that
  * parity tests can inject the exact tensors the oracle / reference used, and
  * the sample-parallel multi-GPU path can hand every rank its slice of ONE global draw
    (same seed on all ranks, rows [r*S/R, (r+1)*S/R) of the sample dimension).
"""
import contextlib

import torch

_injected = {}
_shard = None   # (rank, world, generator) for sample-parallel runs


@contextlib.contextmanager
def inject(**tensors):
    """Use the given tensors (by name: eps_theta, eps_u, eps_f) instead of drawing."""
    old = dict(_injected)
    _injected.update(tensors)
    try:
        yield
    finally:
        _injected.clear()
        _injected.update(old)


def set_shard(rank, world, seed, device, counts=None):
    """Sample-parallel mode: every rank draws the same global tensor from a generator seeded with `seed` and keeps its
    rows of the sample dimension.  `counts[r]` = hyper-samples of rank r (default: the same number on every rank, taken
    from the requested shape); rank r owns rows [sum(counts[:r]), sum(counts[:r+1])) of a (sum(counts), ...) draw."""
    global _shard
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    _shard = (rank, world, gen, None if counts is None else [int(c) for c in counts])


def clear_shard():
    global _shard
    _shard = None


def draw(name, shape, device, sample_dim=0):
    """Standard-normal tensor `name` of `shape`; `sample_dim` is the dim that indexes hyper-samples S."""
    t = _injected.get(name)
    if t is not None:
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
        return t.to(device)
    if _shard is not None:
        rank, world, gen, counts = _shard
        full = list(shape)
        if counts is None:
            full[sample_dim] *= world
            first = rank * shape[sample_dim]
        else:
            assert shape[sample_dim] == counts[rank], (name, shape, counts, rank)
            full[sample_dim] = sum(counts)
            first = sum(counts[:rank])
        g = torch.randn(*full, device=device, generator=gen)
        return g.narrow(sample_dim, first, shape[sample_dim])
    return torch.randn(*shape, device=device)
