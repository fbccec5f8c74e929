"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from oracle import vargp_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PARAM_KEYS = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar', 'prior_log_mean', 'prior_log_logvar']
GRAD_KEYS = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']

# Tolerances (north star: ELBO rtol 1e-4 in fp32)
RTOL_SCALAR = 1e-4
# The 2-D toy problem at tasks t>0 is ill-conditioned (clustered inducing points, K_uu eigenvalues at
# the 1e-4 jitter floor): the REFERENCE's own fp32 result is 0.9e-4..1.3e-4 away from its fp64 result
# there (tests/diag_toy.py), so agreement with the fp32 golden to 1e-4 is not a meaningful bar for
# those cases; they are held to 5e-4 against the golden AND against the fp64 oracle.
RTOL_SCALAR_ILLCOND = 5e-4
ILLCOND_CASES = ('toy_t1', 'toy_t2')


def rtol_for(name):
    return RTOL_SCALAR_ILLCOND if name in ILLCOND_CASES else RTOL_SCALAR

ATOL_PRED, RTOL_PRED = 1e-3, 1e-3
ATOL_PROBS = 1e-4
REL_L2_GRAD = 1e-3


def load_case(name):
    """-> (g, params, prev, x, y, noise) with tensors rebuilt from the fixture (or regenerated
    from the seed for the full-size case that stores outputs only)."""
    g = np.load(os.path.join(GOLDEN, f'{name}.npz'))
    S, F_, C, M, D, B, n_prev, seed = [int(v) for v in g['meta']]
    if 'x' in g.files:
        params = {k: torch.from_numpy(g[f'p_{k}']) for k in PARAM_KEYS}
        prev = [{k: torch.from_numpy(g[f'prev{i}_{k}']) for k in ['z', 'u_mean', 'u_tril_vec']}
                for i in range(n_prev)]
        noise = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('n_')}
        x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    else:
        params, prev, x, y, noise = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=seed, kind=str(g['kind']),
                                                     ell=float(g['ell']) if 'ell' in g.files else 0.5)
    return g, params, prev, x, y, noise


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def to_dev(obj, dev):
    if isinstance(obj, dict):
        return {k: to_dev(v, dev) for k, v in obj.items()}
    if isinstance(obj, list):
        return [to_dev(v, dev) for v in obj]
    return obj.to(dev)


def load_retrain_case(name):
    """-> (g, params, prev, x, y, noise) of a VARGPRetrain fixture (inputs stored)."""
    g = np.load(os.path.join(GOLDEN, f'{name}.npz'))
    n_prev = int(g['meta'][6])
    params = {k: torch.from_numpy(g[f'p_{k}']) for k in PARAM_KEYS}
    prev = [{k: torch.from_numpy(g[f'prev{i}_{k}']) for k in ['z', 'u_mean', 'u_tril_vec']} for i in range(n_prev)]
    noise = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('n_')}
    return g, params, prev, torch.from_numpy(g['x']), torch.from_numpy(g['y']), noise
