"""CPU: the block-structured form of the t > 0 ELBO (tests/block_algorithm.py — what the native program
vargp_elbo_tn computes) equals the oracle's restatement of the reference's linear_joint chain, in fp64 to rounding and
in fp32 to the parity tolerance, for ep_var_mean True and False, and its hand-derived backward equals autograd."""
import numpy as np
import pytest
import torch

import block_algorithm as blk
from oracle import vargp_oracle as orc


def _d(o):
    return {k: v.double() for k, v in o.items()}


@pytest.mark.parametrize('n_prev', [0, 1, 3])
@pytest.mark.parametrize('epm', [True, False])
def test_block_form_equals_chain_fp64(n_prev, epm):
    params, prev, x, y, nz = orc.make_problem(2, 3, 3, 6, 4, 10, n_prev=n_prev, seed=5, kind='gauss', n_v=3)
    params, prev, x, nz = _d(params), [_d(p) for p in prev], x.double(), _d(nz)
    want = orc.loss(params, prev, x, y, nz, ep_var_mean=epm)
    got = blk.forward(params, prev, x, y, nz, ep_var_mean=epm)
    for a, b, k in zip(got, want, ['kl_hypers', 'kl_u', 'nll']):
        np.testing.assert_allclose(a.item(), b.item(), rtol=1e-9, err_msg=k)


@pytest.mark.parametrize('name', ['wtoy_t1', 'wtoy_t2', 'wtoy_t2_nomean', 'smnist_small_t1'])
def test_block_form_vs_reference_golden_fp32(name):
    from helpers import load_case, RTOL_SCALAR
    g, params, prev, x, y, nz = load_case(name)
    epm = bool(int(g['ep_var_mean'])) if 'ep_var_mean' in g.files else True
    got = blk.forward(params, prev, x, y, nz, ep_var_mean=epm)
    for a, k in zip(got, ['kl_hypers', 'kl_u', 'nll']):
        np.testing.assert_allclose(a.item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)


@pytest.mark.parametrize('n_prev', [0, 1, 3])
def test_hand_backward_equals_autograd(n_prev):
    params, prev, x, y, nz = orc.make_problem(2, 3, 3, 6, 4, 10, n_prev=n_prev, seed=6, kind='gauss')
    params, prev, x, nz = _d(params), [_d(p) for p in prev], x.double(), _d(nz)
    if not prev:
        pytest.skip('the block program is only used for t > 0 (t = 0 has its own program)')
    _, want = orc.elbo_step(params, prev, x, y, nz, beta=3.0, n_total=70)
    got = blk.step_with_hand_backward(params, prev, x, y, nz, seeds=(3.0, 1.0, 7.0))
    for k in want:
        err = ((got[k] - want[k]).norm() / want[k].norm()).item()
        assert err < 1e-9, (k, err)
