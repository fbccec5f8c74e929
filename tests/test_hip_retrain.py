"""GPU: VARGPRetrain (vargp_amd/vargp_retrain.py, reference var_gp/vargp_retrain.py:11-267) against golden vectors
produced by the reference: loss triple and the gradients of the current and of the re-optimised earlier-task parameters."""
import numpy as np
import pytest
import torch

from helpers import load_retrain_case, rel_l2, to_dev, RTOL_SCALAR, REL_L2_GRAD

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _build(params, prev, S, F_):
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.vargp_retrain import VARGPRetrain
    D = params['z'].shape[-1]
    kern = RBFKernel(D, prior_log_mean=params['prior_log_mean'], prior_log_logvar=params['prior_log_logvar'])
    gp = VARGPRetrain(params['z'], kern, MulticlassSoftmax(n_f=F_), n_var_samples=S,
                      prev_params=[{k: v.clone().to(DEV) for k, v in p.items()} for p in prev])
    with torch.no_grad():
        gp.kernel.log_mean.copy_(params['log_mean'])
        gp.kernel.log_logvar.copy_(params['log_logvar'])
        gp.u_mean.copy_(params['u_mean'])
        gp.u_tril_vec.copy_(params['u_tril_vec'])
    return gp.to(DEV)


@pytest.mark.parametrize('name', ['retrain_wtoy_t1', 'retrain_wtoy_t2'])
def test_retrain_loss_grads_vs_reference_golden(name):
    from vargp_amd import noise
    g, params, prev, x, y, nz = load_retrain_case(name)
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    gp = _build(params, prev, S, F_)
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        total = float(g['beta']) * kl_h + kl_u + (float(g['n_total']) / x.shape[0]) * nll
        total.backward()
        with torch.no_grad():
            probs = gp.predict(x.to(DEV))
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll), ('total', total)]:
        np.testing.assert_allclose(v.item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    grads = dict(z=gp.z.grad, u_mean=gp.u_mean.grad, u_tril_vec=gp.u_tril_vec.grad, log_mean=gp.kernel.log_mean.grad,
                 log_logvar=gp.kernel.log_logvar.grad)
    for k, v in grads.items():
        assert rel_l2(v.cpu(), g[f'grad_{k}']) < REL_L2_GRAD, k
    for i, pd in enumerate(gp.retrain_params):
        for k in ('z', 'u_mean', 'u_tril_vec'):
            assert rel_l2(pd[k].grad.cpu(), g[f'grad_retrain{i}_{k}']) < REL_L2_GRAD, (i, k)
    assert probs.shape == (x.shape[0], params['z'].shape[0])
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, atol=1e-5)


def test_retrain_first_task_equals_vargp():
    """Without earlier tasks the model is the plain first-task VAR-GP (vargp_retrain.py:171-190)."""
    from oracle import vargp_oracle as orc
    from vargp_amd import noise
    S, F_, C, M, D, B = 2, 3, 3, 12, 6, 32
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=3, kind='gauss')
    gp = _build(params, [], S, F_)
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
    want = orc.loss(params, [], x, y, nz)
    for v, w in zip((kl_h, kl_u, nll), want):
        np.testing.assert_allclose(v.item(), w.item(), rtol=RTOL_SCALAR)
