"""GPU: DeepRBFKernel (vargp_amd/kernels.py, reference var_gp/kernels.py:80-96) — the feature map on the MFMA GEMM with
the fused bias / ReLU pass, forward and backward — inside VARGP.loss against golden vectors produced by the reference;
create_clf(dkl=True) incl. the carry-over of the feature map between tasks (var_gp/vargp.py:218-235)."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import rel_l2, to_dev, GOLDEN, RTOL_SCALAR, REL_L2_GRAD, ATOL_PROBS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_linear_act_fwd_bwd():
    from vargp_amd import ops
    x = orc.hash_normal((3, 37, 20), 1).float()
    w = orc.hash_normal((50, 20), 2).float() * 0.3
    b = orc.hash_normal((50,), 3).float() * 0.1
    gy = orc.hash_normal((3, 37, 50), 4).float()
    for relu in (True, False):
        xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = ops.linear_act(xd, wd, bd, relu)
        (y * gy.to(DEV)).sum().backward()
        x6, w6, b6 = (t.double().requires_grad_(True) for t in (x, w, b))
        y6 = torch.nn.functional.linear(x6, w6, b6)
        y6 = torch.relu(y6) if relu else y6
        (y6 * gy.double()).sum().backward()
        assert rel_l2(y.detach().cpu(), y6.detach()) < 1e-6
        for a, c in ((xd, x6), (wd, w6), (bd, b6)):
            assert rel_l2(a.grad.cpu(), c.grad) < 1e-5


@pytest.mark.parametrize('name', ['dkl_t0', 'dkl_t1'])
def test_deep_kernel_vs_reference_golden(name):
    from vargp_amd import noise
    from vargp_amd.kernels import DeepRBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.vargp import VARGP
    g = np.load(f'{GOLDEN}/{name}.npz')
    S, F_, C, M, D, B, n_prev, seed = [int(v) for v in g['meta']]
    params, prev, x, y, nz, phi = orc.make_dkl_problem(S, F_, C, M, D, B, n_prev, seed)
    kern = DeepRBFKernel(D, prior_log_mean=params['prior_log_mean'], prior_log_logvar=params['prior_log_logvar'])
    kern.phi.load_state_dict(phi)
    gp = VARGP(params['z'], kern, MulticlassSoftmax(n_f=F_), n_var_samples=S,
               prev_params=[{k: v.clone() for k, v in p.items()} for p in prev])
    with torch.no_grad():
        gp.kernel.log_mean.copy_(params['log_mean'])
        gp.kernel.log_logvar.copy_(params['log_logvar'])
        gp.u_mean.copy_(params['u_mean'])
        gp.u_tril_vec.copy_(params['u_tril_vec'])
    gp = gp.to(DEV)
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        total = float(g['beta']) * kl_h + kl_u + (float(g['n_total']) / B) * nll
        total.backward()
        with torch.no_grad():
            probs = gp.predict(x.to(DEV))
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll), ('total', total)]:
        np.testing.assert_allclose(v.item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    grads = dict(z=gp.z.grad, u_mean=gp.u_mean.grad, u_tril_vec=gp.u_tril_vec.grad, log_mean=gp.kernel.log_mean.grad,
                 log_logvar=gp.kernel.log_logvar.grad)
    for k, v in grads.items():
        assert rel_l2(v.cpu(), g[f'grad_{k}']) < REL_L2_GRAD, k
    for k, v in gp.kernel.phi.named_parameters():
        if k == '4.bias':      # translation invariance in feature space: exactly 0 up to rounding
            assert v.grad.abs().max().item() < 1e-4
            continue
        assert rel_l2(v.grad.cpu(), g[f'grad_phi_{k}']) < REL_L2_GRAD, k
    np.testing.assert_allclose(probs.cpu().numpy(), g['probs'], atol=ATOL_PROBS)


def test_create_clf_dkl_carries_feature_map_over():
    from vargp_amd.datasets import ToyDataset
    from vargp_amd.vargp import VARGP
    ds = ToyDataset()
    gp0 = VARGP.create_clf(ds, M=8, dkl=True)
    sd = {k: v.clone() for k, v in gp0.state_dict().items()}
    assert 'kernel.phi.0.weight' in sd and 'kernel.phi.4.bias' in sd and sd['kernel.log_mean'].shape == (65,)
    gp1 = VARGP.create_clf(ds, M=8, dkl=True, prev_params=[sd])
    assert torch.equal(gp1.kernel.phi[2].weight, gp0.kernel.phi[2].weight)
    assert torch.equal(gp1.kernel.prior_log_mean, gp0.kernel.log_mean.detach())
    assert sorted(gp1.prev_params[0]) == ['u_mean', 'u_tril_vec', 'z']
