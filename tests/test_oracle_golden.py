"""CPU: the oracle (oracle/vargp_oracle.py) against golden vectors produced by the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import (load_case, rel_l2, RTOL_SCALAR, ATOL_PRED, RTOL_PRED, ATOL_PROBS, REL_L2_GRAD,
                     GRAD_KEYS, GOLDEN)

E2E = ['toy_t0', 'toy_t1', 'toy_t2', 'smnist_small_t0', 'smnist_small_t1', 'wtoy_t1', 'wtoy_t2', 'wtoy_t1_nomean',
       'wtoy_t2_nomean']


@pytest.fixture(scope='module')
def ops():
    return np.load(f'{GOLDEN}/ops.npz')


@pytest.mark.parametrize('tag', ['toy', 'mnist'])
def test_rbf_gram(ops, tag):
    th, x, y = (torch.from_numpy(ops[f'rbf_{tag}_{k}']) for k in ['theta', 'x', 'y'])
    kuu = orc.rbf_gram(th, x)
    kuf = orc.rbf_gram(th, x, y)
    np.testing.assert_allclose(kuu.numpy(), ops[f'rbf_{tag}_kuu'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(kuf.numpy(), ops[f'rbf_{tag}_kuf'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(orc.rbf_gram(th, x, y, full_gram=True).numpy(), ops[f'rbf_{tag}_kuf'],
                               rtol=2e-5, atol=1e-6)
    # K_uu diagonal is exactly gamma^2
    g2 = (2 * th[:, -1]).exp()
    assert torch.equal(kuu.diagonal(dim1=-2, dim2=-1), g2.view(-1, 1, 1).expand(kuu.shape[:-1]))


def test_tril_pack(ops):
    v = torch.from_numpy(ops['tril_vec'])
    np.testing.assert_allclose(orc.vec2tril(v).numpy(), ops['tril_mat'], rtol=1e-6, atol=0)
    np.testing.assert_allclose(orc.mat2trilvec(orc.vec2tril(v)).numpy(), ops['tril_back'], rtol=1e-6)


def test_linear_gaussian_ops(ops):
    t = {k: torch.from_numpy(ops[k]) for k in ops.files if k.startswith(('lg_', 'lj_', 'lmd_', 'gc_', 'kl_'))}
    mu, Sig, Lz, LzK = orc.linear_joint(t['lg_m'], t['lg_S'], t['lg_Kzx'], t['lg_Kzz'], t['lg_V'], t['lg_b'])
    for got, want in [(mu, 'lj_mu'), (Sig, 'lj_Sig'), (Lz, 'lj_Lz'), (LzK, 'lj_LzKzx')]:
        np.testing.assert_allclose(got.numpy(), ops[want], rtol=2e-4, atol=2e-5)
    mu, var, _, _ = orc.linear_marginal_diag(t['lg_m'], t['lg_S'], t['lg_Kzz'], t['lg_Kzx'],
                                             orc.rbf_diag(t['lg_theta']))
    np.testing.assert_allclose(mu.numpy(), ops['lmd_mu'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(var.numpy(), ops['lmd_var'], rtol=2e-4, atol=2e-5)
    Lz = orc.chol(t['lg_Kzz'])
    LzK = torch.linalg.solve_triangular(Lz, t['lg_Kzx'], upper=False)
    mu, Sig = orc.gp_cond(t['lg_m'], t['lg_Kxx'], Lz, LzK)
    np.testing.assert_allclose(mu.numpy(), ops['gc_mu'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(Sig.numpy(), ops['gc_Sig'], rtol=2e-4, atol=2e-5)
    kl = orc.mvn_kl(t['lg_m'].squeeze(-1), t['kl_Lq'], torch.zeros_like(t['lg_m'].squeeze(-1)), t['kl_Lp'])
    np.testing.assert_allclose(kl.numpy(), ops['kl_val'], rtol=1e-5)


def test_likelihood(ops):
    mu, var, y, eps = (torch.from_numpy(ops[f'lik_{k}']) for k in ['mu', 'var', 'y', 'eps'])
    np.testing.assert_allclose(orc.softmax_nll(mu, var, y, eps).numpy(), ops['lik_nll'], rtol=1e-6)
    np.testing.assert_allclose(orc.softmax_predict(mu, var, eps).numpy(), ops['lik_probs'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('name', E2E)
def test_e2e_loss_grads_predict(name):
    g, params, prev, x, y, noise = load_case(name)
    epm = bool(int(g['ep_var_mean'])) if 'ep_var_mean' in g.files else True     # False pins the u_<t / gp_cond branch
    sc, grads = orc.elbo_step(params, prev, x, y, noise, beta=float(g['beta']), n_total=float(g['n_total']),
                              ep_var_mean=epm)
    for k in ['kl_hypers', 'kl_u', 'nll', 'total']:
        np.testing.assert_allclose(sc[k].item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    for k in GRAD_KEYS:
        assert rel_l2(grads[k], g[f'grad_{k}']) < REL_L2_GRAD, k
    pmu, pvar, _ = orc.forward(params, prev, x, noise)
    np.testing.assert_allclose(pmu.numpy(), g['pred_mu'], rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(pvar.numpy(), g['pred_var'], rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(orc.predict(params, prev, x, noise).numpy(), g['probs'], atol=ATOL_PROBS)


@pytest.mark.parametrize('name', ['smnist_full_t0', 'pmnist_full_t0', 'pmnist_red_t1', 'pmnist_red_t2', 'smnist_s64_t0',
                                  'smnist_full_t0_mnist', 'smnist_full_t0_mnist_l25', 'smnist_full_t1'])
def test_e2e_full_size(name):
    """Outputs-only fixtures (inputs regenerated from the seed, outputs from the reference): Cfg2 (S3 F10 C10 M100 D784
    B512), Cfg3 task 0 at full size (S10 M200) and tasks 1, 2 (Mt = 400, 600) at reduced D/B, Cfg4's S = 64 unsharded;
    Cfg2 on MNIST-like data (19 %-dense U[0,1] pixels) at the initial lengthscale 0.5 (K_uf underflows to 0: the z-gradient
    is exactly 0) and at lengthscale 2.5 (K_uf ~ 1e-3)."""
    g, params, prev, x, y, noise = load_case(name)
    sc, grads = orc.elbo_step(params, prev, x, y, noise, beta=float(g['beta']), n_total=float(g['n_total']))
    for k in ['kl_hypers', 'kl_u', 'nll', 'total']:
        np.testing.assert_allclose(sc[k].item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    for k in GRAD_KEYS:
        np.testing.assert_allclose(grads[k].double().norm().item(), float(g[f'gradnorm_{k}']), rtol=1e-3)
    assert rel_l2(grads['log_mean'], g['grad_log_mean']) < REL_L2_GRAD
    assert rel_l2(grads['u_mean'], g['grad_u_mean']) < REL_L2_GRAD
    assert rel_l2(grads['z'][:, :4, :], g['grad_z_head']) < REL_L2_GRAD


def test_fp64_matches_fp32():
    """The oracle is dtype-generic; fp64 run bounds the fp32 self-noise."""
    params, prev, x, y, noise = orc.make_problem(2, 3, 4, 8, 6, 16, n_prev=1, seed=3, kind='toy', dtype=torch.float64)
    sc64, _ = orc.elbo_step(params, prev, x, y, noise)
    f32 = lambda o: {k: v.float() for k, v in o.items()}
    sc32, _ = orc.elbo_step(f32(params), [f32(p) for p in prev], x.float(), y, f32(noise))
    for k in sc64:
        np.testing.assert_allclose(sc32[k].item(), sc64[k].item(), rtol=1e-4)


@pytest.mark.parametrize('name', ['retrain_wtoy_t1', 'retrain_wtoy_t2'])
def test_retrain_loss_and_grads(name):
    """VARGPRetrain (var_gp/vargp_retrain.py:119-233): the oracle's restatement against the reference's loss triple and
    the gradients of the current and of the re-optimised earlier-task parameters."""
    from helpers import load_retrain_case
    g, params, prev, x, y, noise = load_retrain_case(name)
    names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
    leaf = dict(params)
    for k in names:
        leaf[k] = params[k].clone().requires_grad_(True)
    retrain = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in prev]
    kl_h, kl_u, nll = orc.retrain_loss(leaf, retrain, prev, x, y, noise)
    total = float(g['beta']) * kl_h + kl_u + (float(g['n_total']) / x.shape[0]) * nll
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll), ('total', total)]:
        np.testing.assert_allclose(v.item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    total.backward()
    for k in names:
        assert rel_l2(leaf[k].grad, g[f'grad_{k}']) < REL_L2_GRAD, k
    for i, p in enumerate(retrain):
        for k in p:
            assert rel_l2(p[k].grad, g[f'grad_retrain{i}_{k}']) < REL_L2_GRAD, (i, k)


@pytest.mark.parametrize('name', ['dkl_t0', 'dkl_t1'])
def test_deep_kernel_loss_and_grads(name):
    """DeepRBFKernel (var_gp/kernels.py:80-96) inside VARGP.loss: the oracle with its feature-map hook against the
    reference's loss triple and gradients (model parameters and the feature map's weights)."""
    g = np.load(f'{GOLDEN}/{name}.npz')
    S, F_, C, M, D, B, n_prev, seed = [int(v) for v in g['meta']]
    params, prev, x, y, noise, phi = orc.make_dkl_problem(S, F_, C, M, D, B, n_prev, seed)
    names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
    leaf = dict(params)
    for k in names:
        leaf[k] = params[k].clone().requires_grad_(True)
    phi = {k: v.clone().requires_grad_(True) for k, v in phi.items()}
    with orc.deep_kernel(phi):
        kl_h, kl_u, nll = orc.loss(leaf, prev, x, y, noise)
        total = float(g['beta']) * kl_h + kl_u + (float(g['n_total']) / B) * nll
        total.backward()
        with torch.no_grad():
            probs = orc.predict(params, prev, x, noise)
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll), ('total', total)]:
        np.testing.assert_allclose(v.item(), float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    for k in names:
        assert rel_l2(leaf[k].grad, g[f'grad_{k}']) < REL_L2_GRAD, k
    for k, v in phi.items():
        if k == '4.bias':      # the RBF kernel is translation-invariant in feature space: this gradient is exactly 0
            assert v.grad.abs().max().item() < 1e-4 and np.abs(g[f'grad_phi_{k}']).max() < 1e-4
            continue
        assert rel_l2(v.grad, g[f'grad_phi_{k}']) < REL_L2_GRAD, k
    np.testing.assert_allclose(probs.numpy(), g['probs'], atol=ATOL_PROBS)


TRAJ = ['traj_wtoy_t0', 'traj_wtoy_t1', 'traj_mid_t0']


def load_trajectory(name):
    """-> (g, params, prev, x, y, noise_of(k), steps, lr, beta, n_total) of a trajectory fixture (inputs regenerated)."""
    g = np.load(f'{GOLDEN}/{name}.npz')
    S, F_, C, M, D, B, n_prev, seed = [int(v) for v in g['meta']]
    params, prev, x, y, _ = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=seed, kind=str(g['kind']))
    noise_of = lambda k: orc.step_noise(S, F_, C, M, D, B, n_prev, seed, k)
    return g, params, prev, x, y, noise_of, len(g['triples']), float(g['lr']), float(g['beta']), float(g['n_total'])


def check_compact_final(final, g, tol):
    """Final parameters of a full-size trajectory fixture (make_golden.trajectory_case(compact=True)): the small tensors in
    full, the first inducing points of z / first packed entries of u_tril_vec, and for every tensor the norm of its total
    displacement from the initial value (the part of the parameter the optimiser produced)."""
    for k in ('u_mean', 'log_mean', 'log_logvar'):
        assert rel_l2(final[k], g[f'final_{k}']) < tol, k
    assert rel_l2(final['z'][:, :4, :], g['final_z_head']) < tol
    assert rel_l2(final['u_tril_vec'][:, :64], g['final_u_tril_vec_head']) < tol
    for k, v in final.items():
        np.testing.assert_allclose(torch.as_tensor(v).double().norm().item(), float(g[f'finalnorm_{k}']), rtol=tol, err_msg=k)


def test_adam_trajectory_full_size_matches_reference():
    """Six Adam steps of the reference at BASELINE config 2's real shape (S3 F10 C10 M100 D784 B512)."""
    g, params, prev, x, y, noise_of, steps, lr, beta, n_total = load_trajectory('traj_full_t0')
    triples, final = orc.adam_trajectory(params, prev, x, y, noise_of, steps, lr, beta, n_total)
    np.testing.assert_allclose(triples.numpy(), g['triples'], rtol=5e-5)
    check_compact_final(final, g, 1e-4)
    for k, v in final.items():      # the displacement itself (6 steps of lr 1e-2: ~6e-2 per element), not hidden behind the value
        d = (v.double() - params[k].double()).norm().item()
        np.testing.assert_allclose(d, float(g[f'deltanorm_{k}']), rtol=2e-3, err_msg=k)


@pytest.mark.parametrize('name', TRAJ)
def test_adam_trajectory_matches_reference(name):
    """Multi-step fixture: the reference's loop (zero_grad, loss, combine, backward, Adam step) with fresh
    injected noise per step; the oracle in the same loop reproduces every step's loss triple and the final parameters."""
    g, params, prev, x, y, noise_of, steps, lr, beta, n_total = load_trajectory(name)
    triples, final = orc.adam_trajectory(params, prev, x, y, noise_of, steps, lr, beta, n_total)
    np.testing.assert_allclose(triples.numpy(), g['triples'], rtol=5e-5)
    for k, v in final.items():
        assert rel_l2(v, g[f'final_{k}']) < 1e-4, k
