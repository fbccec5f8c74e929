"""GPU: parity at the sizes of the BASELINE configs beyond Cfg2 — Cfg3 (Permuted-MNIST: M = 200, S = 10, Mt up to
1000 here), Cfg4 (Split-MNIST task 0 with S = 64, unsharded and split over two ranks) and Cfg5 (M = 2048 stress) —
against golden vectors produced by the reference (tests/golden/make_golden.py) and against the fp64 oracle.
Tolerances: helpers.py (ELBO scalars 1e-4, gradients rel-L2 1e-3, predictive moments 1e-3, probabilities 1e-4)."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import load_case, rel_l2, to_dev, RTOL_SCALAR, ATOL_PRED, RTOL_PRED, ATOL_PROBS, REL_L2_GRAD, GRAD_KEYS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _d(o):
    return {k: v.double() for k, v in o.items()}


# ------------------------------------------------------------------------------------------------------------
# outputs-only goldens at (near) full size
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['pmnist_full_t0', 'pmnist_red_t1', 'pmnist_red_t2', 'smnist_s64_t0'])
def test_full_size_vs_reference_golden(name):
    """Cfg3 task 0 (S10 F10 C10 M200 D784 B512), Cfg3 tasks 1, 2 (Mt = 400, 600; D 64, B 128), Cfg4 on one GPU (S = 64)."""
    from test_hip_e2e import _run
    g, (params, prev, x, y, nz), sc, grads, pmu, pvar, probs = _run(name)
    for k in ['kl_hypers', 'kl_u', 'nll', 'total']:
        np.testing.assert_allclose(sc[k], float(g[k]), rtol=RTOL_SCALAR, err_msg=k)
    for k in GRAD_KEYS:
        np.testing.assert_allclose(grads[k].double().norm().item(), float(g[f'gradnorm_{k}']), rtol=1e-3, err_msg=k)
    assert rel_l2(grads['log_mean'].cpu(), g['grad_log_mean']) < REL_L2_GRAD
    assert rel_l2(grads['u_mean'].cpu(), g['grad_u_mean']) < REL_L2_GRAD
    assert rel_l2(grads['z'][:, :4, :].cpu(), g['grad_z_head']) < REL_L2_GRAD
    ns = g['pred_mu'].shape[0]                 # S > 16: the fixture keeps the first four hyper-samples
    np.testing.assert_allclose(pmu[:ns].numpy(), g['pred_mu'], rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(pvar[:ns].numpy(), g['pred_var'], rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(probs.numpy(), g['probs'], atol=ATOL_PROBS)


# ------------------------------------------------------------------------------------------------------------
# Cholesky + inverse factor at Cfg3-late / Cfg5 sizes
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,nb', [(1000, 2), (2048, 1)])
def test_chol_inv_large_fwd_bwd(n, nb):
    """L = chol(K + eps I), T = L^-1 and the gradient of <gL, L> + <gT, T> at n = 1000 / 2048 (blocked path: register
    diagonal blocks + MFMA panel / trailing GEMMs) against fp64 LAPACK + autograd on an RBF kernel matrix."""
    from vargp_amd import ops
    D = 64
    z = orc.hash_normal((nb, n, D), 5) * np.sqrt(0.25 / D)
    th = torch.full((1, D + 1), np.log(0.5), dtype=torch.float64)
    K64 = orc.rbf_gram(th, z)[0]                                   # (nb, n, n) fp64
    A = K64.float()
    gL = orc.hash_normal((nb, n, n), 7).tril().float() / n
    gT = orc.hash_normal((nb, n, n), 9).tril().float() / n
    Ad = A.to(DEV).requires_grad_(True)
    L, T = ops.chol_inv(Ad, 1e-4)
    ((L * gL.to(DEV)).sum() + (T * gT.to(DEV)).sum()).backward()
    A64 = A.double().requires_grad_(True)
    L64 = torch.linalg.cholesky(A64 + 1e-4 * torch.eye(n, dtype=torch.float64))
    T64 = torch.linalg.solve_triangular(L64, torch.eye(n, dtype=torch.float64).expand(nb, n, n), upper=False)
    ((L64 * gL.double()).sum() + (T64 * gT.double()).sum()).backward()
    # bar: no worse than twice what fp32 LAPACK (the reference's torch.cholesky / triangular_solve) reaches on this matrix
    L32 = torch.linalg.cholesky(A + 1e-4 * torch.eye(n))
    T32 = torch.linalg.solve_triangular(L32, torch.eye(n).expand(nb, n, n), upper=False)
    assert rel_l2(L.detach().cpu(), L64.detach()) < 2.0 * rel_l2(L32, L64.detach()) + 1e-6
    assert rel_l2(T.detach().cpu(), T64.detach()) < 2.0 * rel_l2(T32, T64.detach()) + 1e-6
    eye_err = (T.detach().cpu().double() @ L.detach().cpu().double() - torch.eye(n, dtype=torch.float64)).abs().max().item()
    assert eye_err < 1e-4, eye_err
    g64 = 0.5 * (A64.grad + A64.grad.mT)                           # the op returns the symmetric gradient
    assert rel_l2(Ad.grad.cpu(), g64) < 1e-4


# ------------------------------------------------------------------------------------------------------------
# continual chain at Mt = 1000 (Cfg3 task 4 shape in M and Mt; reduced S, C, D, B so that the fp64 oracle runs in seconds)
# ------------------------------------------------------------------------------------------------------------
def test_chain_mt1000_vs_fp64_oracle():
    from vargp_amd import noise
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D, B = 2, 4, 3, 200, 64, 96
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=4, seed=33, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        (10.0 * kl_h + kl_u + 50.0 * nll).backward()
        with torch.no_grad():
            probs = gp.predict(x.to(DEV))
    sc, og = orc.elbo_step(_d(params), [_d(p) for p in prev], x.double(), y, _d(nz), beta=10.0, n_total=50 * B)
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)]:
        np.testing.assert_allclose(v.item(), sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k, g in grads_of(gp).items():
        assert rel_l2(g.cpu(), og[k]) < REL_L2_GRAD, k
    np.testing.assert_allclose(probs.cpu().numpy(), orc.predict(_d(params), [_d(p) for p in prev], x.double(), _d(nz)).numpy(),
                               atol=ATOL_PROBS)


# ------------------------------------------------------------------------------------------------------------
# Cfg3, the LAST task of the 10-task sequence: M = 200, nine earlier tasks, Mt = 2000, D = 784 (full), reduced S, C, B so
# that the fp64 oracle (which walks the reference's nine-fold linear_joint chain, vargp.py:35-88) runs in seconds
# ------------------------------------------------------------------------------------------------------------
def test_chain_mt2000_task9_vs_fp64_oracle():
    from vargp_amd import noise, ops
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D, B = 2, 4, 2, 200, 784, 128
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=9, seed=37, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    assert gp._use_block_program() and len(gp.prev_params) == 9
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        (1.64 * kl_h + kl_u + 50.0 * nll).backward()
        with torch.no_grad():
            pmu, pvar = gp(x.to(DEV))
            probs = gp.predict(x.to(DEV))
    assert ops.linalg_error_count() == 0
    p64, q64 = _d(params), [_d(p) for p in prev]
    sc, og = orc.elbo_step(p64, q64, x.double(), y, _d(nz), beta=1.64, n_total=50 * B)
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)]:
        np.testing.assert_allclose(v.item(), sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k, g in grads_of(gp).items():
        assert rel_l2(g.cpu(), og[k]) < REL_L2_GRAD, k
    m64, v64, _ = orc.forward(p64, q64, x.double(), _d(nz))
    np.testing.assert_allclose(pmu.cpu().numpy(), m64.numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(pvar.cpu().numpy(), v64.numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(probs.cpu().numpy(), orc.predict(p64, q64, x.double(), _d(nz)).numpy(), atol=ATOL_PROBS)


# ------------------------------------------------------------------------------------------------------------
# Cfg5: M = 2048, D = 784 -- the N-tiled ELBO AND its five gradients (vargp_elbo_tn_begin / _tile / _end), even and ragged tiles
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('N,tile', [(1536, 512), (1300, 512)])
def test_tiled_elbo_m2048_vs_fp64_oracle(N, tile):
    from vargp_amd import noise, ops
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D = 1, 4, 2, 2048, 784
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, N, n_prev=0, seed=39, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    with noise.inject(**to_dev(nz, DEV)):
        sc_t = [v.item() for v in gp.elbo_tiled(x.to(DEV), y.to(DEV), tile, beta=10.0, scale=3.0)]
    assert ops.linalg_error_count() == 0
    g_t = {k: v.cpu().clone() for k, v in grads_of(gp).items()}
    sc, og = orc.elbo_step(_d(params), [], x.double(), y, _d(nz), beta=10.0, n_total=3 * N)
    # the oracle's nll is a sum over the N points (likelihoods.py:45-46), as the tiles' sum is
    for v, k in zip(sc_t, ['kl_hypers', 'kl_u', 'nll']):
        np.testing.assert_allclose(v, sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k in g_t:
        assert rel_l2(g_t[k], og[k]) < REL_L2_GRAD, k


# ------------------------------------------------------------------------------------------------------------
# Cfg5: M = 2048, the tiled predictive sweep
# ------------------------------------------------------------------------------------------------------------
def test_predict_tiled_m2048_vs_untiled_and_oracle():
    from vargp_amd import noise
    from gpu_common import build_gp
    S, F_, C, M, D, N = 1, 4, 2, 2048, 784, 1536
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, N, n_prev=0, seed=35, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    xd = x.to(DEV)
    with torch.no_grad(), noise.inject(**to_dev(nz, DEV)):
        whole = gp.predict(xd)
    tile = 512
    parts = []
    with torch.no_grad():      # the sweep draws the likelihood noise per tile: hand each tile its columns
        for i in range(0, N, tile):
            with noise.inject(eps_theta=nz['eps_theta'].to(DEV), eps_f=nz['eps_f'][..., i:i + tile].contiguous().to(DEV)):
                parts.append(gp.predict(xd[i:i + tile]))
    np.testing.assert_allclose(torch.cat(parts).cpu().numpy(), whole.cpu().numpy(), atol=2e-6)

    # predict(tile=) asks for eps_theta once and for eps_f once per tile, in order: hand each request its columns
    from vargp_amd import noise as nmod
    calls = {'i': 0}
    real_draw = nmod.draw

    def draw(name, shape, device, sample_dim=0):
        if name == 'eps_f':
            i = calls['i']
            calls['i'] += shape[-1]
            return nz['eps_f'][..., i:i + shape[-1]].contiguous().to(device)
        return nz[name].to(device)

    nmod.draw = draw
    try:
        with torch.no_grad():
            tiled = gp.predict(xd, tile=tile)
    finally:
        nmod.draw = real_draw
    np.testing.assert_allclose(tiled.cpu().numpy(), whole.cpu().numpy(), atol=2e-6)
    want = orc.predict(_d(params), [], x.double(), _d(nz))
    np.testing.assert_allclose(whole.cpu().numpy(), want.numpy(), atol=ATOL_PROBS)


# ------------------------------------------------------------------------------------------------------------
# Cfg4: S = 64 split over two ranks (sharing cuda:0; gloo carries the flat gradient buffer) == one process with all 64
# ------------------------------------------------------------------------------------------------------------
S_TOTAL, F4, C4, M4, D4, B4 = 64, 10, 10, 100, 784, 512


def _cfg4_model(S):
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.vargp import VARGP
    torch.manual_seed(0)
    xall, yall = mnist_like(4096, D4, C4, kind='gauss', seed=1)
    z = torch.stack([xall[yall == c][:M4] for c in range(C4)])
    gp = VARGP(z, RBFKernel(D4), MulticlassSoftmax(n_f=F4), n_var_samples=S).to(DEV)
    return gp, xall[:B4].to(DEV), yall[:B4].to(DEV)


def _cfg4_worker(rank, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    try:
        from vargp_amd import ops
        from vargp_amd.train import ElboTrainer
        ops.set_cholesky_error_mode('defer')
        gp, x, y = _cfg4_model(S_TOTAL // 2)
        tr = ElboTrainer(gp, lr=1e-3, beta=10.0, n_total=12000, noise_seed=5)
        outs = [[o.item() for o in tr.step(x, y)] for _ in range(2)]
        torch.cuda.synchronize()
        if rank == 0:
            q.put((outs, {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_cfg4_s64_two_rank_split_equals_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg4_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs2, sd2 = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    ops.set_cholesky_error_mode('defer')
    try:
        gp, x, y = _cfg4_model(S_TOTAL)
        tr = ElboTrainer(gp, lr=1e-3, beta=10.0, n_total=12000, noise_seed=5)
        outs1 = [[o.item() for o in tr.step(x, y)] for _ in range(2)]
        sd1 = {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}
    finally:
        noise.clear_shard()
        ops.set_cholesky_error_mode('raise')
    np.testing.assert_allclose(np.array(outs2), np.array(outs1), rtol=1e-4)
    for k in sd1:
        err = np.linalg.norm(sd2[k] - sd1[k]) / max(np.linalg.norm(sd1[k]), 1e-30)
        assert err < 1e-4, (k, err)
