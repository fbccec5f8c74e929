"""GPU: the route bench.py TIMES, at the shapes it times.

The metric's step is `ElboTrainer.step_graph` with the trainer's defaults: the native program driven directly (no autograd
node), the likelihood deferred into the backward's tile kernel (`defer_softmax`), the hyper-parameter backward finished inside
the Yogi launch (`defer_hyper`), one hipGraph per step.  The full-size parity tests of test_hip_e2e.py go through
`VARGP.loss` as an autograd node; here the trainer itself -- eager and captured -- is held, at BASELINE config 2's real shape
(S3 F10 C10 M100 D784 B512), at Permuted-MNIST task 0 (S10 M200: the block program) and at Split-MNIST task 1 (Mt = 200),
to (a) the loss triple the REFERENCE produced on these inputs (tests/golden/*.npz, rtol 1e-4), (b) every gradient tensor of
the oracle on the same inputs and noise (rel-L2 1e-3; read from the trainer's gradient buffers after the step -- the optimiser
launch is what finishes the two hyper-parameter gradients on the deferred route), and (c) a six-step reference trajectory in
the reference's loop shape (experiments/vargp.py:29-37) with a stock Adam through eager steps and graph replays."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import load_case, rel_l2, to_dev, RTOL_SCALAR, REL_L2_GRAD, GRAD_KEYS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

_ORACLE = {}


def _oracle_grads(name):
    """orc.elbo_step of a fixture's inputs, once per session (the eager and graph variants share it)."""
    if name not in _ORACLE:
        g, params, prev, x, y, nz = load_case(name)
        _ORACLE[name] = orc.elbo_step(params, prev, x, y, nz, beta=float(g['beta']), n_total=float(g['n_total']))
    return _ORACLE[name]


def _grads(gp):
    return dict(z=gp.z.grad, u_mean=gp.u_mean.grad, u_tril_vec=gp.u_tril_vec.grad, log_mean=gp.kernel.log_mean.grad,
                log_logvar=gp.kernel.log_logvar.grad)


def _trainer_step(name, graph, optimizer=None):
    """One step of ElboTrainer (its defaults) on the fixture's inputs with the fixture's noise injected ->
    (g, trainer, triple, gradients as left in the trainer's buffers by that step)."""
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    g, params, prev, x, y, nz = load_case(name)
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    gp = build_gp(params, prev, S, F_)
    xd, yd = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer' if graph else 'raise')
    try:
        # lr tiny: the step's parameter update must not be what is measured; the gradients are read from the buffers
        tr = ElboTrainer(gp, lr=1e-9, beta=float(g['beta']), n_total=float(g['n_total']), optimizer=optimizer)
        nzd = {k: v.contiguous() for k, v in to_dev(nz, DEV).items()}
        with noise.inject(**nzd):
            if graph:
                tr.capture(xd, yd)
                out = tr.step_graph()
            else:
                out = tr.step(xd, yd)
            torch.cuda.synchronize()
            triple = [float(v) for v in out]
        grads = {k: v.detach().clone().cpu() for k, v in _grads(gp).items()}
        if graph:
            assert ops.linalg_error_count() == 0      # 'defer' mode: the factorisations' info words, read back once
    finally:
        ops.set_cholesky_error_mode('raise')
    return g, tr, triple, grads


FULL = ['smnist_full_t0', 'pmnist_full_t0', 'smnist_full_t1', 'smnist_full_t0_mnist_l25']


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
@pytest.mark.parametrize('name', FULL)
def test_trainer_route_full_size_vs_reference_golden_and_oracle(name, graph):
    g, tr, triple, grads = _trainer_step(name, graph)
    # the route under test is the one the bench times: native program, deferred likelihood (first-task program), deferred
    # hyper-parameter backward inside the Yogi launch
    assert tr._t0 and tr._defer_hyper()
    n_prev = int(g['meta'][6])
    M = int(g['meta'][3])
    assert tr._tn == (n_prev > 0 or M > 104), (name, tr._tn)
    for k, v in zip(['kl_hypers', 'kl_u', 'nll'], triple):
        np.testing.assert_allclose(v, float(g[k]), rtol=RTOL_SCALAR, err_msg=f'{name} {k} vs the reference')
    total = float(g['beta']) * triple[0] + triple[1] + float(g['n_total']) / int(g['meta'][5]) * triple[2]
    np.testing.assert_allclose(total, float(g['total']), rtol=RTOL_SCALAR)
    # gradient checks against the reference's own outputs stored in the fixture ...
    for k in GRAD_KEYS:
        np.testing.assert_allclose(grads[k].double().norm().item(), float(g[f'gradnorm_{k}']), rtol=1e-3, err_msg=k)
    assert rel_l2(grads['log_mean'], g['grad_log_mean']) < REL_L2_GRAD
    assert rel_l2(grads['u_mean'], g['grad_u_mean']) < REL_L2_GRAD
    assert rel_l2(grads['z'][:, :4, :], g['grad_z_head']) < REL_L2_GRAD
    # ... and, tensor by tensor, against the oracle on the same inputs and noise
    _, og = _oracle_grads(name)
    for k in GRAD_KEYS:
        assert rel_l2(grads[k], og[k]) < REL_L2_GRAD, (name, k)


def test_trainer_route_sgd_parameter_delta_is_the_oracle_gradient():
    """The same through a stock optimiser: one plain-SGD step of size lr moves every parameter by -lr * gradient, so the
    one-step displacement of the parameters themselves (what a training run sees) is the oracle's gradient."""
    name, lr = 'smnist_full_t0', 1e-3
    from gpu_common import build_gp
    g, params, prev, x, y, nz = load_case(name)
    g, tr, triple, grads = _trainer_step(name, graph=True, optimizer=lambda ps: torch.optim.SGD(ps, lr=lr))
    assert not tr._defer_hyper()                      # a stock optimiser: the program's own last kernel finishes the gradients
    _, og = _oracle_grads(name)
    gp = tr.gp
    now = dict(z=gp.z, u_mean=gp.u_mean, u_tril_vec=gp.u_tril_vec, log_mean=gp.kernel.log_mean, log_logvar=gp.kernel.log_logvar)
    for k in GRAD_KEYS:
        delta = (params[k].double() - now[k].detach().cpu().double().view_as(params[k])) / lr
        # (the displacement is a difference of fp32 parameters: for z, |z| ~ 0.5 and |lr g| ~ 1e-6 leave 2-3 digits)
        tol = 5e-2 if k == 'z' else 5e-3
        assert rel_l2(delta, og[k]) < tol, (k, rel_l2(delta, og[k]))


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_trainer_route_full_size_reference_trajectory(graph):
    """Six Adam steps of the reference at BASELINE config 2's real shape (tests/golden/traj_full_t0.npz) through the trainer:
    every step's loss triple within 1e-4, the final parameters within 1e-3."""
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    from test_oracle_golden import load_trajectory, check_compact_final
    g, params, prev, x, y, noise_of, steps, lr, beta, n_total = load_trajectory('traj_full_t0')
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    gp = build_gp(params, prev, S, F_)
    x, y = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer' if graph else 'raise')
    try:
        tr = ElboTrainer(gp, beta=beta, n_total=n_total, optimizer=lambda ps: torch.optim.Adam(ps, lr=lr, capturable=graph))
        triples = []
        if graph:
            nz0 = {kk: v.to(DEV).clone() for kk, v in noise_of(0).items()}
            with noise.inject(**nz0):
                tr.capture(x, y)
                for k in range(steps):
                    for kk, v in noise_of(k).items():
                        nz0[kk].copy_(v)
                    triples.append([float(v) for v in tr.step_graph()])
        else:
            for k in range(steps):
                with noise.inject(**{kk: v.to(DEV) for kk, v in noise_of(k).items()}):
                    triples.append([float(v) for v in tr.step(x, y)])
    finally:
        ops.set_cholesky_error_mode('raise')
    rel = np.abs(np.array(triples) - g['triples']) / np.abs(g['triples'])
    assert rel.max() < 1e-4, rel.max(axis=1)
    final = {k: v.detach().cpu() for k, v in dict(z=gp.z, u_mean=gp.u_mean, u_tril_vec=gp.u_tril_vec,
                                                   log_mean=gp.kernel.log_mean, log_logvar=gp.kernel.log_logvar).items()}
    check_compact_final(final, g, 1e-3)
    for k, v in final.items():
        d = (v.double() - params[k].double().view_as(v)).norm().item()
        np.testing.assert_allclose(d, float(g[f'deltanorm_{k}']), rtol=5e-3, err_msg=k)


def test_trainer_routes_by_batch_size_like_loss():
    """the trainer picks its native program from the minibatch it is given, as VARGP.loss does (a first-task model
    leaves the LDS-resident program when S C ceil(B / 64) exceeds VARGP.T0_TILE_UNITS_MAX = 16384 tile units)."""
    from vargp_amd import noise
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    S, F_, C, M, D = 16, 2, 16, 32, 40
    for B, want_tn in ((64, False), (4160, True)):
        params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=4, kind='gauss')
        gp = build_gp(params, prev, S, F_)
        assert gp._use_block_program(B) == want_tn
        tr = ElboTrainer(gp, lr=1e-9)
        with noise.inject(**to_dev(nz, DEV)):
            out = [float(v) for v in tr.step(x.to(DEV), y.to(DEV))]
        assert tr._tn == want_tn
        sc, og = orc.elbo_step(params, prev, x, y, nz)
        np.testing.assert_allclose(out, [sc[k].item() for k in ('kl_hypers', 'kl_u', 'nll')], rtol=RTOL_SCALAR)
        for k, v in _grads(gp).items():
            assert rel_l2(v.cpu(), og[k]) < REL_L2_GRAD, (B, k)


def test_k_step_graph_equals_single_step_graphs():
    """bench.py times K steps per hipGraph launch (ElboTrainer.capture_unrolled): the same K steps as K launches of the one-step
    graph -- same device-side noise stream (counter-based generator), same optimiser step counts -- at BASELINE config 2's shape."""
    from vargp_amd import ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    g, params, prev, x, y, nz = load_case('smnist_full_t0')
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    xd, yd = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer')
    try:
        res = []
        for k in (1, 3):
            gp = build_gp(params, prev, S, F_)
            tr = ElboTrainer(gp, lr=3e-3, beta=float(g['beta']), n_total=float(g['n_total']), noise_seed=99)
            tr.capture(xd, yd)
            if k > 1:
                tr.capture_unrolled(xd, yd, k)
                outs = [[float(v) for v in tr.step_graph_k()] for _ in range(6 // k)]
            else:
                outs = [[float(v) for v in tr.step_graph()] for _ in range(6)]
            torch.cuda.synchronize()
            res.append((outs[-1], {n: p.detach().cpu().clone() for n, p in gp.named_parameters()}))
        assert ops.linalg_error_count() == 0
    finally:
        ops.set_cholesky_error_mode('raise')
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    for n in res[0][1]:
        assert rel_l2(res[1][1][n], res[0][1][n]) < 1e-5, n
