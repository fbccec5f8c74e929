"""GPU: the native block-structured ELBO program for models with previous tasks (vargp_elbo_tn_fwd / _bwd,
csrc/elbo_tn.hip) against the fp64 oracle (which follows the reference's linear_joint chain), against the composed
per-op path (the reference's op-by-op structure on the same HIP kernels), as a first-task program (nblk = 1) against
the first-task program, in its forward-only (predict) mode, and driven directly by the trainer."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import rel_l2, to_dev, RTOL_SCALAR, REL_L2_GRAD, ATOL_PROBS, ATOL_PRED, RTOL_PRED

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _d(o):
    return {k: v.double() for k, v in o.items()}


# (S, F, C, M, D, B, n_prev): M % 4 != 0 (unaligned diagonal blocks), B % 4 != 0, C > 16 (generic softmax kernels),
# D <= 32 (direct distance form), Mt <= 100 (register-resident factorisation) and Mt > 100 (blocked), one to four
# earlier tasks, tiny / degenerate sizes
SHAPES = [(2, 3, 3, 12, 40, 48, 1), (2, 3, 3, 33, 40, 50, 2), (3, 2, 4, 20, 64, 37, 1), (2, 2, 18, 12, 36, 24, 2),
          (2, 3, 2, 20, 2, 64, 3), (1, 2, 2, 60, 48, 40, 4), (1, 1, 1, 3, 33, 2, 1), (2, 2, 2, 100, 36, 52, 1),
          (2, 2, 3, 50, 24, 32, 1),
          # blocked factorisation with a last panel narrower than 50 (Mt = 144: 100 + 44), four panels (Mt = 312: 3 x 100 + 12),
          # blocks wider than a panel (M = 120, Mt = 240: the task blocks and the 100-wide panels do not line up)
          (2, 2, 2, 36, 40, 33, 3), (1, 2, 2, 104, 36, 40, 2), (2, 1, 2, 120, 48, 28, 1)]


@pytest.mark.parametrize('shape', SHAPES, ids=[str(s) for s in SHAPES])
def test_program_matches_fp64_oracle(shape):
    from vargp_amd import noise
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D, B, n_prev = shape
    kind = 'wtoy' if D == 2 else 'gauss'
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=41, kind=kind)
    gp = build_gp(params, prev, S, F_)
    assert gp._tn_applicable()
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        (2.0 * kl_h + kl_u + 7.0 * nll).backward()
        with torch.no_grad():
            pmu, pvar = gp(x.to(DEV))
            probs = gp.predict(x.to(DEV))
    sc, og = orc.elbo_step(_d(params), [_d(p) for p in prev], x.double(), y, _d(nz), beta=2.0, n_total=7 * B)
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)]:
        np.testing.assert_allclose(v.item(), sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k, g in grads_of(gp).items():
        assert rel_l2(g.cpu(), og[k]) < REL_L2_GRAD, k
    m64, v64, _ = orc.forward(_d(params), [_d(p) for p in prev], x.double(), _d(nz))
    np.testing.assert_allclose(pmu.cpu().numpy(), m64.numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(pvar.cpu().numpy(), v64.numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(probs.cpu().numpy(), orc.predict(_d(params), [_d(p) for p in prev], x.double(), _d(nz)).numpy(),
                               atol=ATOL_PROBS)


@pytest.mark.parametrize('name', ['wtoy_t2', 'smnist_small_t1', 'pmnist_red_t1'])
def test_program_equals_composed_path(name):
    """Same inputs through the program and through the reference-shaped composition of per-op kernels
    (compute_q's linear_joint chain, linear_marginal_diag, the conditional prior + MVN KL)."""
    from vargp_amd import noise
    from gpu_common import build_gp, grads_of
    from helpers import load_case, GRAD_KEYS
    g, params, prev, x, y, nz = load_case(name)
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    res = []
    for fused in (True, False):
        gp = build_gp(params, prev, S, F_)
        gp.fused_tasks = fused
        with noise.inject(**to_dev(nz, DEV)):
            kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
            (float(g['beta']) * kl_h + kl_u + (float(g['n_total']) / x.shape[0]) * nll).backward()
            with torch.no_grad():
                probs = gp.predict(x.to(DEV))
        res.append(((kl_h.item(), kl_u.item(), nll.item()), {k: v.cpu() for k, v in grads_of(gp).items()}, probs.cpu()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-5)
    for k in GRAD_KEYS:
        assert rel_l2(res[0][1][k], res[1][1][k]) < 2e-4, k
    np.testing.assert_allclose(res[0][2].numpy(), res[1][2].numpy(), atol=2e-5)


@pytest.mark.parametrize('shape', [(2, 3, 3, 40, 48, 64), (2, 2, 2, 200, 64, 96)])
def test_program_single_block_equals_first_task_program(shape):
    """nblk = 1: the same program is a first-task ELBO for any M; against the first-task program (csrc/elbo_t0.hip)."""
    from vargp_amd import fused
    from gpu_common import build_gp
    S, F_, C, M, D, B = shape
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=43, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    k = gp.kernel
    xd, yd = x.to(DEV), y.to(DEV)
    et, ef = nz['eps_theta'].to(DEV), nz['eps_f'].to(DEV)
    seeds = torch.tensor([2.0, 1.0, 5.0], device=DEV)
    args = (k.log_mean.detach(), k.log_logvar.detach(), k.prior_log_mean, k.prior_log_logvar, gp.z.detach(),
            gp.u_mean.detach(), gp.u_tril_vec.detach())
    out = []
    for which in ('t0', 'tn'):
        grads = [torch.empty_like(t) for t in (k.log_mean, k.log_logvar, gp.z, gp.u_mean, gp.u_tril_vec)]
        if which == 't0':
            prog = fused.T0Program(S, C, M, D, B, F_, DEV)
            sc = prog.forward(*args, xd, yd, et, ef).clone()
        else:
            prog = fused.TnProgram(S, C, M, D, B, F_, 1, DEV)
            sc = prog.forward(*args, *gp._tn_operands(), xd, yd, et, ef).clone()
        prog.backward(seeds, *grads)
        out.append((sc.cpu(), [t.cpu() for t in grads]))
    np.testing.assert_allclose(out[1][0].numpy(), out[0][0].numpy(), rtol=2e-5)
    for a, b, name in zip(out[1][1], out[0][1], ['log_mean', 'log_logvar', 'z', 'u_mean', 'u_tril_vec']):
        assert rel_l2(a, b) < 2e-4, name


def test_trainer_direct_program_equals_autograd_route_t1():
    """ElboTrainer drives TnProgram directly; the same steps through gp.loss -> autograd -> Yogi give the same parameters.
    Also: hipGraph replay of the step == eager."""
    from vargp_amd import noise, ops
    from vargp_amd.optim import Yogi
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    S, F_, C, M, D, B = 2, 3, 4, 24, 48, 64
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=2, seed=8, kind='gauss')
    xd, yd = x.to(DEV), y.to(DEV)
    beta, n_total = 1.5, 10 * B
    gp_a, gp_b = build_gp(params, prev, S, F_), build_gp(params, prev, S, F_)
    tr = ElboTrainer(gp_a, lr=3e-3, beta=beta, n_total=n_total)
    assert tr._tn
    opt = Yogi([p for p in gp_b.parameters() if p.requires_grad], lr=3e-3)
    for it in range(3):
        nzd = {k: (v + 0.1 * it).to(DEV) for k, v in nz.items()}
        with noise.inject(**nzd):
            out_a = [float(v) for v in tr.step(xd, yd)]
            for p in gp_b.parameters():
                p.grad = None
            kl_h, kl_u, nll = gp_b.loss(xd, yd)
            (beta * kl_h + kl_u + (n_total / B) * nll).backward()
            opt.step()
        np.testing.assert_allclose(out_a, [kl_h.item(), kl_u.item(), nll.item()], rtol=2e-5)
    # (K-split products accumulate with float atomics: run-to-run differences of a few ulp, amplified by Yogi's 1/sqrt(v))
    for (k, pa), (_, pb) in zip(gp_a.named_parameters(), gp_b.named_parameters()):
        assert rel_l2(pa.detach().cpu(), pb.detach().cpu()) < 5e-5, ('direct vs autograd', k)

    # graph replay == eager (native noise: same seed, same counter)
    ops.set_cholesky_error_mode('defer')
    ops.reset_linalg_errors()
    try:
        res = []
        for mode in ('eager', 'graph'):
            gp = build_gp(params, prev, S, F_)
            tr = ElboTrainer(gp, lr=3e-3, beta=beta, n_total=n_total, noise_seed=3)
            if mode == 'graph':
                tr.capture(xd, yd)
            for _ in range(3):
                out = tr.step_graph(xd, yd) if mode == 'graph' else tr.step(xd, yd)
            torch.cuda.synchronize()
            res.append(([o.item() for o in out], {k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}))
        np.testing.assert_allclose(res[1][0], res[0][0], rtol=2e-5)
        for k in res[0][1]:
            assert rel_l2(res[1][1][k], res[0][1][k]) < 5e-5, ('graph vs eager', k)
        assert ops.linalg_error_count() == 0
    finally:
        ops.set_cholesky_error_mode('raise')


def test_program_non_pd_is_flagged():
    """Duplicate inducing points across tasks with zero jitter would break the factorisation; with the reference's jitter
    the matrix stays PD.  A NaN input must be flagged (info != 0) and NaN-poison the result, not crash."""
    from vargp_amd import ops
    from gpu_common import build_gp
    S, F_, C, M, D, B = 1, 2, 2, 16, 8, 16
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=1, seed=9, kind='gauss')
    params['z'][0, 0, 0] = float('nan')
    gp = build_gp(params, prev, S, F_)
    ops.set_cholesky_error_mode('raise')
    with pytest.raises(torch.linalg.LinAlgError):
        gp.loss(x.to(DEV), y.to(DEV))
    # 'lazy': no synchronisation inside loss(); the error surfaces at the first host read of a value of that step ...
    ops.set_cholesky_error_mode('lazy')
    try:
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        with pytest.raises(torch.linalg.LinAlgError, match='lazily'):
            nll.item()
        # ... or at the next factorising call, or on request
        gp.loss(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        with pytest.raises(torch.linalg.LinAlgError, match='lazily'):
            ops.check_linalg_errors()
    finally:
        ops.set_cholesky_error_mode('raise')


@pytest.mark.parametrize('n_prev,M,N,tile', [(0, 40, 200, 64), (2, 24, 150, 64), (0, 130, 96, 96)])
def test_tiled_elbo_equals_untiled_and_oracle(n_prev, M, N, tile):
    """vargp_elbo_tn_begin / _tile / _end (loss and gradient over a data set swept in minibatch tiles, ragged last tile
    included) == the one-minibatch program on all N points == the fp64 oracle."""
    from vargp_amd import noise
    from gpu_common import build_gp, grads_of
    S, F_, C, D = 2, 3, 3, 40
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, N, n_prev=n_prev, seed=47, kind='gauss')
    xd, yd = x.to(DEV), y.to(DEV)
    gp = build_gp(params, prev, S, F_)
    with noise.inject(**to_dev(nz, DEV)):
        sc_t = [v.item() for v in gp.elbo_tiled(xd, yd, tile, beta=2.0, scale=3.0)]
    g_t = {k: v.cpu().clone() for k, v in grads_of(gp).items()}
    sc, og = orc.elbo_step(_d(params), [_d(p) for p in prev], x.double(), y, _d(nz), beta=2.0, n_total=3 * N)
    for v, k in zip(sc_t, ['kl_hypers', 'kl_u', 'nll']):
        np.testing.assert_allclose(v, sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k in g_t:
        assert rel_l2(g_t[k], og[k]) < REL_L2_GRAD, k
    # native noise path runs and is finite
    gp2 = build_gp(params, prev, S, F_)
    out = gp2.elbo_tiled(xd, yd, tile, noise_seed=5)
    assert all(torch.isfinite(v).item() for v in out) and all(bool(torch.isfinite(g).all()) for g in grads_of(gp2).values())


# ------------------------------------------------------------------------------------------------------------
# evaluation on the block program: predict(x, tile=) of models with previous tasks (reference: VARGP.predict
# vargp.py:196-198 called per batch by compute_accuracy, train_utils.py:21-35)
# ------------------------------------------------------------------------------------------------------------
def _draw_by_columns(nz):
    """noise.draw stand-in: eps_theta whole, eps_f handed out in column order (the tiled sweep asks once per tile)."""
    calls = {'i': 0}

    def draw(name, shape, device, sample_dim=0):
        if name == 'eps_f':
            i = calls['i']
            calls['i'] += shape[-1]
            return nz['eps_f'][..., i:i + shape[-1]].contiguous().to(device)
        return nz[name].to(device)
    return draw


@pytest.mark.parametrize('n_prev,M,N,tile', [(1, 24, 150, 64), (3, 40, 200, 96), (0, 130, 100, 32)])
def test_predict_tiled_block_program_equals_untiled_and_oracle(n_prev, M, N, tile):
    from vargp_amd import noise as nmod
    from gpu_common import build_gp
    S, F_, C, D = 2, 3, 3, 40
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, N, n_prev=n_prev, seed=51, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    xd = x.to(DEV)
    with torch.no_grad(), nmod.inject(**to_dev(nz, DEV)):
        whole = gp.predict(xd)
    real_draw = nmod.draw
    nmod.draw = _draw_by_columns(nz)
    try:
        with torch.no_grad():
            tiled = gp.predict(xd, tile=tile)
    finally:
        nmod.draw = real_draw
    assert gp._tn_eval is not None and gp._tn_eval.forward_only           # forward-only programs only (no gradient buffers)
    assert gp._tn_eval.shape[4] >= tile
    np.testing.assert_allclose(tiled.cpu().numpy(), whole.cpu().numpy(), atol=2e-6)
    want = orc.predict(_d(params), [_d(p) for p in prev], x.double(), _d(nz))
    np.testing.assert_allclose(tiled.cpu().numpy(), want.numpy(), atol=ATOL_PROBS)


def test_eval_program_serves_narrower_batches_and_compute_accuracy_syncs_once():
    """One forward-only workspace per model: a ragged last batch runs on the program carved for the full batch (tile
    calls), with the same moments as an exact-shape evaluation; compute_accuracy (per-batch and shared-hyper sweeps) agrees
    with a plain loop."""
    from torch.utils.data import TensorDataset
    from vargp_amd import noise as nmod
    from vargp_amd.train_utils import compute_accuracy
    from gpu_common import build_gp
    S, F_, C, M, D, N = 2, 3, 3, 20, 40, 150
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, N, n_prev=2, seed=53, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    xd = x.to(DEV)
    et = nz['eps_theta'].to(DEV)
    with torch.no_grad():
        with nmod.inject(eps_theta=et):
            mu64, var64 = gp(xd[:64])                       # carves the program for B = 64
        prog = gp._tn_eval
        with nmod.inject(eps_theta=et):
            mu22, var22 = gp(xd[128:150])                   # 22 < 64: served by the same program
        assert gp._tn_eval is prog
    m64, v64, _ = orc.forward(_d(params), [_d(p) for p in prev], x.double(), _d(nz))
    np.testing.assert_allclose(mu64.cpu().numpy(), m64[..., :64].numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(mu22.cpu().numpy(), m64[..., 128:150].numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    np.testing.assert_allclose(var22.cpu().numpy(), v64[..., 128:150].numpy(), rtol=RTOL_PRED, atol=ATOL_PRED)
    # accuracy: map_est-like determinism is not available, so compare the two sweep modes against a loop under fixed noise
    ds = TensorDataset(x, y)

    def fixed_draw(name, shape, device, sample_dim=0):
        return torch.zeros(shape, device=device)
    real_draw = nmod.draw
    nmod.draw = fixed_draw
    try:
        with torch.no_grad():
            want = (gp.predict(xd).argmax(-1).cpu() == y).float().mean().item()
        a = compute_accuracy(ds, gp, batch_size=64, device=DEV)
        b = compute_accuracy(ds, gp, batch_size=64, device=DEV, shared_hypers=True)
    finally:
        nmod.draw = real_draw
    assert abs(a - want) < 1e-6 and abs(b - want) < 1e-6, (a, b, want)


def test_loss_without_backward_does_not_leak_programs():
    """A validation ELBO under no_grad, or a loss whose graph is dropped, must hand the cached program back (ADVICE r2:
    busy was only cleared by backward); two losses combined before one backward still work (a cached spare)."""
    from vargp_amd import noise
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D, B = 2, 3, 3, 16, 40, 32
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=1, seed=55, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    xd, yd = x.to(DEV), y.to(DEV)
    with noise.inject(**to_dev(nz, DEV)):
        with torch.no_grad():
            gp.loss(xd, yd)
        prog = next(iter(gp._tn_progs.values()))
        assert not prog.busy
        out = gp.loss(xd, yd)                # graph recorded, then dropped without a backward
        assert prog.busy
        del out
        assert not prog.busy
        for _ in range(3):
            gp.loss(xd, yd)                  # results dropped at once
        assert len(gp._tn_progs) == 1 and not gp._tn_spares
        a = gp.loss(xd, yd)
        b = gp.loss(xd, yd)                  # second node while the first still owns the workspace -> spare
        assert sum(len(v) for v in gp._tn_spares.values()) == 1
        (a[1] + a[2] + b[1] + b[2] + a[0]).backward()
        g2 = {k: v.clone() for k, v in grads_of(gp).items()}
        for p in gp.parameters():
            p.grad = None
        c = gp.loss(xd, yd)
        (2.0 * (c[1] + c[2]) + c[0]).backward()
        assert sum(len(v) for v in gp._tn_spares.values()) == 1     # the spare is cached, not re-allocated
    for k, g in grads_of(gp).items():
        assert rel_l2(g2[k], g) < 1e-5, k


@pytest.mark.parametrize('shape', [(3, 2, 3, 24, 40, 48, 1), (2, 3, 2, 56, 36, 64, 2), (5, 2, 2, 100, 64, 128, 1)],
                         ids=['t1', 't2', 't1_M100'])
def test_ep_var_mean_false_runs_on_the_block_program(shape):
    """ep_var_mean=False (reference vargp.py:137-152: the KL keeps the conditional prior's mean at n_v samples of u_<t) is a
    variant of the SAME native program (tn_nm_* kernels), not the op-by-op composition: loss triple and all five gradients
    against the fp64 oracle, through VARGP.loss and through the trainer's direct use of the program."""
    from vargp_amd import noise
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp, grads_of
    S, F_, C, M, D, B, n_prev = shape
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=61, kind='gauss')
    assert nz['eps_u'].shape == (S, S, C, n_prev * M)
    d64 = lambda o: {k: v.double() for k, v in o.items()}
    sc, og = orc.elbo_step(d64(params), [d64(p) for p in prev], x.double(), y, d64(nz), beta=2.0, n_total=7 * B, ep_var_mean=False)
    sc1, _ = orc.elbo_step(d64(params), [d64(p) for p in prev], x.double(), y, d64(nz), beta=2.0, n_total=7 * B, ep_var_mean=True)
    assert abs(sc['kl_u'].item() - sc1['kl_u'].item()) > 1e-3 * abs(sc1['kl_u'].item())        # the mean term is not negligible here
    gp = build_gp(params, prev, S, F_, ep_var_mean=False)
    assert gp._use_block_program(B) and gp.var_mean_mask == 0.0
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        (2.0 * kl_h + kl_u + 7.0 * nll).backward()
    assert len(gp._tn_progs) == 1 and next(iter(gp._tn_progs.values())).desc.no_var_mean == 1
    for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)]:
        np.testing.assert_allclose(v.item(), sc[k].item(), rtol=RTOL_SCALAR, err_msg=k)
    for k, g in grads_of(gp).items():
        assert rel_l2(g.cpu(), og[k]) < REL_L2_GRAD, k
    # the trainer route (program driven directly, gradients into the optimiser's buffers)
    gp2 = build_gp(params, prev, S, F_, ep_var_mean=False)
    tr = ElboTrainer(gp2, lr=1e-9, beta=2.0, n_total=7 * B)
    assert tr._t0 and tr._tn
    with noise.inject(**to_dev(nz, DEV)):
        out = [float(v) for v in tr.step(x.to(DEV), y.to(DEV))]
    np.testing.assert_allclose(out, [sc[k].item() for k in ('kl_hypers', 'kl_u', 'nll')], rtol=RTOL_SCALAR)
    for k, g in grads_of(gp2).items():
        assert rel_l2(g.cpu(), og[k]) < REL_L2_GRAD, ('trainer', k)
