"""GPU: optimiser kernel, hyper-parameter ops, and the captured-graph training step."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import rel_l2, load_case, to_dev

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _yogi_ref(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-3):
    """Zaheer et al. 2018, as implemented by torch_optimizer.Yogi (recalled; parity unpinned)."""
    m = b1 * m + (1 - b1) * g
    g2 = g * g
    v = v - (1 - b2) * torch.sign(v - g2) * g2
    denom = v.sqrt() / np.sqrt(1 - b2 ** t) + eps
    return p - (lr / (1 - b1 ** t)) * m / denom, m, v


def test_yogi_matches_published_algorithm():
    from vargp_amd.optim import Yogi
    torch.manual_seed(0)
    ps = [torch.randn(37, 5), torch.randn(1000), torch.randn(3, 3, 3)]
    params = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    opt = Yogi(params, lr=1e-2)
    ref = [(p.double(), torch.full_like(p, 1e-6).double(), torch.full_like(p, 1e-6).double()) for p in ps]
    for t in range(1, 6):
        gs = [torch.randn_like(p) for p in ps]
        for p, g in zip(params, gs):
            p.grad = g.to(DEV)
        opt.step()
        ref = [_yogi_ref(p, g.double(), m, v, t, 1e-2) for (p, m, v), g in zip(ref, gs)]
    for p, (rp, _, _) in zip(params, ref):
        assert rel_l2(p.detach().cpu(), rp) < 1e-5


def test_hyper_ops():
    from vargp_amd import ops
    D1, S = 785, 3
    m, v, m0, v0 = (orc.hash_normal((D1,), 1).float() * 0.3, -2 + 0.1 * orc.hash_normal((D1,), 2).float(),
                    0.1 * orc.hash_normal((D1,), 3).float(), 0.1 * orc.hash_normal((D1,), 4).float())
    eps = orc.hash_normal((S, D1), 5).float()
    md, vd = m.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    th = ops.hyper_sample(md, vd, eps.to(DEV))
    kl = ops.hyper_kl(md, vd, m0.to(DEV), v0.to(DEV))
    w = orc.hash_normal((S, D1), 6).float()
    ((th * w.to(DEV)).sum() + 3.0 * kl).backward()
    m64, v64 = m.double().requires_grad_(True), v.double().requires_grad_(True)
    th64 = orc.sample_hypers(m64, v64, eps.double())
    kl64 = orc.kl_hypers(m64, v64, m0.double(), v0.double())
    ((th64 * w.double()).sum() + 3.0 * kl64).backward()
    assert rel_l2(th.detach().cpu(), th64.detach()) < 1e-6
    np.testing.assert_allclose(kl.item(), kl64.item(), rtol=1e-5)
    assert rel_l2(md.grad.cpu(), m64.grad) < 1e-5 and rel_l2(vd.grad.cpu(), v64.grad) < 1e-5


def test_graph_step_equals_eager_step():
    """A step replayed from a captured hipGraph produces the same parameters as the eager step."""
    import copy
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    g, params, prev, x, y, nz = load_case('smnist_small_t0')
    xd, yd = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer')
    ops.reset_linalg_errors()
    try:
        results = []
        for mode in ('eager', 'graph'):
            gp = build_gp(params, prev, 3, 10)
            tr = ElboTrainer(gp, lr=1e-3, beta=10.0, n_total=12000)
            with noise.inject(**to_dev(nz, DEV)):
                if mode == 'graph':
                    tr.capture(xd, yd, warmup=2)        # capture() restores parameters / optimiser state / noise stream
                    for _ in range(3):
                        out = tr.step_graph()
                else:
                    for _ in range(3):
                        out = tr.step(xd, yd)
            torch.cuda.synchronize()
            results.append(({k: v.detach().cpu().clone() for k, v in gp.state_dict().items()},
                            [o.item() for o in out]))
        (sd_e, out_e), (sd_g, out_g) = results
        np.testing.assert_allclose(out_g, out_e, rtol=1e-5)
        for k in sd_e:
            assert rel_l2(sd_g[k], sd_e[k]) < 1e-5, k
        assert ops.linalg_error_count() == 0
    finally:
        ops.set_cholesky_error_mode('raise')


def test_graph_step_equals_eager_step_with_a_side_stream():
    """Many hyper-samples (S C + C chains > a third of the CUs): vargp_elbo_t0_fwd / _bwd fork a side stream for the pivot /
    adjoint chains (SideFork, csrc/core.hip).  Under hipGraph capture the fork / join must become graph dependencies: three
    replayed steps == three eager steps; and the first use of the side stream may be under capture (fresh process state is not
    needed: the eager pass below runs second)."""
    from oracle import vargp_oracle as orc
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    S, F_, C, M, D, B = 10, 2, 10, 100, 256, 128          # 110 chains; D >= 256: the Gram matrices built by the chain workgroups
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=5, kind='gauss')
    xd, yd = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer')
    ops.reset_linalg_errors()
    try:
        results = []
        for mode in ('graph', 'eager'):
            gp = build_gp(params, prev, S, F_)
            tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B)
            with noise.inject(**to_dev(nz, DEV)):
                if mode == 'graph':
                    tr.capture(xd, yd, warmup=1)
                    for _ in range(3):
                        out = tr.step_graph()
                else:
                    for _ in range(3):
                        out = tr.step(xd, yd)
            torch.cuda.synchronize()
            results.append(({k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}, [o.item() for o in out]))
        (sd_g, out_g), (sd_e, out_e) = results
        np.testing.assert_allclose(out_g, out_e, rtol=1e-5)
        for k in sd_e:
            assert rel_l2(sd_g[k], sd_e[k]) < 1e-5, k
        assert ops.linalg_error_count() == 0
        # ... and the eager result is the oracle's (one step, same noise)
        gp = build_gp(params, prev, S, F_)
        tr = ElboTrainer(gp, lr=1e-9, beta=2.0, n_total=10 * B)
        with noise.inject(**to_dev(nz, DEV)):
            out = [float(v) for v in tr.step(xd, yd)]
        sc, _ = orc.elbo_step(params, prev, x, y, nz, beta=2.0, n_total=10 * B)
        np.testing.assert_allclose(out, [sc[k].item() for k in ('kl_hypers', 'kl_u', 'nll')], rtol=1e-4)
    finally:
        ops.set_cholesky_error_mode('raise')


@pytest.mark.parametrize('C,M,B,D', [(4, 20, 100, 2), (10, 16, 32, 784), (10, 20, 496, 784)])
def test_graph_replays_stay_finite(C, M, B, D):
    """Regression: every buffer a captured step accumulates into must be re-zeroed by a node of the graph
    (hipMemsetAsync pairs were lost on replay; zeroing is done by a kernel now)."""
    from vargp_amd import ops
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.train import ElboTrainer
    from vargp_amd.vargp import VARGP
    ops.set_cholesky_error_mode('defer')
    try:
        torch.manual_seed(0)
        xall, yall = mnist_like(4096, D, C, kind='gauss', seed=1)
        z = torch.stack([xall[yall == c][:M] for c in range(C)])
        gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=10), n_var_samples=3).to(DEV)
        tr = ElboTrainer(gp, lr=3e-3, beta=10.0, n_total=12000)
        x, y = xall[:B].to(DEV), yall[:B].to(DEV)
        tr.capture(x, y)
        for _ in range(4):
            out = tr.step_graph(x, y)
            torch.cuda.synchronize()
            assert all(torch.isfinite(o).item() for o in out)
            assert all(bool(torch.isfinite(p.grad).all()) for p in gp.parameters())
    finally:
        ops.set_cholesky_error_mode('raise')


def test_graph_survives_ragged_eager_step_and_capture_keeps_state():
    """A captured step, then the ragged last minibatch of an epoch through the eager step (another shape, hence another
    program), then allocations that would reuse a freed workspace, then replays again: the trajectory must equal the
    all-eager one.  (The captured graph holds raw pointers into its program: programs are kept per shape.)  Also:
    capture() itself leaves parameters, optimiser state and the noise stream untouched."""
    from vargp_amd import ops
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.train import ElboTrainer
    from vargp_amd.vargp import VARGP
    C, M, B, D, Br = 4, 24, 96, 40, 56
    ops.set_cholesky_error_mode('defer')
    try:
        xall, yall = mnist_like(2048, D, C, kind='gauss', seed=1)
        x, y = xall[:B].to(DEV), yall[:B].to(DEV)
        xr, yr = xall[B:B + Br].to(DEV), yall[B:B + Br].to(DEV)
        res = []
        for mode in ('eager', 'graph'):
            torch.manual_seed(0)
            z = torch.stack([xall[yall == c][:M] for c in range(C)])
            gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=5), n_var_samples=2).to(DEV)
            tr = ElboTrainer(gp, lr=3e-3, beta=2.0, n_total=2048, noise_seed=11)
            if mode == 'graph':
                before = {k: v.detach().clone() for k, v in gp.state_dict().items()}
                tr.capture(x, y)
                for k, v in gp.state_dict().items():
                    assert torch.equal(v, before[k]), k
            full = (lambda: tr.step_graph(x, y)) if mode == 'graph' else (lambda: tr.step(x, y))
            outs = []
            for epoch in range(2):
                outs.append([o.item() for o in full()])
                outs.append([o.item() for o in full()])
                outs.append([o.item() for o in tr.step(xr, yr)])          # ragged batch: eager, different shape
                junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(8)]   # would land in freed blocks
                del junk
            torch.cuda.synchronize()
            res.append((outs, {k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}))
        np.testing.assert_allclose(np.array(res[1][0]), np.array(res[0][0]), rtol=1e-5)
        for k in res[0][1]:
            assert rel_l2(res[1][1][k], res[0][1][k]) < 1e-5, k
    finally:
        ops.set_cholesky_error_mode('raise')


@pytest.mark.parametrize('n_prev', [0, 1])
def test_epoch_graphs_equal_the_per_step_loop(n_prev):
    """ElboTrainer.capture_epoch / run_epoch -- the minibatch gather inside the graph (vargp_gather_minibatch: batch index read
    from the device-side step count), K steps per launch -- against the per-step loop (index_select + one-step graph per
    minibatch) over the same permutations: the same parameters (to rounding) after two epochs with a ragged tail, for a first-task model
    and a model with one previous task (block program)."""
    from vargp_amd import ops
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.train import ElboTrainer
    from vargp_amd.vargp import VARGP
    C, M, B, D, N = 3, 20, 32, 40, 240          # 7 full minibatches + a ragged one of 16; K = 3: two three-step launches + the one-step rest
    ops.set_cholesky_error_mode('defer')
    try:
        xall, yall = mnist_like(N, D, C, kind='gauss', seed=3)
        data, targets = xall.to(DEV).contiguous(), yall.to(DEV).contiguous()
        gen = torch.Generator().manual_seed(5)
        perms = [torch.randperm(N, generator=gen).to(DEV) for _ in range(2)]
        res = []
        for mode in ('per_step', 'epoch_graphs'):
            torch.manual_seed(0)
            z = torch.stack([xall[yall == c][:M] for c in range(C)])
            prev = []
            if n_prev:
                zp = torch.stack([xall[yall == c][M:2 * M] for c in range(C)])
                prev = [dict(z=zp.to(DEV), u_mean=0.1 * torch.randn(C, M, 1).to(DEV),
                             u_tril_vec=(0.05 * torch.randn(C, M * (M + 1) // 2)).to(DEV))]
            gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=4), n_var_samples=2, prev_params=prev).to(DEV)
            tr = ElboTrainer(gp, lr=3e-3, beta=2.0, n_total=N, noise_seed=17)
            tr.capture(data[:B], targets[:B])
            if mode == 'epoch_graphs':
                assert tr.capture_epoch(data, targets, k=3) is tr
            tr.capture(data[:N % B], targets[:N % B])
            for perm in perms:
                if mode == 'epoch_graphs':
                    out, done = tr.run_epoch(perm)
                    assert done == N // B
                else:
                    for i in range(N // B):
                        out = tr.step_graph_gather(data, targets, perm[i * B:(i + 1) * B])
                out = tr.step_graph_gather(data, targets, perm[(N // B) * B:])
            torch.cuda.synchronize()
            assert ops.linalg_error_count() == 0
            res.append(([o.item() for o in out], {k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}))
        # (same kernels on the same minibatches in the same order; float atomics inside them make two runs of EITHER loop agree to
        # rounding, not to the bit)
        np.testing.assert_allclose(np.array(res[1][0]), np.array(res[0][0]), rtol=1e-5)
        for k in res[0][1]:
            assert rel_l2(res[1][1][k], res[0][1][k]) < 1e-5, k
            assert torch.isfinite(res[0][1][k]).all()
    finally:
        ops.set_cholesky_error_mode('raise')


@pytest.mark.parametrize('shape,map_est', [((3, 3, 4, 56, 784, 64), False), ((2, 3, 3, 24, 40, 32), False),
                                          ((1, 3, 3, 56, 300, 64), True)])
def test_deferred_hyper_backward_equals_its_own_launch(shape, map_est, monkeypatch):
    """The trainer finishes the hyper-parameter backward inside the optimiser's launch (vargp_yogi_step_multi_hyper); with
    VARGP_DEFER_HYPER=0 the program's own last kernel (t0_hyper_bwd_kernel) does it before a plain Yogi step.  Same
    gradients of log_mean / log_logvar and the same parameters after three steps -- D + 1 = 785 > 256 (several blocks of
    the hyper role), a small D, and a MAP estimate of the hyper-parameters (log_logvar receives no gradient)."""
    from vargp_amd import noise
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    S, F_, C, M, D, B = shape
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, seed=21, kind='gauss')
    xd, yd = x.to(DEV), y.to(DEV)
    res = []
    for defer in ('1', '0'):
        monkeypatch.setenv('VARGP_DEFER_HYPER', defer)
        gp = build_gp(params, prev, S, F_)
        gp.kernel.map_est = map_est
        tr = ElboTrainer(gp, lr=3e-3, beta=2.0, n_total=10 * B)
        assert tr._t0 and tr._defer_hyper() == (defer == '1')
        for it in range(3):
            inj = dict(eps_f=(nz['eps_f'] + 0.1 * it).to(DEV))
            if not map_est:
                inj['eps_theta'] = (nz['eps_theta'] - 0.05 * it).to(DEV)
            with noise.inject(**inj):
                tr.step(xd, yd)
        k = gp.kernel
        res.append(dict(g_mean=k.log_mean.grad.clone().cpu(), g_logvar=None if map_est else k.log_logvar.grad.clone().cpu(),
                        params={n: p.detach().clone().cpu() for n, p in gp.named_parameters()}))
    a, b = res
    assert rel_l2(a['g_mean'], b['g_mean']) < 1e-5
    if not map_est:
        assert rel_l2(a['g_logvar'], b['g_logvar']) < 1e-5
    for n in a['params']:
        assert rel_l2(a['params'][n], b['params'][n]) < 1e-5, n


def test_defer_hyper_needs_both_hyper_tensors_in_the_optimiser():
    """with log_logvar frozen (not among the optimiser's parameters) and no MAP estimate, the deferred launch
    would be handed idx_logvar = -1 and reject it; _defer_hyper() must say no and the step must run."""
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    S, F_, C, M, D, B = 2, 3, 3, 24, 40, 32
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, seed=22, kind='gauss')
    gp = build_gp(params, prev, S, F_)
    gp.kernel.log_logvar.requires_grad_(False)
    tr = ElboTrainer(gp, lr=3e-3, beta=1.0, n_total=B)
    assert not tr._defer_hyper()
    out = tr.step(x.to(DEV), y.to(DEV))
    assert all(torch.isfinite(v) for v in out)
