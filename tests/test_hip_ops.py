"""GPU: each C-ABI op (forward and hand-written backward) against a CPU reference computed in
fp64 by torch autograd on the oracle's formulas.  Tolerances are fp32-level and written per test."""
import numpy as np
import pytest
import torch

from oracle import vargp_oracle as orc
from helpers import rel_l2, GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hn(shape, seed, scale=1.0):
    return (scale * orc.hash_normal(shape, seed)).float()


@pytest.fixture(scope='module')
def ops():
    from vargp_amd import ops as o
    return o


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(32, 32, 16), (100, 512, 784), (64, 64, 64), (20, 1, 20), (1, 7, 5),
                                   (130, 70, 33), (257, 129, 100), (3, 3, 2)])
@pytest.mark.parametrize('tA,tB', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_bgemm_layouts(ops, M, N, K, tA, tB):
    A = _hn((2, 3, K, M) if tA else (2, 3, M, K), 1)
    B = _hn((3, N, K) if tB else (3, K, N), 2)       # broadcast over the leading batch dim
    Al = A.mT if tA else A
    Bl = B.mT if tB else B
    D = _hn((2, 3, M, N), 3)
    want = 0.5 * (Al.double() @ Bl.double()) + 2.0 * D.double()
    got = ops.bgemm(Al.to(DEV) if not tA else A.to(DEV).mT, Bl.to(DEV) if not tB else B.to(DEV).mT,
                    alpha=0.5, D=D.to(DEV), beta=2.0)
    scale = (Al.double().abs() @ Bl.double().abs()).max().item() + 1.0
    assert (got.cpu().double() - want).abs().max().item() < 5e-6 * scale


def test_bgemm_asymmetric_identity(ops):
    """A = I with asymmetric B catches a transposed C write (cdna guide §3)."""
    B = torch.arange(64 * 96, dtype=torch.float32).reshape(64, 96)
    got = ops.bgemm(torch.eye(64, device=DEV), B.to(DEV))
    assert torch.equal(got.cpu(), B)
    got = ops.bgemm(B.to(DEV).mT, torch.eye(64, device=DEV))
    assert torch.equal(got.cpu(), B.T)


@pytest.mark.parametrize('n', [40, 100, 200, 300])
def test_bgemm_triangular_hints(ops, n):
    L1 = _hn((4, n, n), 5).tril()
    L2 = _hn((4, n, n), 6).tril()
    X = _hn((4, n, 77), 7)
    for got, want in [
        (ops.bgemm(L1.to(DEV), X.to(DEV), triA=ops.LOWER), L1.double() @ X.double()),
        (ops.bgemm(L1.to(DEV).mT, X.to(DEV), triA=ops.UPPER), L1.mT.double() @ X.double()),
        (ops.bgemm(X.to(DEV).mT, L1.to(DEV), triB=ops.LOWER), X.mT.double() @ L1.double()),
        (ops.bgemm(X.to(DEV).mT, L1.to(DEV).mT, triB=ops.UPPER), X.mT.double() @ L1.mT.double()),
        (ops.bgemm(L1.to(DEV), L2.to(DEV), triA=ops.LOWER, triB=ops.LOWER, triC=ops.LOWER),
         L1.double() @ L2.double()),
    ]:
        assert rel_l2(got.cpu(), want) < 2e-6


def test_matmul_autograd_broadcast(ops):
    A = _hn((3, 4, 20, 20), 8).tril().to(DEV).requires_grad_(True)
    B = _hn((4, 20, 9), 9).to(DEV).requires_grad_(True)
    D = _hn((1, 4, 20, 9), 10).to(DEV).requires_grad_(True)
    out = ops.matmul(A, B, D=D, alpha=-1.0, beta=1.0, triA=ops.LOWER)
    w = _hn(out.shape, 11).to(DEV)
    (out * w).sum().backward()
    A64, B64, D64 = (t.detach().cpu().double().requires_grad_(True) for t in (A, B, D))
    ref = -(A64 @ B64) + D64
    (ref * w.cpu().double()).sum().backward()
    assert rel_l2(out.detach().cpu(), ref.detach()) < 2e-6
    assert rel_l2(A.grad.cpu(), A64.grad.tril()) < 2e-6
    assert rel_l2(B.grad.cpu(), B64.grad) < 2e-6
    assert rel_l2(D.grad.cpu(), D64.grad) < 2e-6


# ---------------------------------------------------------------------------------------------
RBF_CASES = {
    'toy_self': (3, 4, 20, None, 2, 1.5, False),
    'toy_cross': (3, 4, 20, 100, 2, 1.5, False),
    'toy_shared': (3, 4, 20, 100, 2, 1.5, True),
    'mnist_self': (3, 10, 100, None, 784, 0.0179, False),
    'mnist_shared': (3, 10, 100, 512, 784, 0.0179, True),
    'mnist_cross': (2, 3, 40, 24, 784, 0.0179, False),
    'odd_dims': (2, 3, 37, 53, 19, 0.3, False),
}


@pytest.mark.parametrize('case', sorted(RBF_CASES))
def test_rbf_gram_fwd_bwd(ops, case):
    S, C, M, N, D, scale, shared = RBF_CASES[case]
    theta = (np.log(0.5) + 0.05 * orc.hash_normal((S, D + 1), 3)).float()
    X = _hn((C, M, D), 5, scale)
    Y = None if N is None else (_hn((N, D), 7, scale) if shared else _hn((C, N, D), 7, scale))
    th_d, X_d = theta.to(DEV).requires_grad_(True), X.to(DEV).requires_grad_(True)
    Y_d = None if Y is None else Y.to(DEV).requires_grad_(True)
    K = ops.rbf_gram(th_d, X_d, Y_d, shared)
    w = _hn(K.shape, 11)
    (K * w.to(DEV)).sum().backward()

    th64, X64 = theta.double().requires_grad_(True), X.double().requires_grad_(True)
    Y64 = None if Y is None else Y.double().requires_grad_(True)
    Yref = None if Y is None else (Y64.unsqueeze(0).expand(C, -1, -1) if shared else Y64)
    K64 = orc.rbf_gram(th64, X64, Yref)
    (K64 * w.double()).sum().backward()

    # fp32 distance-GEMM noise: |d2| error ~ 1e-6 * (|a|^2 + |b|^2)  ->  relative K error ~ same
    np.testing.assert_allclose(K.detach().cpu().numpy(), K64.detach().numpy(), rtol=2e-4, atol=1e-6)
    if Y is None:   # exact gamma^2 diagonal
        g2 = torch.exp(2 * th_d.detach()[:, -1])
        assert torch.equal(K.detach().diagonal(dim1=-2, dim2=-1), g2.view(-1, 1, 1).expand(S, C, M))
    assert rel_l2(th_d.grad.cpu(), th64.grad) < 1e-3
    assert rel_l2(X_d.grad.cpu(), X64.grad) < 1e-3
    if Y is not None:
        assert rel_l2(Y_d.grad.cpu(), Y64.grad) < 1e-3


def test_rbf_gram_golden(ops):
    g = np.load(f'{GOLDEN}/ops.npz')
    for tag in ['toy', 'mnist']:
        th, x, y = (torch.from_numpy(g[f'rbf_{tag}_{k}']).to(DEV) for k in ['theta', 'x', 'y'])
        np.testing.assert_allclose(ops.rbf_gram(th, x).cpu().numpy(), g[f'rbf_{tag}_kuu'], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(ops.rbf_gram(th, x, y).cpu().numpy(), g[f'rbf_{tag}_kuf'], rtol=2e-4, atol=1e-6)


# ---------------------------------------------------------------------------------------------
def _spd(nb, n, seed):
    A = orc.hash_normal((nb, n, n + 8), seed)
    return ((A @ A.mT) / (n + 8) + 0.05 * torch.eye(n, dtype=torch.float64)).float()


@pytest.mark.parametrize('n', [1, 2, 20, 51, 63, 64, 65, 77, 98, 100, 128, 129, 200, 300])
def test_chol_inv_fwd_bwd(ops, n):
    nb = 3
    A = _spd(nb, n, 20 + n)
    A_d = A.to(DEV).requires_grad_(True)
    L, T = ops.chol_inv(A_d, 1e-4)
    wl, wt = _hn(L.shape, 31).tril(), _hn(T.shape, 33).tril()
    ((L * wl.to(DEV)).sum() + (T * wt.to(DEV)).sum()).backward()

    A64 = A.double().requires_grad_(True)
    L64 = torch.linalg.cholesky(A64 + 1e-4 * torch.eye(n, dtype=torch.float64))
    T64 = torch.linalg.solve_triangular(L64, torch.eye(n, dtype=torch.float64).expand(nb, n, n), upper=False)
    ((L64 * wl.double()).sum() + (T64 * wt.double()).sum()).backward()
    assert rel_l2(L.detach().cpu(), L64.detach()) < 1e-5
    assert rel_l2(T.detach().cpu(), T64.detach()) < 1e-4
    assert torch.equal(L.detach().triu(1), torch.zeros_like(L)) and torch.equal(T.detach().triu(1), torch.zeros_like(T))
    assert rel_l2(A_d.grad.cpu(), A64.grad) < 2e-3
    # L only (no inverse requested), no grad
    L2 = ops.chol(A.to(DEV), 1e-4)
    assert rel_l2(L2.cpu(), L64.detach()) < 1e-5


@pytest.mark.parametrize('n,want_t', [(150, True), (250, False), (257, True)])
def test_chol_blocked_logdet_through_the_c_abi(ops, n, want_t):
    """The blocked factorisation (n > 100) with the log-determinant output of the C ABI (accumulated block by block), with and
    without the inverse factor -- arguments `ops` never passes.  (Round 5: the blocked driver has no working copy of A any more;
    block column 0 and the first update of every other region read A itself.)"""
    from vargp_amd._lib import check, lib, ptr, scratch, stream_ptr
    nb = 2
    A = _spd(nb, n, 70 + n).to(DEV)
    A0 = A.clone()
    L = torch.full_like(A, float('nan'))
    T = torch.full_like(A, float('nan')) if want_t else None
    logdet = torch.full((nb,), float('nan'), device=DEV)
    info = torch.zeros(nb, dtype=torch.int32, device=DEV)
    ws = scratch(lib().vargp_chol_workspace_bytes(nb, n, 0), A.device)
    ws.fill_(float('nan'))          # nothing may depend on what the workspace held
    check(lib().vargp_chol_inv_fwd(ptr(A), 1e-4, ptr(L), ptr(T), ptr(logdet), ptr(info), nb, n, ptr(ws), ws.numel() * 4,
                                   stream_ptr()), 'vargp_chol_inv_fwd')
    torch.cuda.synchronize()
    assert torch.equal(A, A0) and int(info.abs().sum()) == 0
    L64 = torch.linalg.cholesky(A.double().cpu() + 1e-4 * torch.eye(n, dtype=torch.float64))
    assert rel_l2(L.cpu(), L64) < 1e-5 and torch.equal(L.triu(1), torch.zeros_like(L))
    # (the ABI's logdet is sum log diag L: half the log-determinant of A + eps I)
    np.testing.assert_allclose(logdet.cpu().numpy(), L64.diagonal(dim1=-2, dim2=-1).log().sum(-1).numpy(), rtol=1e-5)
    if want_t:
        T64 = torch.linalg.solve_triangular(L64, torch.eye(n, dtype=torch.float64).expand(nb, n, n), upper=False)
        assert rel_l2(T.cpu(), T64) < 1e-4 and torch.equal(T.triu(1), torch.zeros_like(T))


@pytest.mark.parametrize('D,B,n', [(784, 512, 6000), (37, 30, 100), (40, 64, 70)])
def test_gather_minibatch_follows_the_device_step_count(ops, D, B, n):
    """vargp_gather_minibatch: minibatch i = (*step_now - *step_base) of a permutation, read on the DEVICE -- float4 rows and the
    scalar form (D % 4 != 0), positions past the end of the permutation clamp to its last entry."""
    from vargp_amd._lib import check, lib, ptr, stream_ptr
    gen = torch.Generator().manual_seed(D + B)
    data = torch.randn(n, D, generator=gen).to(DEV)
    targets = torch.randint(0, 10, (n,), generator=gen).to(DEV)
    perm = torch.randperm(n, generator=gen).to(DEV)
    now, base = torch.tensor([7.0], device=DEV), torch.tensor([7.0], device=DEV)
    x = torch.full((B, D), float('nan'), device=DEV)
    y = torch.full((B,), -1, dtype=torch.int64, device=DEV)
    for i in range(min(3, (n + B - 1) // B)):
        check(lib().vargp_gather_minibatch(ptr(data), ptr(targets), ptr(perm), ptr(now), ptr(base), n, B, D, ptr(x), ptr(y),
                                           stream_ptr()), 'vargp_gather_minibatch')
        idx = perm[(torch.arange(B, device=DEV) + i * B).clamp(max=n - 1)]
        assert torch.equal(x, data[idx]) and torch.equal(y, targets[idx]), i
        now += 1.0               # (what the first kernel of the ELBO programs does through `bump`)


@pytest.mark.parametrize('n', [20, 64, 100])
def test_chol_not_positive_definite(ops, n):
    """n = 20: rows-across-threads kernel; 64 / 100: the register-resident blocked elimination (chol_small3.h) -- a pivot
    that is not positive in the middle of a four-pivot block, in the last block, and a NaN entry."""
    A = _spd(4, n, 5)
    A[1] = -A[1]
    ops.set_cholesky_error_mode('raise')
    with pytest.raises(torch.linalg.LinAlgError):
        ops.chol(A.to(DEV), 1e-4)
    ops.set_cholesky_error_mode('defer')
    try:
        L = ops.chol(A.to(DEV), 1e-4)
        assert torch.isnan(L[1]).all() and not torch.isnan(L[0]).any() and not torch.isnan(L[2]).any()
        assert ops.linalg_error_count() >= 1
        ops.reset_linalg_errors()
        # indefinite from the leading minor of order n - 2 on (inside the last block), and a NaN in the matrix
        B = _spd(4, n, 6)
        B[2, n - 3, n - 3] = -5.0
        B[3, n // 2, n // 2 - 1] = float('nan')
        B[3, n // 2 - 1, n // 2] = float('nan')
        L, T = ops.chol_inv(B.to(DEV), 1e-4)
        assert torch.isnan(L[2]).all() and torch.isnan(T[2]).all() and torch.isnan(L[3]).all()
        assert not torch.isnan(L[0]).any() and not torch.isnan(L[1]).any() and not torch.isnan(T[1]).any()
        assert ops.linalg_error_count() == 2
    finally:
        ops.set_cholesky_error_mode('raise')
        ops.reset_linalg_errors()


# ---------------------------------------------------------------------------------------------
def test_tril_pack(ops):
    g = np.load(f'{GOLDEN}/ops.npz')
    v = torch.from_numpy(g['tril_vec']).to(DEV).requires_grad_(True)
    m = ops.vec2tril(v, 5)
    np.testing.assert_allclose(m.detach().cpu().numpy(), g['tril_mat'], rtol=1e-6)
    np.testing.assert_allclose(ops.mat2trilvec(m.detach()).cpu().numpy(), g['tril_back'], rtol=1e-6)
    w = _hn(m.shape, 3)
    (m * w.to(DEV)).sum().backward()
    v64 = torch.from_numpy(g['tril_vec']).double().requires_grad_(True)
    (orc.vec2tril(v64) * w.double()).sum().backward()
    assert rel_l2(v.grad.cpu(), v64.grad) < 1e-6


def test_predictive_diag_and_kl(ops):
    S, C, M, B = 2, 3, 17, 70
    P, W = _hn((S, C, M, B), 1), _hn((S, C, M, B), 2)
    a, kd = _hn((S, C, M), 3), (1.0 + orc.hash_uniform((S, C), 4)).float()
    td = [t.to(DEV).requires_grad_(True) for t in (P, W, a, kd)]
    mu, var = ops.predictive_diag(*td)
    w1, w2 = _hn(mu.shape, 5), _hn(mu.shape, 6)
    ((mu * w1.to(DEV)).sum() + (var * w2.to(DEV)).sum()).backward()
    t64 = [t.double().requires_grad_(True) for t in (P, W, a, kd)]
    mu64 = (t64[0] * t64[2].unsqueeze(-1)).sum(-2)
    var64 = t64[3].unsqueeze(-1) - t64[0].pow(2).sum(-2) + t64[1].pow(2).sum(-2)
    ((mu64 * w1.double()).sum() + (var64 * w2.double()).sum()).backward()
    assert rel_l2(mu.detach().cpu(), mu64.detach()) < 1e-6 and rel_l2(var.detach().cpu(), var64.detach()) < 1e-6
    for d, r in zip(td, t64):
        assert rel_l2(d.grad.cpu(), r.grad) < 1e-5

    from vargp_amd import gp_utils
    g = np.load(f'{GOLDEN}/ops.npz')
    m, Lq, Lp = (torch.from_numpy(g[k]).to(DEV) for k in ['lg_m', 'kl_Lq', 'kl_Lp'])
    kl = gp_utils.mvn_kl(m.squeeze(-1), Lq, torch.zeros_like(m.squeeze(-1)), Lp)
    np.testing.assert_allclose(kl.cpu().numpy(), g['kl_val'], rtol=1e-4)


def test_softmax_likelihood(ops):
    g = np.load(f'{GOLDEN}/ops.npz')
    mu, var, y, eps = (torch.from_numpy(g[f'lik_{k}']).to(DEV) for k in ['mu', 'var', 'y', 'eps'])
    mu.requires_grad_(True)
    var.requires_grad_(True)
    nll = ops.softmax_nll(mu, var, eps, y)
    np.testing.assert_allclose(nll.item(), g['lik_nll'], rtol=1e-5)
    (3.0 * nll).backward()
    mu64, var64 = mu.detach().cpu().double().requires_grad_(True), var.detach().cpu().double().requires_grad_(True)
    (3.0 * orc.softmax_nll(mu64, var64, y.cpu(), eps.cpu().double())).backward()
    assert rel_l2(mu.grad.cpu(), mu64.grad) < 1e-5 and rel_l2(var.grad.cpu(), var64.grad) < 1e-5
    probs = ops.softmax_predict(mu.detach(), var.detach(), eps)
    np.testing.assert_allclose(probs.cpu().numpy(), g['lik_probs'], rtol=1e-5, atol=1e-7)


def test_linear_gaussian_ops_golden(ops):
    from vargp_amd import gp_utils
    g = np.load(f'{GOLDEN}/ops.npz')
    t = {k: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith('lg_')}
    cache = {}
    mu, Sig = gp_utils.linear_joint(t['lg_m'], t['lg_S'], t['lg_Kzx'], t['lg_Kzz'], t['lg_V'], t['lg_b'], cache=cache)
    for got, want in [(mu, 'lj_mu'), (Sig, 'lj_Sig'), (cache['Lz'], 'lj_Lz'), (cache['Lz_Kzx'], 'lj_LzKzx')]:
        np.testing.assert_allclose(got.cpu().numpy(), g[want], rtol=5e-4, atol=5e-5)
    kd = torch.exp(2 * t['lg_theta'][:, -1:]).unsqueeze(-2)
    mu, var = gp_utils.linear_marginal_diag(t['lg_m'], t['lg_S'], t['lg_Kzz'], t['lg_Kzx'], kd)
    np.testing.assert_allclose(mu.cpu().numpy(), g['lmd_mu'], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(var.cpu().numpy(), g['lmd_var'], rtol=5e-4, atol=5e-5)
    mu, Sig = gp_utils.gp_cond(t['lg_m'], t['lg_Kzz'], t['lg_Kzx'], t['lg_Kxx'])
    np.testing.assert_allclose(mu.cpu().numpy(), g['gc_mu'], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(Sig.cpu().numpy(), g['gc_Sig'], rtol=5e-4, atol=5e-5)


def test_trsm_lower_fwd_bwd(ops):
    """X = L^-1 B through T (vargp_trsm_lower_*) against torch.linalg.solve_triangular + autograd in fp64."""
    nb, n, nrhs = 3, 70, 45
    A = _hn((nb, n, n), 21)
    A = A @ A.mT / n + torch.eye(n)
    B = _hn((nb, n, nrhs), 22)
    gX = _hn((nb, n, nrhs), 23)
    L, T = ops.chol_inv(A.to(DEV), 0.0)
    Ld, Bd = L.detach().requires_grad_(True), B.to(DEV).requires_grad_(True)
    X = ops.trsm_lower(Ld, T.detach(), Bd)
    (X * gX.to(DEV)).sum().backward()
    L6, B6 = L.detach().cpu().double().requires_grad_(True), B.double().requires_grad_(True)
    X6 = torch.linalg.solve_triangular(L6, B6, upper=False)
    (X6 * gX.double()).sum().backward()
    assert rel_l2(X.detach().cpu(), X6.detach()) < 1e-5
    assert rel_l2(Bd.grad.cpu(), B6.grad) < 1e-5
    assert rel_l2(Ld.grad.cpu(), L6.grad.tril()) < 1e-5
