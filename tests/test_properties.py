"""Property tests (hypothesis): K_uu symmetric with an exact gamma^2 diagonal; Cholesky reconstructs and
T L = I; mat2trilvec(vec2tril(v)) equals v off the diagonal and softplus(v) on it; predictive probabilities sum to 1;
linearity of the likelihood's Monte-Carlo mean.  The CPU half runs on the oracle, the `gpu` half on the HIP ops with
randomly drawn shapes (including the awkward ones: M or N not multiples of 4, a single row, D on either side of the
direct / MFMA switch)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import vargp_oracle as orc

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


def _data(shape, seed, scale=1.0):
    return (scale * orc.hash_normal(shape, seed)).float()


# ---------------------------------------------------------------------------------------------------------------
# CPU: the oracle
# ---------------------------------------------------------------------------------------------------------------
@settings(max_examples=25, **COMMON)
@given(S=st.integers(1, 3), C=st.integers(1, 3), M=st.integers(1, 24), D=st.integers(1, 40), seed=st.integers(0, 10 ** 6))
def test_oracle_kuu_symmetric_exact_diagonal(S, C, M, D, seed):
    theta = np.log(0.5) + 0.1 * _data((S, D + 1), seed)
    x = _data((C, M, D), seed + 1, 0.7)
    K = orc.rbf_gram(theta, x)
    g2 = (2 * theta[:, -1]).exp()
    assert torch.equal(K.diagonal(dim1=-2, dim2=-1), g2.view(S, 1, 1).expand(S, C, M))
    assert torch.allclose(K, K.mT, rtol=0, atol=1e-6)
    assert (K <= g2.view(S, 1, 1, 1) * (1 + 1e-5)).all() and (K >= 0).all()


@settings(max_examples=25, **COMMON)
@given(m=st.integers(1, 30), nb=st.integers(1, 4), seed=st.integers(0, 10 ** 6))
def test_oracle_tril_pack_roundtrip(m, nb, seed):
    v = _data((nb, m * (m + 1) // 2), seed)
    L = orc.vec2tril(v, m)
    back = orc.mat2trilvec(L)
    r, c = torch.tril_indices(m, m)
    want = torch.where(r == c, torch.nn.functional.softplus(v), v)
    assert torch.equal(back, want)
    assert torch.equal(L.triu(1), torch.zeros_like(L))
    assert (L.diagonal(dim1=-2, dim2=-1) > 0).all()


@settings(max_examples=20, **COMMON)
@given(S=st.integers(1, 3), F_=st.integers(1, 4), C=st.integers(2, 6), B=st.integers(1, 20), seed=st.integers(0, 10 ** 6))
def test_oracle_probs_sum_to_one(S, F_, C, B, seed):
    mu, var = _data((S, C, B), seed), 0.1 + orc.hash_uniform((S, C, B), seed + 1).float()
    probs = orc.softmax_predict(mu, var, _data((S, F_, C, B), seed + 2))
    np.testing.assert_allclose(probs.sum(-1).numpy(), 1.0, atol=1e-5)
    assert (probs >= 0).all()


# ---------------------------------------------------------------------------------------------------------------
# GPU: the HIP ops
# ---------------------------------------------------------------------------------------------------------------
DEV = 'cuda:0'


@pytest.mark.gpu
@settings(max_examples=20, **COMMON)
@given(S=st.integers(1, 3), C=st.integers(1, 3), M=st.integers(1, 150), D=st.sampled_from([1, 2, 7, 32, 33, 40, 100]),
       seed=st.integers(0, 10 ** 6))
def test_hip_kuu_symmetric_exact_diagonal(S, C, M, D, seed):
    from vargp_amd import ops
    theta = (np.log(0.5) + 0.1 * _data((S, D + 1), seed)).to(DEV)
    x = _data((C, M, D), seed + 1, 0.5 / np.sqrt(D)).to(DEV)
    K = ops.rbf_gram(theta, x)
    g2 = (2 * theta[:, -1]).exp()
    assert torch.equal(K.diagonal(dim1=-2, dim2=-1), g2.view(S, 1, 1).expand(S, C, M))
    assert torch.allclose(K, K.mT, rtol=0, atol=2e-6)
    want = orc.rbf_gram(theta.cpu().double(), x.cpu().double())
    assert torch.allclose(K.cpu().double(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@settings(max_examples=15, **COMMON)
@given(n=st.integers(1, 260), nb=st.integers(1, 3), seed=st.integers(0, 10 ** 6))
def test_hip_cholesky_reconstructs(n, nb, seed):
    from vargp_amd import ops
    A = _data((nb, n, n), seed)
    A = (A @ A.mT / n + torch.eye(n)).to(DEV)
    L, T = ops.chol_inv(A, 1e-4)
    eye = torch.eye(n, device=DEV)
    assert torch.equal(L.triu(1), torch.zeros_like(L)) and torch.equal(T.triu(1), torch.zeros_like(T))
    rec = (L.double() @ L.double().mT - (A.double() + 1e-4 * eye.double())).abs().max().item()
    assert rec < 2e-5 * A.abs().max().item() * max(1.0, n / 50)
    assert (T.double() @ L.double() - eye.double()).abs().max().item() < 1e-4


@pytest.mark.gpu
@settings(max_examples=20, **COMMON)
@given(m=st.integers(1, 120), nb=st.integers(1, 4), seed=st.integers(0, 10 ** 6))
def test_hip_tril_pack_roundtrip(m, nb, seed):
    from vargp_amd import ops
    v = _data((nb, m * (m + 1) // 2), seed).to(DEV)
    L = ops.vec2tril(v, m)
    back = ops.mat2trilvec(L)
    r, c = torch.tril_indices(m, m)
    want = torch.where((r == c).to(DEV), torch.nn.functional.softplus(v), v)
    assert torch.allclose(back, want, rtol=1e-6, atol=1e-7)
    assert torch.equal(L.triu(1), torch.zeros_like(L))


@pytest.mark.gpu
@settings(max_examples=20, **COMMON)
@given(S=st.integers(1, 3), F_=st.integers(1, 4), C=st.integers(2, 20), B=st.integers(1, 70), seed=st.integers(0, 10 ** 6))
def test_hip_probs_sum_to_one_and_nll_matches(S, F_, C, B, seed):
    from vargp_amd import ops
    mu, var = _data((S, C, B), seed), 0.1 + orc.hash_uniform((S, C, B), seed + 1).float()
    eps = _data((S, F_, C, B), seed + 2)
    y = (torch.arange(B) * 7 % C).to(torch.int64)
    probs = ops.softmax_predict(mu.to(DEV), var.to(DEV), eps.to(DEV))
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(probs.cpu().numpy(), orc.softmax_predict(mu, var, eps).numpy(), atol=1e-5)
    nll = ops.softmax_nll(mu.to(DEV), var.to(DEV), eps.to(DEV), y.to(DEV))
    np.testing.assert_allclose(nll.item(), orc.softmax_nll(mu, var, y, eps).item(), rtol=2e-5)
