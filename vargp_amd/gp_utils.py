"""Linear-Gaussian building blocks of VAR-GP on the HIP kernels.

Function names and argument meaning follow the reference's `var_gp/gp_utils.py`; what differs is
underneath: `torch.cholesky` + `torch.triangular_solve` become ONE factorisation that also emits the
explicit inverse factor T = Lz^-1 (`vargp_chol_inv_fwd`), after which every "solve" is a
triangular-aware MFMA GEMM with T (`vargp_bgemm`), and the column reductions are fused
(`vargp_predictive_diag_*`).
"""
import math

import torch

from . import ops
from .ops import LOWER, UPPER, JITTER


def cholesky(M, eps=JITTER):
    """L with M + eps I = L L^T  (gp_utils.py:5-11).  Raises torch.linalg.LinAlgError on a
    non-positive-definite matrix unless ops.set_cholesky_error_mode('defer')."""
    return ops.chol(M, eps)


def cholesky_inv(M, eps=JITTER):
    """(L, T = L^-1) from one kernel."""
    return ops.chol_inv(M, eps)


def rev_cholesky(L):
    """L L^T  (gp_utils.py:14-19)."""
    return ops.matmul(L, L.mT, triA=LOWER, triB=UPPER)


def vec2tril(vec, m=None):
    """packed (.., m(m+1)/2) -> lower-triangular (.., m, m), softplus on the diagonal (gp_utils.py:22-49)."""
    if m is None:
        m = int((math.sqrt(8.0 * vec.shape[-1] + 1.0) - 1.0) / 2.0)
    return ops.vec2tril(vec, m)


def mat2trilvec(mat):
    """(.., m, m) -> packed lower triangle, row-major tril_indices order (gp_utils.py:52-65)."""
    return ops.mat2trilvec(mat)


def _inverse_factor(Lz):
    """T = Lz^-1 for a factor that did not come with one (re-factorises Lz Lz^T without jitter)."""
    return ops.chol_inv(ops.matmul(Lz, Lz.mT, triA=LOWER, triB=UPPER), 0.0)[1]


def gp_cond(u, Kzz, Kzx, Kxx, Lz=None, Lz_Kzx=None, Tz=None):
    """p(f|u): mu = (Lz^-1 Kzx)^T Lz^-1 u,  Sigma = Kxx - (Lz^-1 Kzx)^T (Lz^-1 Kzx)  (gp_utils.py:68-98)."""
    if Tz is None:
        Tz = ops.chol_inv(Kzz)[1] if Lz is None else _inverse_factor(Lz)
    Lz_u = ops.matmul(Tz, u, triA=LOWER)
    if Lz_Kzx is None:
        Lz_Kzx = ops.matmul(Tz, Kzx, triA=LOWER)
    mu = ops.matmul(Lz_Kzx.mT, Lz_u)
    Sigma = ops.matmul(Lz_Kzx.mT, Lz_Kzx, D=Kxx, alpha=-1.0, beta=1.0)
    return mu, Sigma


def linear_joint(m, S, Kzx, Kzz, V, b, cache=None):
    """N(z; m, S) N(x; A z + b, V), A = Kxz Kzz^-1  ->  joint mean (.., M+N, 1) and covariance
    (.., M+N, M+N)  (gp_utils.py:101-147)."""
    Lz, Tz = ops.chol_inv(Kzz)
    Lz_m = ops.matmul(Tz, m, triA=LOWER)
    Lz_Kzx = ops.matmul(Tz, Kzx, triA=LOWER)
    Amb = ops.matmul(Lz_Kzx.mT, Lz_m, D=b)                 # A m + b
    Lz_S = ops.matmul(Tz, S, triA=LOWER)
    AS = ops.matmul(Lz_Kzx.mT, Lz_S)
    SAt = AS.mT
    Lz_SAt = ops.matmul(Tz, SAt, triA=LOWER)
    VASAt = ops.matmul(Lz_SAt.mT, Lz_Kzx, D=V)             # V + A S A^T
    bshape = AS.shape[:-2]
    mu = torch.cat([m.expand(*bshape, -1, -1), Amb], dim=-2)
    Sigma = torch.cat([torch.cat([S.expand(*bshape, -1, -1), SAt], dim=-1),
                       torch.cat([AS, VASAt], dim=-1)], dim=-2)
    if isinstance(cache, dict):
        cache.update(dict(Lz_Kzx=Lz_Kzx, Lz=Lz, Tz=Tz))
    return mu, Sigma


def block_joint(K, means, trils):
    """The chain of linear_joint calls over tasks 0 .. t (vargp.py:35-88: q(u_<=t | theta) folded task by task) in ONE
    factorisation.  K (S, C, Mt, Mt) = kernel matrix of the inducing points of all tasks in task order; means[i] (C, M_i, 1),
    trils[i] (C, M_i, M_i) the per-task variational means / Cholesky factors.  With K' = K + eps I, L = chol(K'), T = L^-1 and
    the diagonal blocks T_ii (identities pinned in fp64 by tests/test_block_algorithm.py):
        a = [T_ii m_i]_i,   H = blockdiag(T_ii Lu_i),   mu_<=t = L a,   S_<=t = (L H)(L H)^T
    -- every K_zz factor of the chain is a leading block of L (its per-step jitter is the diagonal of K'), and the joint over
    the tasks before the last one is the leading block of mu / S.  Returns L, T, mu (S, C, Mt, 1), Sigma (S, C, Mt, Mt)."""
    L, T = ops.chol_inv(K)
    a_blk, lh_blk, o = [], [], 0
    for m, Lu in zip(means, trils):
        n = m.size(-2)
        T_ii = T[..., o:o + n, o:o + n].contiguous()
        a_blk.append(ops.matmul(T_ii, m.unsqueeze(0), triA=LOWER))                                        # T_ii m_i
        H_i = ops.matmul(T_ii, Lu.unsqueeze(0), triA=LOWER, triB=LOWER, triC=LOWER)
        lh_blk.append(ops.matmul(L[..., :, o:o + n].contiguous(), H_i, triB=LOWER))                       # columns of L H
        o += n
    a = torch.cat(a_blk, dim=-2)
    LH = torch.cat(lh_blk, dim=-1)                                                                        # lower: chol of S_<=t
    return L, T, ops.matmul(L, a, triA=LOWER), ops.matmul(LH, LH.mT)


def marginal_prepare(m, S, Kzz):
    """The x-independent part of linear_marginal_diag: Lz = chol(Kzz + eps I), Tz = Lz^-1, a = Lz^-1 m,
    G = Lz^-1 chol(S + eps I).  Reusable across minibatch tiles that share the hyper-sample."""
    Lz, Tz = ops.chol_inv(Kzz)
    a = ops.matmul(Tz, m, triA=LOWER)                      # Lz^-1 m
    LS = ops.chol(S)                                       # chol(S + eps I), gp_utils.py:182
    G = ops.matmul(Tz, LS, triA=LOWER, triB=LOWER, triC=LOWER)   # Lz^-1 L_S (lower)
    return dict(Lz=Lz, Tz=Tz, Lz_m=a, G=G)


def marginal_apply(prep, Kzx, Kxx_diag):
    """mu, var (.., B) for one block of columns Kzx (.., M, B) given marginal_prepare()'s factors."""
    Tz, a, G = prep['Tz'], prep['Lz_m'], prep['G']
    P = ops.matmul(Tz, Kzx, triA=LOWER)                    # Lz^-1 Kzx
    W = ops.matmul(G.mT, P, triA=UPPER)
    bshape = P.shape[:-2]
    per_column = Kxx_diag.shape[-1] != 1          # a full (.., B) prior diagonal instead of gamma^2
    kd = Kxx_diag.new_zeros(bshape) if per_column else \
        torch.broadcast_to(Kxx_diag, (*bshape, 1)).squeeze(-1).contiguous()
    mu, var = ops.predictive_diag(P, W, a.squeeze(-1).expand(*bshape, -1), kd)
    if per_column:
        var = var + Kxx_diag
    return mu, var, P


def linear_marginal_diag(m, S, Kzz, Kzx, Kxx_diag, cache=None):
    """Diagonal of the marginal of N(z; m, S) N(y; A z, V): mu = A m,
    var = Kxx_diag - diag(Kxz Kzz^-1 Kzx) + diag(A (S + eps I) A^T)   (gp_utils.py:150-191).
    m (.., M, 1); S (.., M, M); Kzz (.., M, M); Kzx (.., M, B); Kxx_diag broadcastable to (.., 1)."""
    prep = marginal_prepare(m, S, Kzz)
    mu, var, P = marginal_apply(prep, Kzx, Kxx_diag)
    if isinstance(cache, dict):
        cache.update(dict(Lz=prep['Lz'], Lz_Kzx=P, Tz=prep['Tz'], Lz_m=prep['Lz_m']))
    return mu, var


def mvn_kl(mu_q, Lq, mu_p, Lp, Tp=None, d=None):
    """KL(N(mu_q, Lq Lq^T) || N(mu_p, Lp Lp^T)) over the last dim, broadcasting batch dims — what
    torch's kl_divergence(MVN, MVN) computes for the reference (var_gp/vargp.py:182-190).
    Tp = Lp^-1 and d = Lp^-1 (mu_q - mu_p) may be passed in when the caller already has them."""
    if Tp is None:
        Tp = _inverse_factor(Lp)
    G = ops.matmul(Tp, Lq, triA=LOWER, triB=LOWER, triC=LOWER)
    if d is None:
        d = ops.matmul(Tp, (mu_q - mu_p).unsqueeze(-1), triA=LOWER).squeeze(-1)
    bshape = torch.broadcast_shapes(G.shape[:-2], d.shape[:-1])
    ldp = ops.logdet_tril(Lp).expand(bshape)
    ldq = ops.logdet_tril(Lq).expand(bshape)
    return ops.mvn_kl_from_factors(G.expand(*bshape, -1, -1), d.expand(*bshape, -1), ldp, ldq)
