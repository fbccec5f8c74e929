"""The block-structured form of the t > 0 ELBO that the native program (vargp_amd/csrc/elbo_tn.hip) computes, restated
in plain torch (any dtype) WITH its hand-derived backward.  TEST INFRASTRUCTURE: tests/test_block_algorithm.py checks it
against the oracle (which follows the reference's linear_joint chain, var_gp/vargp.py:35-88, gp_utils.py:101-191) in
fp64, so that the identities below are pinned on the CPU before any kernel runs.

Identities (K' = K(z_<=t, z_<=t) + eps I, L = chol(K'), T = L^-1, blocks of size M in task order, D = blockdiag(L_ii)):
  * every Lz of the chain is a leading block of L (the leading block of a Cholesky factor is the factor of the leading
    block; the jitter the reference adds to each Kzz, gp_utils.py:5-11, is the diagonal of K');
  * A_i = K_{i,<i} (K_{<i,<i} + eps I)^-1 = L_{i,<i} T_{<i,<i},  so  I - A = D T  and  (I - A)^-1 = L D^-1:
        mu_<=t = L a,          a = [T_ii m_i]_i                          (joint mean of the chain)
        S_<=t  = L H H^T L^T,  H = blockdiag(T_ii Lu_i)                  (joint covariance; L H is its Cholesky factor)
  * predictive moments (gp_utils.py:150-191) with P = T K_uf, V2 = T^T P, W = H^T P (block-diagonal product):
        mu_b = sum_m P a,   var_b = gamma^2 - |P_b|^2 + |W_b|^2 + eps |V2_b|^2
    (the eps term is the jitter the reference adds to S_<=t before its Cholesky, gp_utils.py:182);
  * the prior covariance of p(u_t | u_<t) plus jitter (vargp.py:146-155) is L_tt L_tt^T, so with ep_var_mean
        KL = sum log diag L_tt - sum log diag Lu_t + 0.5 (|H_t|_F^2 + |a_t|^2 - M);
    without it (var_mean_mask = 0) the difference of the means is  m_t - (0) ... see `forward`.
No Mt x Mt Cholesky of S_<=t, no chain of joint covariances: one kernel matrix, one factorisation, GEMMs.
"""
import math

import torch

from oracle import vargp_oracle as orc

JITTER = orc.JITTER


def _blocks(prev, params):
    zs = [p['z'] for p in prev] + [params['z']]
    ms = [p['u_mean'] for p in prev] + [params['u_mean']]
    Lus = [orc.vec2tril(p['u_tril_vec']) for p in prev] + [orc.vec2tril(params['u_tril_vec'])]
    return torch.cat(zs, dim=-2), torch.stack(ms, dim=1).squeeze(-1), torch.stack(Lus, dim=1)   # (C,Mt,D) (C,nb,M) (C,nb,M,M)


def forward(params, prev, x, y, noise, ep_var_mean=True):
    """-> (kl_hypers, kl_u, nll), ctx"""
    theta = orc.sample_hypers(params['log_mean'], params['log_logvar'], noise['eps_theta'])
    S = theta.shape[0]
    z_all, m_blk, Lu_blk = _blocks(prev, params)
    C, Mt, D = z_all.shape
    M = params['z'].shape[-2]
    nb = Mt // M
    K = orc.rbf_gram(theta, z_all)                                             # (S,C,Mt,Mt)
    Kuf = orc.rbf_gram(theta, z_all, x.unsqueeze(0).expand(C, -1, -1))        # (S,C,Mt,B)
    L = orc.chol(K)
    eye = torch.eye(Mt, dtype=K.dtype)
    T = torch.linalg.solve_triangular(L, eye.expand_as(L), upper=False)
    Td = torch.stack([T[..., i * M:(i + 1) * M, i * M:(i + 1) * M] for i in range(nb)], dim=2)   # (S,C,nb,M,M)
    a = (Td @ m_blk.unsqueeze(0).unsqueeze(-1)).squeeze(-1)                    # (S,C,nb,M)
    H = Td @ Lu_blk.unsqueeze(0)                                               # (S,C,nb,M,M) lower
    P = T @ Kuf
    V2 = T.transpose(-1, -2) @ P
    Pb = P.reshape(S, C, nb, M, -1)
    W = (H.transpose(-1, -2) @ Pb).reshape(S, C, Mt, -1)
    g2 = (2.0 * theta[:, -1]).exp().view(S, 1, 1)
    mu = (P * a.reshape(S, C, Mt, 1)).sum(-2)
    var = g2 - P.pow(2).sum(-2) + W.pow(2).sum(-2) + JITTER * V2.pow(2).sum(-2)
    nll = orc.softmax_nll(mu, var, y, noise['eps_f'])
    Ltt = L[..., Mt - M:, Mt - M:]
    Lu_t = Lu_blk[:, -1]
    logdet = Ltt.diagonal(dim1=-2, dim2=-1).log().sum(-1) - Lu_t.diagonal(dim1=-2, dim2=-1).log().sum(-1).unsqueeze(0)
    if ep_var_mean or nb == 1:
        dvec = a[:, :, -1]                                                     # T_tt (var_mu - prior_mu) = T_tt m_t
        kl = logdet + 0.5 * (H[:, :, -1].pow(2).sum((-2, -1)) + dvec.pow(2).sum(-1) - M)   # (S,C)
        kl_u = kl.sum(-1).mean(0)
    else:
        # var_mu = u_mean, prior_mu = L_{t,<t} T_{<t,<t} u_<t with u_<t = mu_<t + chol(S_<t) eps_u   (vargp.py:137-152)
        #   T_{<t,<t} mu_<t = a_<t,  T_{<t,<t} chol(S_<t) = H_<t   =>   prior_mu = L_{t,<t} (a_<t + H_<t eps_u)
        n_lt = Mt - M
        eps_u = noise['eps_u']                                                  # (n_v,S,C,n_lt)
        Hlt = torch.block_diag(*[torch.zeros(M, M)] * 0) if False else None
        e = eps_u.reshape(eps_u.shape[0], S, C, nb - 1, M, 1)
        wv = a[:, :, :-1].unsqueeze(0) + (H[:, :, :-1].unsqueeze(0) @ e).squeeze(-1)      # (n_v,S,C,nb-1,M)
        prior_mu = (L[..., Mt - M:, :n_lt].unsqueeze(0) @ wv.reshape(-1, S, C, n_lt, 1)).squeeze(-1)   # (n_v,S,C,M)
        Ttt = T[..., Mt - M:, Mt - M:]
        dvec = (Ttt.unsqueeze(0) @ (params['u_mean'].squeeze(-1) - prior_mu).unsqueeze(-1)).squeeze(-1)
        kl = logdet.unsqueeze(0) + 0.5 * (H[:, :, -1].pow(2).sum((-2, -1)).unsqueeze(0) + dvec.pow(2).sum(-1) - M)
        kl_u = kl.sum(-1).mean(0).mean(0)
    kl_h = orc.kl_hypers(params['log_mean'], params['log_logvar'], params['prior_log_mean'], params['prior_log_logvar'])
    return kl_h, kl_u, nll


def step_with_hand_backward(params, prev, x, y, noise, seeds=(1.0, 1.0, 1.0)):
    """ep_var_mean=True.  Gradients of  s_h kl_hypers + s_u kl_u + s_n nll  wrt (z, u_mean, u_tril_vec, log_mean,
    log_logvar) with the middle of the backward (everything between the kernel matrices and the likelihood) written out
    by hand, GEMM by GEMM, exactly as vargp_elbo_tn_bwd sequences it; only the two ends (RBF kernel matrices as a function
    of theta and z; softmax likelihood as a function of mu, var) use autograd here — those kernels have their own tests."""
    s_h, s_u, s_n = seeds
    names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
    leaf = dict(params)
    for k in names:
        leaf[k] = params[k].detach().clone().requires_grad_(True)
    theta = orc.sample_hypers(leaf['log_mean'], leaf['log_logvar'], noise['eps_theta'])
    S = theta.shape[0]
    z_prev = torch.cat([p['z'] for p in prev], dim=-2)
    z_all = torch.cat([z_prev, leaf['z']], dim=-2)
    C, Mt, D = z_all.shape
    M = params['z'].shape[-2]
    nb = Mt // M
    K = orc.rbf_gram(theta, z_all)
    Kuf = orc.rbf_gram(theta, z_all, x.unsqueeze(0).expand(C, -1, -1))
    Kd, Kufd = K.detach(), Kuf.detach()
    # ---- forward (detached: the hand-written part) --------------------------------------------------------------
    _, m_blk, Lu_blk = _blocks(prev, params)
    L = orc.chol(Kd)
    T = torch.linalg.solve_triangular(L, torch.eye(Mt, dtype=K.dtype).expand_as(L), upper=False)
    blkd = lambda X: torch.stack([X[..., i * M:(i + 1) * M, i * M:(i + 1) * M] for i in range(nb)], dim=2)
    Td = blkd(T)
    a = (Td @ m_blk.unsqueeze(0).unsqueeze(-1)).squeeze(-1)
    H = Td @ Lu_blk.unsqueeze(0)
    P = T @ Kufd
    V2 = T.mT @ P
    W = (H.mT @ P.reshape(S, C, nb, M, -1)).reshape(S, C, Mt, -1)
    g2 = (2.0 * theta.detach()[:, -1]).exp()
    mu = (P * a.reshape(S, C, Mt, 1)).sum(-2).requires_grad_(True)
    var = (g2.view(S, 1, 1) - P.pow(2).sum(-2) + W.pow(2).sum(-2) + JITTER * V2.pow(2).sum(-2)).requires_grad_(True)
    nll = orc.softmax_nll(mu, var, y, noise['eps_f'])
    gmu, gvar = torch.autograd.grad(s_n * nll, [mu, var])
    # ---- backward, by hand ---------------------------------------------------------------------------------------
    g = s_u / S
    gkd = gvar.sum(-1)                                                   # (S,C): d / d gamma^2
    gW = 2.0 * W * gvar.unsqueeze(-2)
    gV2 = 2.0 * JITTER * V2 * gvar.unsqueeze(-2)
    gP = a.reshape(S, C, Mt, 1) * gmu.unsqueeze(-2) - 2.0 * P * gvar.unsqueeze(-2)
    ga = (P @ gmu.unsqueeze(-1)).squeeze(-1).reshape(S, C, nb, M).clone()
    Pb, gWb = P.reshape(S, C, nb, M, -1), gW.reshape(S, C, nb, M, -1)
    gH = Pb @ gWb.mT                                                     # (S,C,nb,M,M)
    gP = gP + (H @ gWb).reshape(S, C, Mt, -1)
    gP = gP + T @ gV2
    ga[:, :, -1] += g * a[:, :, -1]                                      # KL
    gH[:, :, -1] += g * H[:, :, -1]
    gT = (gP @ Kufd.mT + P @ gV2.mT).tril()
    gTd = ga.unsqueeze(-1) @ m_blk.unsqueeze(0).unsqueeze(-2) + gH @ Lu_blk.unsqueeze(0).mT      # (S,C,nb,M,M)
    for i in range(nb):
        gT[..., i * M:(i + 1) * M, i * M:(i + 1) * M] += gTd[:, :, i].tril()
    gKuf = T.mT @ gP
    # Cholesky backward with a diagonal gL (g / L_jj on the last block): P_low = tril(L^T gL - gT T^T), and the
    # lower triangle of L^T diag(.) is its diagonal, g on the last block:  Smat = 0.5 sym(tril(-gT T^T)) + 0.5 g I_t
    w1 = (gT @ T.mT).tril()
    Smat = -0.5 * (w1 + w1.mT - torch.diag_embed(w1.diagonal(dim1=-2, dim2=-1)))
    idx = torch.arange(Mt - M, Mt)
    Smat[..., idx, idx] += 0.5 * g
    gK = T.mT @ Smat @ T
    g_m = (Td[:, :, -1].mT @ ga[:, :, -1].unsqueeze(-1)).sum(0)          # (C,M,1)
    gLu = (Td[:, :, -1].mT @ gH[:, :, -1]).sum(0).tril()                 # (C,M,M)
    Lu_t = Lu_blk[:, -1]
    gLu = gLu - s_u * torch.diag_embed(1.0 / Lu_t.diagonal(dim1=-2, dim2=-1))
    # ---- the two ends ------------------------------------------------------------------------------------------------
    Lu_leaf = orc.vec2tril(leaf['u_tril_vec'])
    kl_h = orc.kl_hypers(leaf['log_mean'], leaf['log_logvar'], params['prior_log_mean'], params['prior_log_logvar'])
    g2_leaf = (2.0 * theta[:, -1]).exp()
    surrogate = (K * gK).sum() + (Kuf * gKuf).sum() + (g2_leaf * gkd.sum(-1)).sum() + (Lu_leaf * gLu).sum() \
        + (leaf['u_mean'] * g_m).sum() + s_h * kl_h
    grads = torch.autograd.grad(surrogate, [leaf[k] for k in names])
    return dict(zip(names, grads))
