"""CPU oracle for the VAR-GP ELBO hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a from-scratch CPU (torch, fp32 or fp64) *restatement* of the algorithm in the
reference's `var_gp/{kernels,gp_utils,vargp,likelihoods}.py`.  It exists only so that tests,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` have something to check the
HIP path against / to time.  Nothing under `vargp_amd/` may import it.

Parity status: PINNED.  The reference has no golden vectors of its own; this oracle is
pinned against outputs of the reference itself, produced in the build container by importing
`/root/reference` (see `tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`) and
checked by `tests/test_oracle_golden.py`.

Style: purely functional.  Parameters are plain dicts of tensors, noise is always passed in
explicitly (the reference draws it from the global torch generator):

    params = dict(z[C,M,D], u_mean[C,M,1], u_tril_vec[C,M(M+1)/2],
                  log_mean[D+1], log_logvar[D+1], prior_log_mean[D+1], prior_log_logvar[D+1])
    prev   = [dict(z, u_mean, u_tril_vec), ...]          (frozen earlier tasks)
    noise  = dict(eps_theta[S,D+1], eps_f[S,F,C,B], eps_u[n_v,S,C,Mt-M] (t>0 only))

Every function cites the reference lines it restates.
"""
import math

import torch
import torch.nn.functional as F

JITTER = 1e-4  # reference: var_gp/gp_utils.py:5 (eps default)


# ----------------------------------------------------------------------------------------------
# Deep-kernel feature map  (reference: DeepRBFKernel, var_gp/kernels.py:80-96)
# ----------------------------------------------------------------------------------------------
_feature_map = None


def deep_features(phi, x):
    """Linear(D,256) - ReLU - Linear(256,256) - ReLU - Linear(256,feat); phi = {'0.weight', '0.bias', '2.*', '4.*'}."""
    h = F.relu(F.linear(x, phi['0.weight'], phi['0.bias']))
    h = F.relu(F.linear(h, phi['2.weight'], phi['2.bias']))
    return F.linear(h, phi['4.weight'], phi['4.bias'])


class deep_kernel:
    """with deep_kernel(phi): every rbf_gram inside acts on phi(.) -- the DeepRBFKernel ablation of VARGP.create_clf."""

    def __init__(self, phi):
        self.phi = phi

    def __enter__(self):
        global _feature_map
        self.old, _feature_map = _feature_map, self.phi

    def __exit__(self, *a):
        global _feature_map
        _feature_map = self.old


# ----------------------------------------------------------------------------------------------
# RBF / ARD kernel  (reference: var_gp/kernels.py:24-60)
# ----------------------------------------------------------------------------------------------
def rbf_gram(theta, x, y=None, full_gram=False):
    """K[s,...,i,j] = g2_s * exp(-0.5 * ||x_i/sig_s - y_j/sig_s||^2).

    theta: (S, D+1) = [log lengthscales, log gamma]; x: (..., M, D); y: (..., N, D) or None (y=x).
    Distances are formed as |a|^2 + |b|^2 - 2 a.b on pre-scaled inputs with no clamp
    (var_gp/kernels.py:44-56).  For y=None the squared norms are the diagonal of the same Gram
    that supplies a.b, which makes the K diagonal exactly g2 (kernels.py:47-48,54).

    full_gram=True reproduces the reference's (wasteful) B x B Gram of y whose diagonal is the
    only part used (kernels.py:51,54); used by the cpu_baseline timing so the work matches.
    """
    if _feature_map is not None:      # deep kernel (var_gp/kernels.py:80-96): the RBF kernel acts on phi(x), phi(y)
        x = deep_features(_feature_map, x)
        y = deep_features(_feature_map, y) if y is not None else None
    S = theta.shape[0]
    lead = x.dim() - 2
    th = theta.reshape(S, *([1] * lead), 1, -1)
    sig = th[..., :-1].exp()
    g2 = (2.0 * th[..., -1:]).exp()
    a = x.unsqueeze(0) / sig
    aa = a @ a.transpose(-1, -2)
    if y is None:
        ab = aa
        na = nb = aa.diagonal(dim1=-2, dim2=-1)
    else:
        b = y.unsqueeze(0) / sig
        ab = a @ b.transpose(-1, -2)
        na = aa.diagonal(dim1=-2, dim2=-1)
        if full_gram:
            nb = (b @ b.transpose(-1, -2)).diagonal(dim1=-2, dim2=-1)
        else:
            nb = (b * b).sum(-1)
    d2 = -2.0 * ab + na.unsqueeze(-1) + nb.unsqueeze(-2)
    return g2 * (-0.5 * d2).exp()


def rbf_diag(theta):
    """gamma^2 as (S,1,1)  (var_gp/kernels.py:58-60)."""
    return (2.0 * theta[..., -1:]).exp().unsqueeze(-2)


def sample_hypers(log_mean, log_logvar, eps_theta):
    """theta = mean + std * eps  (var_gp/kernels.py:62-68; Normal.rsample)."""
    return log_mean + eps_theta * log_logvar.exp().sqrt()


def kl_hypers(log_mean, log_logvar, prior_log_mean, prior_log_logvar):
    """sum_d KL(N(m,v) || N(m0,v0))  (var_gp/kernels.py:70-77; torch _kl_normal_normal)."""
    sq = (log_logvar.exp().sqrt() / prior_log_logvar.exp().sqrt()).pow(2)
    t1 = ((log_mean - prior_log_mean) / prior_log_logvar.exp().sqrt()).pow(2)
    return (0.5 * (sq + t1 - 1.0 - sq.log())).sum()


# ----------------------------------------------------------------------------------------------
# Linear-Gaussian utilities  (reference: var_gp/gp_utils.py)
# ----------------------------------------------------------------------------------------------
def chol(mat, eps=JITTER):
    """lower Cholesky of mat + eps*I  (gp_utils.py:5-11)."""
    eye = torch.eye(mat.shape[-1], dtype=mat.dtype)
    return torch.linalg.cholesky(mat + eps * eye)


def llt(L):
    """L L^T  (gp_utils.py:14-19)."""
    return L @ L.transpose(-1, -2)


def vec2tril(vec, m=None):
    """packed row-major lower triangle -> (.., m, m), softplus on the diagonal (gp_utils.py:22-49)."""
    if m is None:
        m = int((math.sqrt(8.0 * vec.shape[-1] + 1.0) - 1.0) / 2.0)
    r, c = torch.tril_indices(m, m)
    out = vec.new_zeros(*vec.shape[:-1], m, m)
    vals = torch.where(r == c, F.softplus(vec), vec)
    out[..., r, c] = vals
    return out


def mat2trilvec(mat):
    """(.., m, m) -> packed lower triangle, row-major (gp_utils.py:52-65)."""
    r, c = torch.tril_indices(mat.shape[-1], mat.shape[-1])
    return mat[..., r, c]


def _lsolve(L, rhs):
    return torch.linalg.solve_triangular(L, rhs, upper=False)


def gp_cond(u, Kxx, Lz, Lz_Kzx):
    """p(f|u): mu = (Lz^-1 Kzx)^T Lz^-1 u, Sig = Kxx - (Lz^-1Kzx)^T(Lz^-1Kzx)  (gp_utils.py:68-98)."""
    Lz_u = _lsolve(Lz, u)
    mu = Lz_Kzx.transpose(-1, -2) @ Lz_u
    Sig = Kxx - Lz_Kzx.transpose(-1, -2) @ Lz_Kzx
    return mu, Sig


def linear_joint(m, S, Kzx, Kzz, V, b):
    """N(z;m,S) N(x;Az+b,V), A = Kxz Kzz^-1 -> joint mean/cov and the (Lz, Lz^-1 Kzx) cache
    (gp_utils.py:101-147)."""
    Lz = chol(Kzz)
    Lz_m = _lsolve(Lz, m)
    Lz_Kzx = _lsolve(Lz, Kzx)
    Am = Lz_Kzx.transpose(-1, -2) @ Lz_m
    Lz_S = _lsolve(Lz, S)
    AS = Lz_Kzx.transpose(-1, -2) @ Lz_S
    SAt = AS.transpose(-1, -2)
    Lz_SAt = _lsolve(Lz, SAt)
    ASAt = Lz_SAt.transpose(-1, -2) @ Lz_Kzx
    mu = torch.cat([m, Am + b], dim=-2)
    Sig = torch.cat([torch.cat([S, SAt], dim=-1), torch.cat([AS, V + ASAt], dim=-1)], dim=-2)
    return mu, Sig, Lz, Lz_Kzx


def linear_marginal_diag(m, S, Kzz, Kzx, Kxx_diag):
    """diag of marginal: mu = A m, var = Kxx - diag(Kxz Kzz^-1 Kzx) + diag(A (S+eps I) A^T)
    (gp_utils.py:150-191).  Returns also (Lz, Lz^-1 Kzx)."""
    Lz = chol(Kzz)
    Lz_m = _lsolve(Lz, m)
    Lz_Kzx = _lsolve(Lz, Kzx)
    mu = (Lz_Kzx.transpose(-1, -2) @ Lz_m).squeeze(-1)
    d1 = Lz_Kzx.pow(2).sum(-2)
    Lz_LS = _lsolve(Lz, chol(S))
    d2 = (Lz_LS.transpose(-1, -2) @ Lz_Kzx).pow(2).sum(-2)
    return mu, Kxx_diag - d1 + d2, Lz, Lz_Kzx


def mvn_kl(mu_q, Lq, mu_p, Lp):
    """KL(N(mu_q, LqLq^T) || N(mu_p, LpLp^T)), batched, event = last dim
    (torch.distributions.kl._kl_multivariatenormal_multivariatenormal, called from vargp.py:182-190)."""
    n = mu_q.shape[-1]
    half_logdet = Lp.diagonal(dim1=-2, dim2=-1).log().sum(-1) - Lq.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    bshape = torch.broadcast_shapes(Lq.shape[:-2], Lp.shape[:-2], mu_q.shape[:-1], mu_p.shape[:-1])
    Lp_b = Lp.expand(*bshape, n, n)
    Lq_b = Lq.expand(*bshape, n, n)
    t2 = torch.linalg.solve_triangular(Lp_b, Lq_b, upper=False).pow(2).sum((-2, -1))
    dm = (mu_q - mu_p).expand(*bshape, n).unsqueeze(-1)
    t3 = torch.linalg.solve_triangular(Lp_b, dm, upper=False).pow(2).sum((-2, -1))
    return half_logdet + 0.5 * (t2 + t3 - n)


# ----------------------------------------------------------------------------------------------
# Likelihood  (reference: var_gp/likelihoods.py:7-63)
# ----------------------------------------------------------------------------------------------
def softmax_logp(mu, var, eps_f):
    """f = mu + sqrt(var)*eps, log_softmax over classes; (S,F,C,B)  (likelihoods.py:13-31)."""
    f = mu.unsqueeze(1) + var.sqrt().unsqueeze(1) * eps_f
    return F.log_softmax(f, dim=-2)


def softmax_nll(mu, var, y, eps_f):
    """sum_b mean_{s,f} -logp[y_b]  (likelihoods.py:33-47)."""
    lp = softmax_logp(mu, var, eps_f)                       # S,F,C,B
    idx = y.view(1, 1, 1, -1).expand(lp.shape[0], lp.shape[1], 1, -1)
    picked = lp.gather(2, idx).squeeze(2)                   # S,F,B
    return (-picked).mean(1).mean(0).sum()


def softmax_predict(mu, var, eps_f):
    """probs (B,C) = exp(logsumexp_{s,f} logp) / (S F)  (likelihoods.py:49-63)."""
    lp = softmax_logp(mu, var, eps_f)
    lp = lp.reshape(-1, *mu.shape[-2:])
    return (lp.logsumexp(0).exp() / lp.shape[0]).T


# ----------------------------------------------------------------------------------------------
# VAR-GP model  (reference: var_gp/vargp.py:35-198)
# ----------------------------------------------------------------------------------------------
def compute_q(theta, params, prev):
    """Fold previous tasks into q(u_<t|theta) and q(u_<=t|theta)  (vargp.py:35-88)."""
    S = theta.shape[0]
    z_lt = prev[0]['z']
    mu_lt = prev[0]['u_mean'].unsqueeze(0).expand(S, -1, -1, -1)
    S_lt = llt(vec2tril(prev[0]['u_tril_vec'])).unsqueeze(0).expand(S, -1, -1, -1)
    for p in prev[1:]:
        Kzx = rbf_gram(theta, z_lt, p['z'])
        Kzz = rbf_gram(theta, z_lt)
        V = llt(vec2tril(p['u_tril_vec'])).unsqueeze(0).expand(S, -1, -1, -1)
        b = p['u_mean'].unsqueeze(0).expand(S, -1, -1, -1)
        mu_lt, S_lt, _, _ = linear_joint(mu_lt, S_lt, Kzx, Kzz, V, b)
        z_lt = torch.cat([z_lt, p['z']], dim=-2)
    Kzx = rbf_gram(theta, z_lt, params['z'])
    Kzz = rbf_gram(theta, z_lt)
    V = llt(vec2tril(params['u_tril_vec'])).unsqueeze(0).expand(S, -1, -1, -1)
    b = params['u_mean'].unsqueeze(0).expand(S, -1, -1, -1)
    mu_leq, S_leq, Lz_lt, Lz_lt_Kzx = linear_joint(mu_lt, S_lt, Kzx, Kzz, V, b)
    z_leq = torch.cat([z_lt, params['z']], dim=-2)
    return mu_lt, S_lt, mu_leq, S_leq, z_leq, Lz_lt, Lz_lt_Kzx


def compute_pf_diag(theta, x, mu_leq, S_leq, z_leq, full_gram=False):
    """p(f) marginal diag at the batch  (vargp.py:90-113)."""
    xf = x.unsqueeze(0).expand(z_leq.shape[0], -1, -1)
    Kzz = rbf_gram(theta, z_leq)
    Kzx = rbf_gram(theta, z_leq, xf, full_gram=full_gram)
    return linear_marginal_diag(mu_leq, S_leq, Kzz, Kzx, rbf_diag(theta))


def forward(params, prev, x, noise, want_kl=False, ep_var_mean=True, full_gram=False):
    """(pred_mu, pred_var)[S,C,B] and, if want_kl, the four KL ingredients  (vargp.py:115-175)."""
    theta = sample_hypers(params['log_mean'], params['log_logvar'], noise['eps_theta'])
    M = params['z'].shape[-2]
    kl_parts = None
    if prev:
        mu_lt, S_lt, mu_leq, S_leq, z_leq, Lz_lt, Lz_lt_Kzx = compute_q(theta, params, prev)
        pmu, pvar, _, _ = compute_pf_diag(theta, x, mu_leq, S_leq, z_leq, full_gram)
        if want_kl:
            # u_<t ~ N(mu_<t, S_<t): Cholesky WITHOUT jitter (MultivariateNormal, vargp.py:137-138)
            Ls = torch.linalg.cholesky(S_lt)
            u_lt = mu_lt.squeeze(-1).unsqueeze(0) + (Ls.unsqueeze(0) @ noise['eps_u'].unsqueeze(-1)).squeeze(-1)
            u_lt = u_lt.unsqueeze(-1)                                         # n_v,S,C,M<,1
            Kzz_t = rbf_gram(theta, params['z']).unsqueeze(0)
            prior_mu, prior_cov = gp_cond(u_lt, Kzz_t, Lz_lt.unsqueeze(0), Lz_lt_Kzx.unsqueeze(0))
            var_mu = prior_mu * float(ep_var_mean) + params['u_mean'].unsqueeze(0).unsqueeze(0)
            var_L = vec2tril(params['u_tril_vec'], M).unsqueeze(0).unsqueeze(0)
            kl_parts = (var_mu.squeeze(-1), var_L, prior_mu.squeeze(-1), chol(prior_cov))
    else:
        Lu = vec2tril(params['u_tril_vec'], M)
        pmu, pvar, Lz, _ = compute_pf_diag(theta, x, params['u_mean'], llt(Lu), params['z'], full_gram)
        if want_kl:
            mu_t = params['u_mean'].squeeze(-1).unsqueeze(0).unsqueeze(0)
            kl_parts = (mu_t, Lu.unsqueeze(0).unsqueeze(0), torch.zeros_like(mu_t), Lz.unsqueeze(0))
    return pmu, pvar, kl_parts


def loss(params, prev, x, y, noise, ep_var_mean=True, full_gram=False):
    """(kl_hypers, kl_u, nll)  (vargp.py:177-194)."""
    pmu, pvar, (mu_q, Lq, mu_p, Lp) = forward(params, prev, x, noise, True, ep_var_mean, full_gram)
    nll = softmax_nll(pmu, pvar, y, noise['eps_f'])
    kl_u = mvn_kl(mu_q, Lq, mu_p, Lp).sum(-1).mean(0).mean(0)
    kl_h = kl_hypers(params['log_mean'], params['log_logvar'],
                     params['prior_log_mean'], params['prior_log_logvar'])
    return kl_h, kl_u, nll


def predict(params, prev, x, noise):
    """probs (B,C)  (vargp.py:196-198)."""
    pmu, pvar, _ = forward(params, prev, x, noise, want_kl=False)
    return softmax_predict(pmu, pvar, noise['eps_f'])


def elbo_step(params, prev, x, y, noise, beta=1.0, n_total=None, ep_var_mean=True, full_gram=False):
    """One ELBO evaluation + gradients as the caller combines them (experiments/vargp.py:32-35):
    total = beta*kl_hypers + kl_u + (N/B)*nll.  Returns (scalars dict, grads dict)."""
    names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
    leaf = dict(params)
    for k in names:
        leaf[k] = params[k].detach().clone().requires_grad_(True)
    kl_h, kl_u, nll = loss(leaf, prev, x, y, noise, ep_var_mean, full_gram)
    scale = (n_total if n_total is not None else x.shape[0]) / x.shape[0]
    total = beta * kl_h + kl_u + scale * nll
    grads = torch.autograd.grad(total, [leaf[k] for k in names])
    return (dict(kl_hypers=kl_h.detach(), kl_u=kl_u.detach(), nll=nll.detach(), total=total.detach()),
            dict(zip(names, grads)))


# ----------------------------------------------------------------------------------------------
# Deterministic synthetic inputs (closed-form, RNG-free: regenerate bit-identically anywhere)
# ----------------------------------------------------------------------------------------------
def _hash01(idx, seed):
    """Closed-form pseudo-random in [0,1) from an int64 index tensor (SplitMix-like integer hash)."""
    x = (idx.to(torch.int64) + 0x9E3779B9 * (seed + 1)) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x.to(torch.float64) / 4294967296.0


def hash_uniform(shape, seed):
    n = 1
    for s in shape:
        n *= s
    return _hash01(torch.arange(n), seed).reshape(shape)


def hash_normal(shape, seed):
    """Box-Muller on two hashed uniforms; float64 result (cast by caller)."""
    u1 = hash_uniform(shape, 2 * seed + 101).clamp_min(1e-12)
    u2 = hash_uniform(shape, 2 * seed + 202)
    return (-2.0 * u1.log()).sqrt() * (2.0 * math.pi * u2).cos()


def make_problem(S, F_, C, M, D, B, n_prev=0, seed=0, kind='gauss', dtype=torch.float32, n_v=None, ell=0.5):
    """Synthetic problem of the named shape.  kind='gauss': x ~ N(0, 0.25/D) so that
    K_uf is O(1); kind='mnist': 19 %-dense U[0,1] pixels; kind='toy': clustered 2-D normals (ill-conditioned at t>0);
    kind='wtoy': well-separated 2-D grid points.  Inducing points are data-like rows.  `ell`: centre of the initial
    lengthscales (reference default 0.5, kernels.py:14; on the 'mnist' data that makes K_uf underflow to exactly 0 --
    ell = 2.5 puts it at O(1e-3), where the distance expansion cancels hardest).
    Returns (params, prev, x, y, noise)."""
    n_v = S if n_v is None else n_v

    def data(n, sd):
        if kind == 'mnist':
            return hash_uniform((n, D), sd) * (hash_uniform((n, D), sd + 7) < 0.19)
        if kind == 'toy':
            return 1.5 * hash_normal((n, D), sd)
        return hash_normal((n, D), sd) * math.sqrt(0.25 / D)

    x = data(B, seed + 1).to(dtype)
    y = (torch.arange(B) % C).to(torch.int64)

    zgrid = None
    if kind == 'wtoy':
        # well-separated 2-D toy (SURVEY §8c caveat 5): per class, the inducing points of ALL tasks are distinct
        # cells of a jittered grid (spacing 0.7 = 1.4 lengthscales at the initial lengthscale 0.5), so that K_uu
        # stays well-conditioned as the tasks accumulate; the minibatch is uniform over the same box
        assert D == 2
        n_all = (n_prev + 1) * M
        side = int(math.ceil(math.sqrt(n_all)))
        order = torch.argsort(hash_uniform((C, side * side), seed + 41), dim=-1)[:, :n_all]      # (C, n_all) cells
        cells = torch.stack([order // side, order % side], dim=-1).to(torch.float64)
        zgrid = 0.7 * (cells - 0.5 * (side - 1)) + 0.1 * (hash_uniform((C, n_all, 2), seed + 43) - 0.5)
        x = (0.7 * side * (hash_uniform((B, 2), seed + 1) - 0.5)).to(dtype)

    def task_params(sd, t=None):
        return dict(
            z=(zgrid[:, t * M:(t + 1) * M] if zgrid is not None else data(C * M, sd).reshape(C, M, D)).to(dtype),
            u_mean=(0.5 * hash_normal((C, M, 1), sd + 3)).to(dtype),
            u_tril_vec=(mat2trilvec(torch.eye(M, dtype=torch.float64).expand(C, M, M))
                        + 0.05 * hash_normal((C, M * (M + 1) // 2), sd + 5)).to(dtype),
        )

    prev = [task_params(seed + 1000 * (t + 1), t) for t in range(n_prev)]
    params = task_params(seed + 11, n_prev)
    log_init = math.log(0.5) + 0.05 * hash_normal((D + 1,), seed + 13)
    if ell != 0.5:
        log_init[:D] += math.log(ell / 0.5)           # the lengthscales only; log gamma stays at log 0.5
    params.update(
        log_mean=log_init.to(dtype),
        log_logvar=(-2.0 * torch.ones(D + 1, dtype=torch.float64) + 0.1 * hash_normal((D + 1,), seed + 17)).to(dtype),
        prior_log_mean=(0.02 * hash_normal((D + 1,), seed + 19)).to(dtype),
        prior_log_logvar=(0.02 * hash_normal((D + 1,), seed + 23)).to(dtype),
    )
    Mt_prev = n_prev * M
    noise = dict(
        eps_theta=hash_normal((S, D + 1), seed + 29).to(dtype),
        eps_f=hash_normal((S, F_, C, B), seed + 31).to(dtype),
    )
    if n_prev:
        noise['eps_u'] = hash_normal((n_v, S, C, Mt_prev), seed + 37).to(dtype)
    return params, prev, x, y, noise


def step_noise(S, F_, C, M, D, B, n_prev, seed, k, dtype=torch.float32):
    """Noise tensors of step k of a multi-step trajectory (closed form, as make_problem's: what the reference's training
    loop, experiments/vargp.py:29-37, draws anew on every loss() call)."""
    sd = seed + 10007 * (k + 1)
    noise = dict(eps_theta=hash_normal((S, D + 1), sd + 29).to(dtype), eps_f=hash_normal((S, F_, C, B), sd + 31).to(dtype))
    if n_prev:
        noise['eps_u'] = hash_normal((S, S, C, n_prev * M), sd + 37).to(dtype)
    return noise


def adam_trajectory(params, prev, x, y, noise_of, steps, lr, beta, n_total, names=('z', 'u_mean', 'u_tril_vec', 'log_mean',
                                                                                  'log_logvar')):
    """The reference's training loop (experiments/vargp.py:29-37: zero_grad, loss, combine, backward, step) with torch's Adam
    (the optimiser the reference keeps commented out at :22; its Yogi is not installed) on this restatement.
    -> (triples (steps, 3), final parameters)."""
    leaf = dict(params)
    for k in names:
        leaf[k] = params[k].detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([leaf[k] for k in names], lr=lr)
    triples = []
    for i in range(steps):
        opt.zero_grad()
        kl_h, kl_u, nll = loss(leaf, prev, x, y, noise_of(i))
        (beta * kl_h + kl_u + (n_total / x.shape[0]) * nll).backward()
        opt.step()
        triples.append([kl_h.item(), kl_u.item(), nll.item()])
    return torch.tensor(triples, dtype=torch.float64), {k: leaf[k].detach() for k in names}


# ----------------------------------------------------------------------------------------------
# VARGPRetrain  (reference: var_gp/vargp_retrain.py:119-233) -- the variant that re-optimises the inducing
# parameters of the earlier tasks and adds an importance-ratio term for the frozen copies
# ----------------------------------------------------------------------------------------------
def mvn_logprob(u, mu, L):
    """log N(u; mu, L L^T), event = last dim  (torch.distributions.MultivariateNormal.log_prob)."""
    n = u.shape[-1]
    bshape = torch.broadcast_shapes(u.shape[:-1], mu.shape[:-1], L.shape[:-2])
    diff = (u - mu).expand(*bshape, n).unsqueeze(-1)
    sol = torch.linalg.solve_triangular(L.expand(*bshape, n, n), diff, upper=False).squeeze(-1)
    return -0.5 * sol.pow(2).sum(-1) - L.diagonal(dim1=-2, dim2=-1).log().sum(-1) - 0.5 * n * math.log(2.0 * math.pi)


def retrain_loss(params, retrain, frozen, x, y, noise):
    """(kl_hypers, kl_u, nll) of VARGPRetrain.loss for a model with earlier tasks (vargp_retrain.py:119-233).
    retrain: the re-optimised copies of the earlier tasks' (z, u_mean, u_tril_vec) [trainable]; frozen: the same
    quantities as constants (the reference's prev_params); noise: eps_theta, eps_f, eps_u_leq (n_v,S,C,Mt),
    eps_u_tilde (n_v,n_v,S,C,Mt-M).  The two u draws are `.sample()` in the reference: no gradient flows through them."""
    theta = sample_hypers(params['log_mean'], params['log_logvar'], noise['eps_theta'])
    # q(u_<=t | theta) from the re-optimised parameters (:131-133)
    _, _, mu_leq, S_leq, z_leq, _, _ = compute_q(theta, params, retrain)
    pmu, pvar, _, _ = compute_pf_diag(theta, x, mu_leq, S_leq, z_leq)
    nll = softmax_nll(pmu, pvar, y, noise['eps_f'])
    # p(u_<=t | theta) = N(0, K(z_<=t))  (:137-138), KL(q || p) with both Cholesky factors jittered (:160-161)
    K_leq = rbf_gram(theta, z_leq)
    L_q = chol(S_leq)
    kl = mvn_kl(mu_leq.squeeze(-1), L_q, torch.zeros_like(mu_leq.squeeze(-1)), chol(K_leq)).sum(-1).mean(0)
    # q(u~_<t | theta), p(u~_<t | theta) from the frozen copies (:141-145)
    mu_lt, S_lt, _, _, _, _, _ = compute_q(theta, params, frozen)
    z_lt = torch.cat([p['z'] for p in frozen], dim=-2)
    K_lt = rbf_gram(theta, z_lt)
    # u_<=t ~ q (no gradient), u~_<t ~ p(u~_<t | u_<=t) (no gradient)  (:148-158)
    with torch.no_grad():
        u_leq = mu_leq.squeeze(-1).unsqueeze(0) + (L_q.unsqueeze(0) @ noise['eps_u_leq'].unsqueeze(-1)).squeeze(-1)
        Lz = chol(K_leq)
        Kzx = rbf_gram(theta, z_leq, z_lt)
        Lz_Kzx = _lsolve(Lz, Kzx)
        p_mu, p_S = gp_cond(u_leq.unsqueeze(-1), K_lt.unsqueeze(0), Lz.unsqueeze(0), Lz_Kzx.unsqueeze(0))
        u_tilde = p_mu.squeeze(-1).unsqueeze(0) + (chol(p_S).unsqueeze(0) @ noise['eps_u_tilde'].unsqueeze(-1)).squeeze(-1)
    # importance ratio  E[log p(u~) - log q(u~)]  (:196-219)
    lp = mvn_logprob(u_tilde, torch.zeros_like(mu_lt.squeeze(-1)), chol(K_lt))
    lq = mvn_logprob(u_tilde, mu_lt.squeeze(-1), chol(S_lt))
    ratio = (lp - lq).sum(-1).mean(-1).mean(-1).mean(-1)
    kl_h = kl_hypers(params['log_mean'], params['log_logvar'], params['prior_log_mean'], params['prior_log_logvar'])
    return kl_h, kl + ratio, nll


def make_dkl_problem(S, F_, C, M, D, B, n_prev, seed, feat=64):
    """Inputs of a deep-kernel case: make_problem's data / inducing points scaled to unit variance in the D-dim input
    space, hyper-parameters of the feat-dim RBF kernel, and deterministic weights of the feature map.
    Returns (params, prev, x, y, noise, phi)."""
    params, prev, x, y, noise = make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=seed, kind='gauss')
    scale = math.sqrt(D / 0.25)
    x = x * scale
    for p in [params] + prev:
        p['z'] = p['z'] * scale
    params.update(log_mean=(math.log(0.5) + 0.05 * hash_normal((feat + 1,), seed + 13)).float(),
                  log_logvar=(-2.0 + 0.1 * hash_normal((feat + 1,), seed + 17)).float(),
                  prior_log_mean=(0.02 * hash_normal((feat + 1,), seed + 19)).float(),
                  prior_log_logvar=(0.02 * hash_normal((feat + 1,), seed + 23)).float())
    noise['eps_theta'] = hash_normal((S, feat + 1), seed + 29).float()
    phi = {}
    for li, (o, i) in zip((0, 2, 4), [(256, D), (256, 256), (feat, 256)]):
        phi[f'{li}.weight'] = (hash_normal((o, i), seed + 61 + li) * (0.6 / math.sqrt(i))).float()
        phi[f'{li}.bias'] = (0.05 * hash_normal((o,), seed + 71 + li)).float()
    return params, prev, x, y, noise, phi
