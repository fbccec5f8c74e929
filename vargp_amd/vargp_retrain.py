"""VAR-GP with re-optimised earlier tasks: API of the reference's `var_gp.vargp_retrain.VARGPRetrain`
(var_gp/vargp_retrain.py:11-267) on the HIP primitives (gp_utils.py / ops.py).

The model keeps a trainable copy of every earlier task's (z, u_mean, u_tril_vec) (`retrain_params`) next to the frozen
originals (`prev_params`) and adds to the KL an importance-ratio term between the prior and the frozen posterior at
samples u~_<t drawn through the re-optimised posterior (vargp_retrain.py:141-169, 196-219).  Every factorisation is
`vargp_chol_inv_fwd` (one launch yields L and T = L^-1), every triangular solve a GEMM with T; the two
`MultivariateNormal.sample()` draws of the reference carry no gradient (they are `sample`, not `rsample`) and are
requested from noise.py by name ('eps_u_leq', 'eps_u_tilde') so that parity tests can inject them.
"""
import math

import torch
import torch.nn as nn

from . import gp_utils, noise, ops
from .gp_utils import vec2tril, rev_cholesky, linear_marginal_diag
from .kernels import RBFKernel
from .likelihoods import MulticlassSoftmax
from .ops import LOWER

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _mvn_logprob(u, mu, L, T):
    """log N(u; mu, L L^T) over the last dim (torch.distributions.MultivariateNormal.log_prob) with T = L^-1:
    -|T (u - mu)|^2 / 2 - sum log diag L - n log(2 pi) / 2.   u (..., n); mu, L, T broadcast over the leading dims."""
    n = u.shape[-1]
    d = ops.matmul(T, (u - mu).unsqueeze(-1), triA=LOWER)              # (..., n, 1)
    sq = ops.matmul(d.mT, d).squeeze(-1).squeeze(-1)                    # |d|^2 as a 1 x 1 product on the same kernels
    return -0.5 * sq - ops.logdet_tril(L) - n * _HALF_LOG_2PI


class VARGPRetrain(nn.Module):
    def __init__(self, z_init, kernel, likelihood, n_var_samples=1, prev_params=None):
        super().__init__()
        # the trainable copies share storage with the caller's tensors, as nn.Parameter(p['z']) does in the
        # reference (vargp_retrain.py:17-25): the "frozen" originals therefore follow the optimiser's in-place updates
        self.prev_params = prev_params
        self.retrain_params = prev_params
        if prev_params:
            self.retrain_params = nn.ModuleList([
                nn.ParameterDict(dict(z=nn.Parameter(p['z']), u_mean=nn.Parameter(p['u_mean']),
                                      u_tril_vec=nn.Parameter(p['u_tril_vec'])))
                for p in prev_params])
        self.M = z_init.size(-2)
        self.kernel = kernel
        self.n_v = n_var_samples
        self.likelihood = likelihood
        self.z = nn.Parameter(z_init.detach().clone())
        out_size = self.z.size(0)
        self.u_mean = nn.Parameter(torch.empty(out_size, self.M, 1).normal_(0., .5))
        self.u_tril_vec = nn.Parameter(torch.ones(out_size, (self.M * (self.M + 1)) // 2))    # (:36: all ones)

    def _frozen(self):
        """prev_params as detached device tensors (the Parameters above alias the same storage)."""
        dev = self.z.device
        return [{k: self.retrain_params[i][k].detach() if self.retrain_params[i][k].device == dev else p[k].to(dev)
                 for k in ('z', 'u_mean', 'u_tril_vec')} for i, p in enumerate(self.prev_params)]

    def compute_q(self, theta, prev_params, cache=None):
        """q(u_<t | theta) and q(u_<=t | theta) for the given earlier-task parameters (vargp_retrain.py:38-92; the reference
        folds the tasks in one by one with gp_utils.linear_joint).  Returns mu_lt, S_lt, mu_leq_t, S_leq_t, z_lt, z_leq_t.

        Evaluated in the block form of DESIGN.md section 3 instead of the chain (gp_utils.block_joint): one kernel matrix over
        the inducing points of all tasks, one factorisation, GEMMs; the joint over the earlier tasks is the leading block of the
        joint over all of them.  Gradients flow through ops.* as before."""
        blocks = list(prev_params) + [dict(z=self.z, u_mean=self.u_mean, u_tril_vec=self.u_tril_vec)]
        n_lt = sum(p['z'].size(-2) for p in blocks[:-1])
        z_leq_t = torch.cat([p['z'] for p in blocks], dim=-2)
        z_lt = z_leq_t[..., :n_lt, :].contiguous()
        L, _, mu_leq_t, S_leq_t = gp_utils.block_joint(self.kernel.compute(theta, z_leq_t), [p['u_mean'] for p in blocks],
                                                       [vec2tril(p['u_tril_vec']) for p in blocks])
        mu_lt = mu_leq_t[..., :n_lt, :].contiguous()
        S_lt = S_leq_t[..., :n_lt, :n_lt].contiguous()
        if isinstance(cache, dict):
            # factor of K(z_<t) + eps I and Lz_<t^-1 K(z_<t, z_t), as the chain's last linear_joint call left them
            cache['Lz_lt'] = L[..., :n_lt, :n_lt]
            cache['Lz_lt_Kz_lt_z_t'] = L[..., n_lt:, :n_lt].mT
        return mu_lt, S_lt, mu_leq_t, S_leq_t, z_lt, z_leq_t

    def compute_pf_diag(self, theta, x, mu_leq_t, S_leq_t, z_leq_t, cache=None):
        """p(f) marginal mean / variance diagonals (S, C, B)  (vargp_retrain.py:94-117)."""
        Kzz = self.kernel.compute(theta, z_leq_t)
        Kzx = self.kernel.compute(theta, z_leq_t, x)
        return linear_marginal_diag(mu_leq_t, S_leq_t, Kzz, Kzx, self.kernel.compute_diag(theta), cache=cache)

    def forward(self, x, loss_cache=False):
        """x (B, D) -> pred_mu, pred_var (S, C, B)  (vargp_retrain.py:119-194)."""
        theta = self.kernel.sample_hypers(self.n_v)
        if self.prev_params:
            mu_lt, S_lt, mu_leq_t, S_leq_t, _, z_leq_t = self.compute_q(theta, self.retrain_params)
            cache_pf = dict()
            pred_mu, pred_var = self.compute_pf_diag(theta, x, mu_leq_t, S_leq_t, z_leq_t, cache=cache_pf)
            if isinstance(loss_cache, dict):
                # p(u_<=t | theta) = N(0, K(z_<=t)): its factor (jittered) came with the predictive moments
                prior_L_leq_t, prior_T_leq_t = cache_pf['Lz'], cache_pf['Tz']
                var_L_leq_t, var_T_leq_t = ops.chol_inv(S_leq_t)
                # q(u~_<t | theta), p(u~_<t | theta) from the frozen copies
                frozen = self._frozen()
                mu_lt_tilde, S_lt_tilde, *_, z_lt_tilde, _ = self.compute_q(theta, frozen)
                prior_S_lt_tilde = self.kernel.compute(theta, z_lt_tilde)
                with torch.no_grad():
                    n_leq = mu_leq_t.shape[-2]
                    C = self.z.size(0)
                    # u_<=t ~ q(u_<=t | theta)   (sample(), no gradient)
                    eps1 = noise.draw('eps_u_leq', (self.n_v, theta.size(0), C, n_leq), x.device, sample_dim=1)
                    u_leq_t = mu_leq_t.detach().squeeze(-1).unsqueeze(0) + \
                        ops.matmul(var_L_leq_t.detach().unsqueeze(0), eps1.unsqueeze(-1), triA=LOWER).squeeze(-1)
                    # u~_<t ~ p(u~_<t | u_<=t, theta) through gp_cond (its own jittered factor of K(z_<=t): the one above)
                    Kzx = self.kernel.compute(theta.detach(), z_leq_t.detach(), z_lt_tilde)
                    Tz = prior_T_leq_t.detach()
                    Lz_Kzx = ops.matmul(Tz, Kzx, triA=LOWER)
                    Lz_u = ops.matmul(Tz.unsqueeze(0), u_leq_t.unsqueeze(-1), triA=LOWER)
                    p_mu = ops.matmul(Lz_Kzx.mT.unsqueeze(0), Lz_u).squeeze(-1)                   # (n_v, S, C, M<)
                    p_S = ops.matmul(Lz_Kzx.mT, Lz_Kzx, D=prior_S_lt_tilde.detach(), alpha=-1.0, beta=1.0)
                    p_L = ops.chol(p_S)
                    n_lt = p_mu.shape[-1]
                    eps2 = noise.draw('eps_u_tilde', (self.n_v, self.n_v, theta.size(0), C, n_lt), x.device, sample_dim=2)
                    u_lt_tilde = p_mu.unsqueeze(0) + ops.matmul(p_L.unsqueeze(0).unsqueeze(0), eps2.unsqueeze(-1),
                                                               triA=LOWER).squeeze(-1)
                var_L_lt_tilde, var_T_lt_tilde = ops.chol_inv(S_lt_tilde)
                prior_L_lt_tilde, prior_T_lt_tilde = ops.chol_inv(prior_S_lt_tilde)
                loss_cache.update(dict(
                    var_mu_leq_t=mu_leq_t.squeeze(-1), var_L_leq_t=var_L_leq_t,
                    prior_mu_leq_t=torch.zeros(1, 1, 1, device=x.device), prior_L_leq_t=prior_L_leq_t,
                    prior_T_leq_t=prior_T_leq_t,
                    var_mu_lt_tilde=mu_lt_tilde.squeeze(-1), var_L_lt_tilde=var_L_lt_tilde, var_T_lt_tilde=var_T_lt_tilde,
                    prior_L_lt_tilde=prior_L_lt_tilde, prior_T_lt_tilde=prior_T_lt_tilde, u_lt_tilde=u_lt_tilde))
        else:
            cache_pf = dict()
            mu_leq_t = self.u_mean
            L_cov_leq_t = vec2tril(self.u_tril_vec, self.M)
            pred_mu, pred_var = self.compute_pf_diag(theta, x, mu_leq_t, rev_cholesky(L_cov_leq_t), self.z, cache=cache_pf)
            if isinstance(loss_cache, dict):
                loss_cache.update(dict(var_mu_t=mu_leq_t.squeeze(-1).unsqueeze(0).unsqueeze(0),
                                       var_L_cov_t=L_cov_leq_t.unsqueeze(0).unsqueeze(0),
                                       prior_mu_t=torch.zeros(1, 1, 1, 1, device=x.device),
                                       prior_L_cov_t=cache_pf.pop('Lz').unsqueeze(0),
                                       prior_T_cov_t=cache_pf.pop('Tz').unsqueeze(0),
                                       prior_d=cache_pf.pop('Lz_m').squeeze(-1).unsqueeze(0)))
        return pred_mu, pred_var

    def loss(self, x, y):
        """(kl_hypers, kl_u, nll)  (vargp_retrain.py:196-233)."""
        loss_cache = dict()
        pred_mu, pred_var = self(x, loss_cache=loss_cache)
        nll = self.likelihood.loss(pred_mu, pred_var, y)
        if self.prev_params:
            c = loss_cache
            kl = gp_utils.mvn_kl(c['var_mu_leq_t'], c['var_L_leq_t'], c['prior_mu_leq_t'], c['prior_L_leq_t'],
                                 Tp=c['prior_T_leq_t'])
            kl_u = kl.sum(dim=-1).mean(dim=0)
            u = c['u_lt_tilde']
            lp = _mvn_logprob(u, torch.zeros(1, device=x.device), c['prior_L_lt_tilde'], c['prior_T_lt_tilde'])
            lq = _mvn_logprob(u, c['var_mu_lt_tilde'], c['var_L_lt_tilde'], c['var_T_lt_tilde'])
            tilde_ratio = (lp - lq).sum(dim=-1).mean(dim=-1).mean(dim=-1).mean(dim=-1)
            kl_u = kl_u + tilde_ratio
        else:
            kl = gp_utils.mvn_kl(loss_cache.pop('var_mu_t'), loss_cache.pop('var_L_cov_t'), loss_cache.pop('prior_mu_t'),
                                 loss_cache.pop('prior_L_cov_t'), Tp=loss_cache.pop('prior_T_cov_t'),
                                 d=loss_cache.pop('prior_d'))
            kl_u = kl.sum(dim=-1).mean(dim=0).mean(dim=0)
        return self.kernel.kl_hypers(), kl_u, nll

    def predict(self, x, tile=None):
        """Class probabilities (B, C)  (vargp_retrain.py:235-237).  `tile`: a large x in chunks of `tile` points (same
        signature as VARGP.predict; every chunk draws its own hyper-parameter sample, as one call per batch would)."""
        if tile is not None and x.size(0) > tile:
            return torch.cat([self.predict(x[i:i + tile]) for i in range(0, x.size(0), tile)], dim=0)
        pred_mu, pred_var = self(x)
        return self.likelihood.predict(pred_mu, pred_var)

    @staticmethod
    def create_clf(dataset, M=20, n_f=10, n_var_samples=3, prev_params=None):
        """Factory (vargp_retrain.py:239-267): inducing points at random data points per class, hyper-prior = the last
        task's hyper-posterior (popped from prev_params, which is mutated like the reference does)."""
        N = len(dataset)
        out_size = torch.unique(dataset.targets).size(0)
        z = torch.stack([dataset[torch.randperm(N)[:M]][0] for _ in range(out_size)])
        prior_log_mean, prior_log_logvar = None, None
        if prev_params:
            prior_log_mean = prev_params[-1].get('kernel.log_mean')
            prior_log_logvar = prev_params[-1].get('kernel.log_logvar')
            for p in prev_params:
                for k in [k for k in p if k.startswith('kernel')]:
                    p.pop(k)
        kernel = RBFKernel(z.size(-1), prior_log_mean=prior_log_mean, prior_log_logvar=prior_log_logvar)
        return VARGPRetrain(z, kernel, MulticlassSoftmax(n_f=n_f), n_var_samples=n_var_samples, prev_params=prev_params)
