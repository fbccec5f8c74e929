"""VAR-GP model: API of the reference's `var_gp.vargp.VARGP` (var_gp/vargp.py:11-243) with the ELBO
hot path on the HIP kernels (see gp_utils.py / ops.py).

Differences that do not change results:
  * the minibatch is never expanded over classes (reference vargp.py:106): the kernel-matrix op takes
    the shared (B, D) block directly;
  * for tasks t > 0 the prior covariance of p(u_t | u_<t, theta) does not depend on the u_<t sample,
    so its Cholesky is computed once per (s, c) instead of n_v times (reference vargp.py:146-155), and
    with ep_var_mean=True (the default) the KL does not depend on the u_<t sample at all,
    so that sample is not drawn;
  * with ep_var_mean=True, `loss` of a model with previous tasks runs as ONE native program in the block-structured
    form of the linear_joint chain (csrc/elbo_tn.hip, fused.TnProgram): one kernel matrix over all inducing points, one
    factorisation, GEMMs; `compute_q` / `compute_pf_diag` / `forward(loss_cache=...)` below keep the reference's
    op-by-op composition (API surface, the ep_var_mean=False ablation, and what the program is tested against);
    gradient-free `forward` / `predict` use the same program for every model.
"""
import os

import torch
import torch.nn as nn

from . import fused, gp_utils, noise, ops
from .gp_utils import vec2tril, mat2trilvec, cholesky, rev_cholesky, gp_cond, block_joint, linear_marginal_diag
from .kernels import RBFKernel, DeepRBFKernel
from .likelihoods import MulticlassSoftmax
from .ops import LOWER


class VARGP(nn.Module):
    def __init__(self, z_init, kernel, likelihood, n_var_samples=1, ep_var_mean=True, prev_params=None):
        super().__init__()
        self.var_mean_mask = float(ep_var_mean)
        self.fused_first_task = True     # VARGP.loss of a first-task model runs as one fused node (fused.py)
        self.fused_tasks = True          # ... and of a model with previous tasks as the block-structured program
        # native block programs: training programs per shape (+ spares while one is owned by a pending backward), ONE
        # forward-only program (moments only, no gradient buffers) sized for the widest batch seen, serving narrower ones
        self._tn_ops, self._tn_progs, self._tn_spares, self._tn_eval, self._tn_eval_exact = None, {}, {}, None, {}
        self._t0_progs, self._t0_spares = {}, {}       # first-task programs (csrc/elbo_t0.hip) of the autograd route, per shape
        # loss() on a native program returns lazy terms (lazy.py: no autograd graph for the caller's linear combination, the
        # backward is one program call); VARGP_LAZY_LOSS=0 or lazy_loss = False: three autograd tensors of one node, as before
        self.lazy_loss = os.environ.get('VARGP_LAZY_LOSS', '1') != '0'
        self._seed_cache, self._gbufs = {}, None
        # frozen earlier tasks: plain dicts, not buffers (same as the reference, vargp.py:17-20);
        # u_tril is materialised lazily on first use because that needs the device the params live on
        self.prev_params = [dict(z=p['z'], u_mean=p['u_mean'], u_tril_vec=p['u_tril_vec'])
                            for p in (prev_params or [])]
        self.M = z_init.size(-2)
        self.kernel = kernel
        self.n_v = n_var_samples
        self.likelihood = likelihood

        self.z = nn.Parameter(z_init.detach().clone())
        out_size = self.z.size(0)
        self.u_mean = nn.Parameter(torch.empty(out_size, self.M, 1).normal_(0., .5))
        # packed identity (vargp.py:32-33): diagonal entries 1 (softplus(1) = 1.3133 effective)
        eye_vec = torch.zeros(self.M * (self.M + 1) // 2)
        idx = torch.arange(self.M)
        eye_vec[idx * (idx + 1) // 2 + idx] = 1.0
        self.u_tril_vec = nn.Parameter(eye_vec.unsqueeze(0).repeat(out_size, 1))

    # ------------------------------------------------------------------------------------------
    def _prev(self, i):
        """previous task i as device tensors with its u_tril (vec2tril once, cached)."""
        p = self.prev_params[i]
        dev = self.z.device
        if 'u_tril' not in p or p['u_tril'].device != dev:
            for k in ('z', 'u_mean', 'u_tril_vec'):
                p[k] = p[k].detach().to(dev)
            with torch.no_grad():
                p['u_tril'] = vec2tril(p['u_tril_vec'])
        return p

    def compute_q(self, theta, cache=None):
        """Fold previous tasks into q(u_<t | theta) and q(u_<=t | theta)  (vargp.py:35-88).
        Returns mu_lt, S_lt, mu_leq_t, S_leq_t, z_leq_t."""
        # block form (gp_utils.block_joint) instead of the reference's chain of linear_joint calls: one
        # kernel matrix over the inducing points of all tasks, one factorisation; the joint over the earlier tasks is the
        # leading block of the joint over all of them
        prev = [self._prev(i) for i in range(len(self.prev_params))]
        n_lt = sum(p['z'].size(-2) for p in prev)
        z_leq_t = torch.cat([p['z'] for p in prev] + [self.z], dim=-2)
        L, T, mu_leq_t, S_leq_t = block_joint(self.kernel.compute(theta, z_leq_t), [p['u_mean'] for p in prev] + [self.u_mean],
                                              [p['u_tril'] for p in prev] + [vec2tril(self.u_tril_vec, self.M)])
        mu_lt = mu_leq_t[..., :n_lt, :].contiguous()
        S_lt = S_leq_t[..., :n_lt, :n_lt].contiguous()
        if isinstance(cache, dict):
            # factors of K(z_<t) + eps I and Lz_<t^-1 K(z_<t, z_t), as the chain's last step left them: leading blocks of L, T
            # and -- K_{t,<} = L_{t,<} L_<<^T -- the transposed off-diagonal block row of L
            cache['Lz_lt'] = L[..., :n_lt, :n_lt].contiguous()
            cache['Tz_lt'] = T[..., :n_lt, :n_lt].contiguous()
            cache['Lz_lt_Kz_lt_z_t'] = L[..., n_lt:, :n_lt].mT.contiguous()
        return mu_lt, S_lt, mu_leq_t, S_leq_t, z_leq_t

    def compute_pf_diag(self, theta, x, mu_leq_t, S_leq_t, z_leq_t, cache=None):
        """p(f) = int p(f | u_<=t) q(u_<=t): mean and variance diagonals (S, C, B)  (vargp.py:90-113)."""
        Kzz = self.kernel.compute(theta, z_leq_t)
        Kzx = self.kernel.compute(theta, z_leq_t, x)          # x (B, D) shared by all classes
        Kxx_diag = self.kernel.compute_diag(theta)
        return linear_marginal_diag(mu_leq_t, S_leq_t, Kzz, Kzx, Kxx_diag, cache=cache)

    # -- the block-structured native program (csrc/elbo_tn.hip) ------------------------------------------------------
    T0_TILE_UNITS_MAX = 16384      # csrc/elbo_t0.hip: kT0TileUnitsMax

    def first_task_as_block(self, B=None):
        """First-task models outside the range of the LDS-resident middles of csrc/elbo_t0.hip (M <= 104 and at most 16384
        (sample, class, 64-column) tile units, i.e. S C <= 2048 at B = 512) run as the one-block case of the block program
        (csrc/elbo_tn.hip: symmetric K_uu tiles, the factorisation's pivot chains beside the K_uf row slices, paired mid-size
        products).  Measured: Permuted-MNIST task 0 (M = 200, S = 10) 557 -> 572 steps/s.  (Until round 6 the limit was 2048
        units and the 64-sample Split-MNIST step took the block program: 314 steps/s against 372 on the multi-tile forms of the
        LDS-resident kernels.)  VARGP_T0_AS_TN=0 / 1 forces it; VARGP_T0_UNITS overrides the limit on both sides."""
        if self.prev_params:
            return False
        env = os.environ.get('VARGP_T0_AS_TN')
        if env is not None:
            return env == '1'
        n_v = 1 if self.kernel.map_est else self.n_v
        # B unknown (the trainer decides its program before it has seen a batch): the reference's 512.
        ntile = (int(B if B is not None else 512) + 63) // 64
        return self.M > 104 or n_v * self.z.size(0) * ntile > int(os.environ.get('VARGP_T0_UNITS', self.T0_TILE_UNITS_MAX))

    def _tn_applicable(self):
        return (self.fused_tasks and type(self.kernel) is RBFKernel and self.z.is_cuda
                and all(p['z'].shape[-2] == self.M for p in self.prev_params))

    def _use_block_program(self, B=None):
        """Does `loss` run on the block program (csrc/elbo_tn.hip)?  Models with previous tasks: ep_var_mean=True only (the
        KL of the ablation depends on a u_<t sample); first-task models: when first_task_as_block() says so -- the mask is
        irrelevant without previous tasks -- and fused_first_task has not been cleared."""
        if not self._tn_applicable():
            return False
        if self.prev_params:
            # ep_var_mean = False (the KL keeps the conditional prior's mean at n_v samples of u_<t): the same program with its
            # tn_nm_* kernels (csrc/elbo_tn.hip), for up to 16 samples
            n_v = 1 if self.kernel.map_est else self.n_v
            return self.var_mean_mask == 1.0 or (self.var_mean_mask == 0.0 and n_v <= 16)
        return self.fused_first_task and self.first_task_as_block(B)

    def _tn_operands(self):
        """z_all (C, Mt, D), rk_all (C, nblk, M, NR): earlier tasks packed once, the last block is the program's scratch."""
        dev = self.z.device
        if self._tn_ops is None or self._tn_ops[0].device != dev:
            prev = [self._prev(i) for i in range(len(self.prev_params))]
            with torch.no_grad():
                self._tn_ops = fused.pack_tn_operands(prev, self.z.size(0), self.M, self.z.size(-1), dev)
        return self._tn_ops

    def _tn_program(self, B):
        """The (cached) training program of this shape.  While a loss() whose backward has not run yet owns its workspace
        (two losses combined before one backward), a spare of the same shape is used -- cached too, never re-allocated per
        step."""
        S = 1 if self.kernel.map_est else self.n_v
        shape = (S, self.z.size(0), self.M, self.z.size(-1), B, self.likelihood.n_f, len(self.prev_params) + 1)
        prog = self._tn_progs.get(shape)
        if prog is None:
            prog = self._tn_progs[shape] = fused.TnProgram(*shape, self.z.device, self.kernel.map_est)
        elif prog.busy:
            spares = self._tn_spares.setdefault(shape, [])
            prog = next((q for q in spares if not q.busy), None)
            if prog is None:
                prog = fused.TnProgram(*shape, self.z.device, self.kernel.map_est)
                spares.append(prog)
        return prog

    def _t0_program(self, B):
        """The (cached) first-task program of this shape for `loss` as an autograd node: descriptor + ~120 MB workspace are
        built once per shape, not per call; a spare while a loss whose backward has not run yet owns the first one."""
        S = 1 if self.kernel.map_est else self.n_v
        shape = (S, self.z.size(0), self.M, self.z.size(-1), B, self.likelihood.n_f)
        prog = self._t0_progs.get(shape)
        if prog is not None and prog.ws.device != self.z.device:
            prog = None
        if prog is None:
            prog = self._t0_progs[shape] = fused.T0Program(*shape, self.z.device, self.kernel.map_est)
        elif prog.busy:
            spares = self._t0_spares.setdefault(shape, [])
            prog = next((q for q in spares if not q.busy and q.ws.device == self.z.device), None)
            if prog is None:
                prog = fused.T0Program(*shape, self.z.device, self.kernel.map_est)
                spares.append(prog)
        return prog

    def _tn_eval_program(self, B, exact=False):
        """The forward-only program (predictive moments, no gradient buffers): one per model, carved for the widest batch
        asked for so far; narrower batches (the ragged last one of a sweep) run on it through the tile calls."""
        S = 1 if self.kernel.map_est else self.n_v
        key = (S, self.z.size(0), self.M, self.z.size(-1), self.likelihood.n_f, len(self.prev_params) + 1)
        if exact:
            # D <= 32 (the direct distance form) has no tile mode: one small program per batch size, kept -- an accuracy sweep
            # with a ragged last batch would otherwise free and re-carve the single workspace twice per data set
            prog = self._tn_eval_exact.get(key + (B,))
            if prog is None or prog.ws.device != self.z.device:
                if len(self._tn_eval_exact) >= 8:
                    self._tn_eval_exact.clear()
                prog = self._tn_eval_exact[key + (B,)] = fused.TnProgram(*key[:4], B, *key[4:], self.z.device,
                                                                         self.kernel.map_est, forward_only=True)
            return prog
        prog = self._tn_eval
        if (prog is None or prog.shape[4] < B or (prog.shape[:4] + prog.shape[5:]) != key
                or prog.ws.device != self.z.device):
            self._tn_eval = None          # release the old workspace before carving the wider one
            prog = self._tn_eval = fused.TnProgram(*key[:4], B, *key[4:], self.z.device, self.kernel.map_est, forward_only=True)
        return prog

    def release_programs(self):
        """Drop every cached native program (workspaces of several GB at Mt ~ 2000); they are re-created on demand."""
        self._tn_progs, self._tn_spares, self._tn_eval, self._tn_eval_exact = {}, {}, None, {}
        self._t0_progs, self._t0_spares = {}, {}

    def _tn_args(self):
        k = self.kernel
        return (k.log_mean.detach(), k.log_logvar.detach(), k.prior_log_mean, k.prior_log_logvar, self.z.detach().contiguous(),
                self.u_mean.detach().contiguous(), self.u_tril_vec.detach().contiguous(), *self._tn_operands())

    def forward(self, x, loss_cache=False):
        """x (B, D) -> pred_mu, pred_var (S, C, B); fills `loss_cache` with the KL ingredients if it is
        a dict  (vargp.py:115-175)."""
        if not torch.is_grad_enabled() and not isinstance(loss_cache, dict) and self._tn_applicable():
            # gradient-free evaluation (predict, accuracy sweeps): the native program, predictive moments only
            kern = self.kernel
            eps_theta = None if kern.map_est else noise.draw('eps_theta', (self.n_v, kern.log_mean.shape[0]), x.device)
            eps_theta = None if eps_theta is None else eps_theta.contiguous()
            # (D <= 32, the direct distance form, has no tile mode: exact-shape program)
            prog = self._tn_eval_program(x.size(0), exact=self.z.size(-1) <= 32)
            if prog.shape[4] == x.size(0):
                prog.forward(*self._tn_args(), x.contiguous(), None, eps_theta, None)
                mu, var = prog.moments()
            else:                                  # narrower than the program: x-independent part + one moments-only tile
                prog.sweep_begin(*self._tn_args(), eps_theta)
                mu, var = prog.sweep_moments(x.contiguous())
            return mu.clone(), var.clone()
        theta = self.kernel.sample_hypers(self.n_v)

        if self.prev_params:
            cache_q = dict()
            mu_lt, S_lt, mu_leq_t, S_leq_t, z_leq_t = self.compute_q(theta, cache=cache_q)
            pred_mu, pred_var = self.compute_pf_diag(theta, x, mu_leq_t, S_leq_t, z_leq_t)

            if isinstance(loss_cache, dict):
                Lz_Kzx = cache_q.pop('Lz_lt_Kz_lt_z_t').unsqueeze(0)          # (1, S, C, M<, M)
                Tz = cache_q.pop('Tz_lt').unsqueeze(0)
                Kzz = self.kernel.compute(theta, self.z).unsqueeze(0)
                # prior covariance Kzz - (Lz^-1 Kzx)^T (Lz^-1 Kzx): independent of u_<t
                prior_cov_t = ops.matmul(Lz_Kzx.mT, Lz_Kzx, D=Kzz, alpha=-1.0, beta=1.0)
                prior_L, prior_T = ops.chol_inv(prior_cov_t)
                if self.var_mean_mask == 1.0:
                    # var_mu - prior_mu = u_mean exactly; prior_mu itself is not needed
                    prior_mu_t = torch.zeros(1, 1, 1, 1, device=x.device)
                    var_mu_t = self.u_mean.squeeze(-1).unsqueeze(0).unsqueeze(0)
                else:
                    # u_<t ~ N(mu_<t, S_<t): Cholesky without jitter (MultivariateNormal, vargp.py:137-138)
                    Ls = ops.chol(S_lt, 0.0)
                    n_lt = mu_lt.shape[-2]
                    eps_u = noise.draw('eps_u', (self.n_v, theta.size(0), self.z.size(0), n_lt), x.device,
                                       sample_dim=1)
                    u_lt = mu_lt.unsqueeze(0) + ops.matmul(Ls.unsqueeze(0), eps_u.unsqueeze(-1), triA=LOWER)
                    Lz_u = ops.matmul(Tz, u_lt, triA=LOWER)
                    prior_mu_t = ops.matmul(Lz_Kzx.mT, Lz_u).squeeze(-1)       # (n_v, S, C, M)
                    var_mu_t = prior_mu_t * self.var_mean_mask + self.u_mean.squeeze(-1).unsqueeze(0).unsqueeze(0)
                var_L_cov_t = vec2tril(self.u_tril_vec, self.M).unsqueeze(0).unsqueeze(0)
                loss_cache.update(dict(var_mu_t=var_mu_t, var_L_cov_t=var_L_cov_t, prior_mu_t=prior_mu_t,
                                       prior_L_cov_t=prior_L, prior_T_cov_t=prior_T))
        else:
            cache_pf = dict()
            mu_leq_t = self.u_mean
            L_cov_leq_t = vec2tril(self.u_tril_vec, self.M)
            pred_mu, pred_var = self.compute_pf_diag(theta, x, mu_leq_t, rev_cholesky(L_cov_leq_t), self.z,
                                                     cache=cache_pf)
            if isinstance(loss_cache, dict):
                mu_t = mu_leq_t.squeeze(-1).unsqueeze(0).unsqueeze(0)            # q(u_1)
                L_cov_t = L_cov_leq_t.unsqueeze(0).unsqueeze(0)
                prior_mu_t = torch.zeros(1, 1, 1, 1, device=x.device)            # p(u_1) = N(0, Lz Lz^T)
                loss_cache.update(dict(var_mu_t=mu_t, var_L_cov_t=L_cov_t, prior_mu_t=prior_mu_t,
                                       prior_L_cov_t=cache_pf.pop('Lz').unsqueeze(0),
                                       prior_T_cov_t=cache_pf.pop('Tz').unsqueeze(0),
                                       # Lz^-1 (mu_q - 0) = Lz^-1 u_mean was already needed for the mean
                                       prior_d=cache_pf.pop('Lz_m').squeeze(-1).unsqueeze(0)))
        return pred_mu, pred_var

    def draw_t0_noise(self, x):
        """(eps_theta, eps_f) of one first-task step: the hyper-parameter noise of RBFKernel.sample_hypers
        (kernels.py:66-67; None under map_est) and the likelihood noise (likelihoods.py:26)."""
        kern = self.kernel
        S = 1 if kern.map_est else self.n_v
        eps_theta = None if kern.map_est else noise.draw('eps_theta', (S, kern.log_mean.shape[0]), x.device)
        eps_f = noise.draw('eps_f', (S, self.likelihood.n_f, self.z.size(0), x.size(0)), x.device)
        return eps_theta, eps_f

    # -- lazy route (lazy.py) ---------------------------------------------------------------------------------------------------
    def _lazy_ok(self):
        """The five parameters are plain trainable leaves without hooks: the program's backward may write their .grad itself."""
        if not (self.lazy_loss and torch.is_grad_enabled()):
            return False
        k = self.kernel
        ps = (k.log_mean, self.z, self.u_mean, self.u_tril_vec) + (() if k.map_est else (k.log_logvar,))
        return all(p.requires_grad and p.is_leaf and not p._backward_hooks and p.is_cuda for p in ps)

    def _seed_tensor(self, coefs):
        t = self._seed_cache.get(coefs)
        if t is None or t.device != self.z.device:
            if len(self._seed_cache) >= 64:
                self._seed_cache.clear()
            t = self._seed_cache[coefs] = torch.tensor(coefs, dtype=torch.float32, device=self.z.device)
        return t

    def _grad_buffers(self):
        """([five buffers that become .grad], [five scratch buffers for accumulation into an existing .grad]), shapes of
        (log_mean, log_logvar, z, u_mean, u_tril_vec); each set is one allocation."""
        k = self.kernel
        ps = (k.log_mean, k.log_logvar, self.z, self.u_mean, self.u_tril_vec)
        if self._gbufs is None or self._gbufs[0][2].device != self.z.device or self._gbufs[0][2].shape != self.z.shape:
            sets = []
            for _ in range(2):
                offs, tot = [], 0
                for p in ps:
                    offs.append(tot)
                    tot += (p.numel() + 63) // 64 * 64
                flat = torch.empty(tot, dtype=torch.float32, device=self.z.device)
                sets.append([flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)])
            self._gbufs = sets
        return self._gbufs

    def draw_u_noise(self, x):
        """eps_u (n_v, S, C, M<) of the u_<t ~ q(u_<t | theta) samples the ep_var_mean = False KL is averaged over (reference
        vargp.py:137-138); None for ep_var_mean = True models and first-task models."""
        if not self.prev_params or self.var_mean_mask == 1.0:
            return None
        S = 1 if self.kernel.map_est else self.n_v
        n_lt = sum(p['z'].shape[-2] for p in self.prev_params)
        return noise.draw('eps_u', (self.n_v, S, self.z.size(0), n_lt), x.device, sample_dim=1).contiguous()

    def loss(self, x, y):
        """(kl_hypers, kl_u, nll); the caller combines beta*kl_hypers + kl_u + (N/B)*nll
        (vargp.py:177-194, experiments/vargp.py:34)."""
        block = self._use_block_program(x.size(0))
        native_t0 = not self.prev_params and self.fused_first_task and type(self.kernel) is RBFKernel and not block
        if (native_t0 or block) and self._lazy_ok():
            return fused.elbo_lazy(self, x, y, block)
        if not self.prev_params and self.fused_first_task and type(self.kernel) is RBFKernel and not block:
            # first task: the native program (csrc/elbo_t0.hip) as one autograd node
            return fused.elbo_t0(self.kernel, self.z, self.u_mean, self.u_tril_vec, x, y, *self.draw_t0_noise(x),
                                 prog=self._t0_program(x.size(0)))
        if block:
            # later tasks: the block-structured program (csrc/elbo_tn.hip) as one autograd node
            eps_theta, eps_f = self.draw_t0_noise(x)
            return fused.elbo_tn(self.kernel, self.z, self.u_mean, self.u_tril_vec, x, y, eps_theta, eps_f,
                                 self._tn_program(x.size(0)), *self._tn_operands(), eps_u=self.draw_u_noise(x))
        loss_cache = dict()
        pred_mu, pred_var = self(x, loss_cache=loss_cache)
        nll = self.likelihood.loss(pred_mu, pred_var, y)
        kl = gp_utils.mvn_kl(loss_cache.pop('var_mu_t'), loss_cache.pop('var_L_cov_t'),
                             loss_cache.pop('prior_mu_t'), loss_cache.pop('prior_L_cov_t'),
                             Tp=loss_cache.pop('prior_T_cov_t'), d=loss_cache.pop('prior_d', None))
        kl_u = kl.sum(dim=-1).mean(dim=0).mean(dim=0)
        kl_hypers = self.kernel.kl_hypers()
        return kl_hypers, kl_u, nll

    def elbo_tiled(self, x, y, tile, beta=1.0, scale=1.0, noise_seed=0):
        """ELBO terms and gradient over a whole data set x (N, D), y (N) swept in minibatch tiles of `tile` points
        (BASELINE config 5: N = 1e6, M = 2048): one hyper-sample set, the kernel matrix of the inducing points and its
        factorisation computed once, K_uf built tile by tile in HBM, every tile's share of the gradient accumulated on the
        device (fused.TnProgram.tiled_step).  Returns (kl_hypers, kl_u, nll summed over the data) and writes the gradient of
        beta kl_hypers + kl_u + scale nll into the .grad of the five trainable tensors.  ep_var_mean=True models only.
        Injected noise (noise.inject: eps_theta (S, D+1), eps_f (S, F, C, N)) is honoured; otherwise the program draws its
        own (counter-based generator keyed by `noise_seed`)."""
        assert self._tn_applicable() and (not self.prev_params or self.var_mean_mask == 1.0)
        kern = self.kernel
        S = 1 if kern.map_est else self.n_v
        prog = self._tn_program(int(tile))       # honours a pending backward of VARGP.loss on the same shape
        eps_theta, eps_f = noise._injected.get('eps_theta'), noise._injected.get('eps_f')
        if eps_f is None:
            if prog._rng is None or prog._rng[0] != int(noise_seed):     # (re-)key the generator: a new seed is a new stream
                self._tiled_counter = torch.zeros(1, dtype=torch.int32, device=x.device)
                prog.set_rng(noise_seed, self._tiled_counter)
            eps_theta = None
        else:
            eps_theta = None if kern.map_est else eps_theta.to(x.device).contiguous()
            eps_f = eps_f.to(x.device)
        params = [kern.log_mean, kern.log_logvar, self.z, self.u_mean, self.u_tril_vec]
        grads = [torch.empty_like(p) for p in params]
        seeds = torch.tensor([beta, 1.0, scale], dtype=torch.float32, device=x.device)
        scal = prog.tiled_step(*self._tn_args(), x.contiguous(), y.contiguous(), seeds, grads, eps_theta, eps_f)
        for p, g in zip(params, grads):
            p.grad = g
        return scal[0].clone(), scal[1].clone(), scal[2].clone()

    def predict(self, x, tile=None):
        """Class probabilities (B, C)  (vargp.py:196-198).  With `tile`, a large x is swept in blocks of
        `tile` points that share ONE hyper-sample and ONE set of x-independent factors (K_uu, its Cholesky
        / inverse, Lz^-1 m, Lz^-1 L_S): the same result as a single call on all of x, in bounded memory."""
        if tile is None or x.size(0) <= tile:
            pred_mu, pred_var = self(x)
            return self.likelihood.predict(pred_mu, pred_var)
        if self.prev_params and not torch.is_grad_enabled() and self._tn_applicable() and self.z.size(-1) > 32:
            # the block program, forward only: K(z_<=t), its factorisation and the small products ONCE (vargp_elbo_tn_begin),
            # then K_uf, P, V2, W and the moments per tile.  (First-task models keep the per-op sweep below: with the factor of
            # S_u + eps I it needs two Mt^2 B products per tile where the block form needs three -- N = 1e6, M = 2048 sweep:
            # 0.93 s against 1.33 s.)
            kern = self.kernel
            eps_theta = None if kern.map_est else noise.draw('eps_theta', (self.n_v, kern.log_mean.shape[0]), x.device)
            prog = self._tn_eval_program(int(tile))
            prog.sweep_begin(*self._tn_args(), None if eps_theta is None else eps_theta.contiguous())
            out = []
            for i in range(0, x.size(0), tile):
                mu, var = prog.sweep_moments(x[i:i + tile].contiguous())
                out.append(self.likelihood.predict(mu, var))
            return torch.cat(out, dim=0)
        theta = self.kernel.sample_hypers(self.n_v)
        if self.prev_params:
            _, _, mu_leq_t, S_leq_t, z_leq_t = self.compute_q(theta)
        else:
            mu_leq_t, z_leq_t = self.u_mean, self.z
            S_leq_t = rev_cholesky(vec2tril(self.u_tril_vec, self.M))
        prep = gp_utils.marginal_prepare(mu_leq_t, S_leq_t, self.kernel.compute(theta, z_leq_t))
        Kxx_diag = self.kernel.compute_diag(theta)
        out = []
        for i in range(0, x.size(0), tile):
            xt = x[i:i + tile]
            mu, var, _ = gp_utils.marginal_apply(prep, self.kernel.compute(theta, z_leq_t, xt), Kxx_diag)
            out.append(self.likelihood.predict(mu, var))
        return torch.cat(out, dim=0)

    @staticmethod
    def create_clf(dataset, M=20, n_f=10, n_var_samples=3, prev_params=None,
                   ep_var_mean=True, map_est_hypers=False, dkl=False):
        """Factory used by the experiment driver (vargp.py:200-243): inducing points at random data
        points per class, hyper-prior = previous task's hyper-posterior (popped from prev_params[-1],
        which is mutated like the reference does)."""
        N = len(dataset)
        out_size = torch.unique(dataset.targets).size(0)
        z = torch.stack([dataset[torch.randperm(N)[:M]][0] for _ in range(out_size)])

        prior_log_mean, prior_log_logvar, phi_params = None, None, None
        if prev_params:
            prior_log_mean = prev_params[-1].get('kernel.log_mean')
            prior_log_logvar = prev_params[-1].get('kernel.log_logvar')
            if dkl:      # the feature map starts from the last task's (vargp.py:218-219)
                phi_params = {k[11:]: v for k, v in prev_params[-1].items() if k.startswith('kernel.phi.')}
            for p in prev_params:
                for k in [k for k in p if k.startswith('kernel')]:
                    p.pop(k)
        if dkl:
            kernel = DeepRBFKernel(z.size(-1), prior_log_mean=prior_log_mean, prior_log_logvar=prior_log_logvar,
                                   map_est=map_est_hypers)
            if phi_params:
                kernel.phi.load_state_dict(phi_params)
        else:
            kernel = RBFKernel(z.size(-1), prior_log_mean=prior_log_mean, prior_log_logvar=prior_log_logvar,
                               map_est=map_est_hypers)
        likelihood = MulticlassSoftmax(n_f=n_f)
        return VARGP(z, kernel, likelihood, n_var_samples=n_var_samples, ep_var_mean=ep_var_mean,
                     prev_params=prev_params)
