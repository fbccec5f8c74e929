"""RBF / ARD kernel with a variational Normal over log-hyperparameters.

API of the reference's `var_gp.kernels.RBFKernel` (var_gp/kernels.py:7-77); the kernel matrices are
built by `vargp_rbf_gram_{fwd,bwd}` (f32-MFMA distance GEMM with fused exp epilogue).
"""
import math

import torch
import torch.nn as nn

from . import noise, ops


class RBFKernel(nn.Module):
    def __init__(self, in_size, prior_log_mean=None, prior_log_logvar=None, map_est=False):
        super().__init__()
        self.map_est = map_est
        # variational parameters over theta = [log lengthscale_1..D, log gamma]  (kernels.py:14-17)
        self.log_mean = nn.Parameter(math.log(0.5) + 0.05 * torch.randn(in_size + 1))
        self.log_logvar = nn.Parameter(torch.full((in_size + 1,), -2.0))
        # hyper-prior (kernels.py:19-22)
        self.register_buffer('prior_log_mean', torch.zeros(in_size + 1) if prior_log_mean is None
                             else prior_log_mean.detach().clone())
        self.register_buffer('prior_log_logvar', torch.zeros(in_size + 1) if prior_log_logvar is None
                             else prior_log_logvar.detach().clone())

    def compute(self, kern_samples, x, y=None):
        """kern_samples (S, D+1); x (...batch, M, D); y (...batch, N, D) or None (= x).
        Returns (S, ...batch, M, N)  (kernels.py:24-56).  A `y` that is an expand() of one (N, D)
        block over the batch dims (how the reference feeds the minibatch, vargp.py:106) is consumed
        without materialising the copies."""
        batch = x.shape[:-2]
        M, D = x.shape[-2:]
        X = x.reshape(-1, M, D)
        Y, shared = None, False
        if y is not None:
            N = y.shape[-2]
            if y.dim() == 2 or all(st == 0 or sz == 1 for st, sz in zip(y.stride()[:-2], y.shape[:-2])):
                Y, shared = y[(0,) * (y.dim() - 2)], True
            else:
                Y = y.expand(*batch, N, D).reshape(-1, N, D)
        K = ops.rbf_gram(kern_samples, X, Y, shared)
        return K.reshape(kern_samples.shape[0], *batch, M, K.shape[-1])

    def compute_diag(self, kern_samples):
        """gamma^2 as (S, 1, 1)  (kernels.py:58-60)."""
        return (2.0 * kern_samples[..., -1:]).exp().unsqueeze(-2)

    def sample_hypers(self, n_hypers):
        """reparameterised theta ~ N(log_mean, exp(log_logvar))  (kernels.py:62-68)."""
        if self.map_est:
            return self.log_mean.unsqueeze(0)
        eps = noise.draw('eps_theta', (n_hypers, self.log_mean.shape[0]), self.log_mean.device)
        return ops.hyper_sample(self.log_mean, self.log_logvar, eps)

    def kl_hypers(self):
        """sum_d KL(q(theta_d) || p(theta_d)), both diagonal Normals  (kernels.py:70-77)."""
        if self.map_est:
            return torch.tensor(0.0, device=self.log_mean.device)
        return ops.hyper_kl(self.log_mean, self.log_logvar, self.prior_log_mean, self.prior_log_logvar)


class DeepRBFKernel(RBFKernel):
    """RBF kernel on learned features phi(x) (reference: var_gp/kernels.py:80-96, the `dkl` ablation of
    VARGP.create_clf).  `phi` is the reference's nn.Sequential (same parameter names, so state dicts and the
    `kernel.phi.*` carry-over of create_clf are interchangeable); it is evaluated by ops.linear_act, i.e. the MFMA GEMM
    with a fused bias / ReLU pass, forward and backward."""

    def __init__(self, in_size, feature_size=64, **kwargs):
        super().__init__(feature_size, **kwargs)
        self.phi = nn.Sequential(
            nn.Linear(in_size, 256),
            nn.ReLU(),
            nn.Linear(256, 256),
            nn.ReLU(),
            nn.Linear(256, feature_size),
        )

    def features(self, x):
        h = ops.linear_act(x, self.phi[0].weight, self.phi[0].bias, True)
        h = ops.linear_act(h, self.phi[2].weight, self.phi[2].bias, True)
        return ops.linear_act(h, self.phi[4].weight, self.phi[4].bias, False)

    def compute(self, kern_samples, x, y=None):
        x = self.features(x)
        if y is not None:
            if y.dim() > 2 and all(st == 0 or sz == 1 for st, sz in zip(y.stride()[:-2], y.shape[:-2])):
                y = y[(0,) * (y.dim() - 2)]           # an expand() of one (N, D) block: map it once
            y = self.features(y)
        return super().compute(kern_samples, x, y=y)
