"""Monte-Carlo softmax likelihood (API of the reference's `var_gp.likelihoods.MulticlassSoftmax`,
var_gp/likelihoods.py:7-63) on the fused `vargp_softmax_*` kernels."""
import torch
import torch.nn as nn

from . import noise, ops


class MulticlassSoftmax(nn.Module):
    def __init__(self, n_f=1):
        super().__init__()
        self.n_f = n_f

    def _eps(self, mu):
        S, C, B = mu.shape
        return noise.draw('eps_f', (S, self.n_f, C, B), mu.device)

    def forward(self, mu, var):
        """log-softmax over classes of f = mu + sqrt(var) eps, (S, F, C, B)  (likelihoods.py:13-31).
        Not on the hot path (loss/predict use the fused kernels); kept for API compatibility."""
        f = mu.unsqueeze(1) + var.sqrt().unsqueeze(1) * self._eps(mu)
        return torch.log_softmax(f, dim=-2)

    def loss(self, pred_mu, pred_var, y):
        """sum_b mean_{s,f} -log p(y_b | f_sfb)  (likelihoods.py:33-47)."""
        return ops.softmax_nll(pred_mu, pred_var, self._eps(pred_mu), y)

    def predict(self, mu, var):
        """class probabilities (B, C) averaged over the S*F samples  (likelihoods.py:49-63)."""
        return ops.softmax_predict(mu, var, self._eps(mu))
