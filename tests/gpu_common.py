"""Helpers for the -m gpu tests."""
import torch

from vargp_amd.kernels import RBFKernel
from vargp_amd.likelihoods import MulticlassSoftmax
from vargp_amd.vargp import VARGP

DEV = 'cuda:0'


def build_gp(params, prev, S, F_, ep_var_mean=True):
    D = params['z'].shape[-1]
    kern = RBFKernel(D, prior_log_mean=params['prior_log_mean'], prior_log_logvar=params['prior_log_logvar'])
    gp = VARGP(params['z'], kern, MulticlassSoftmax(n_f=F_), n_var_samples=S, ep_var_mean=ep_var_mean,
               prev_params=[{k: v.clone() for k, v in p.items()} for p in prev])
    with torch.no_grad():
        gp.kernel.log_mean.copy_(params['log_mean'])
        gp.kernel.log_logvar.copy_(params['log_logvar'])
        gp.u_mean.copy_(params['u_mean'])
        gp.u_tril_vec.copy_(params['u_tril_vec'])
    return gp.to(DEV)


def grads_of(gp):
    return dict(z=gp.z.grad, u_mean=gp.u_mean.grad, u_tril_vec=gp.u_tril_vec.grad,
                log_mean=gp.kernel.log_mean.grad, log_logvar=gp.kernel.log_logvar.grad)
