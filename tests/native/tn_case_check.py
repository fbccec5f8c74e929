"""One shape of tn_random_sweep.py against the oracle in fp32 AND fp64 (is a deviation the conditioning of the problem or ours?).
python tests/native/tn_case_check.py S F C M n_prev D B nomean seed"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import vargp_oracle as orc  # noqa: E402
from helpers import rel_l2, to_dev  # noqa: E402
from gpu_common import build_gp, grads_of  # noqa: E402
from vargp_amd import noise  # noqa: E402

S, F_, C, M, n_prev, D, B, nomean, seed = (int(v) for v in sys.argv[1:10])
kind = 'wtoy' if D == 2 else 'gauss'
params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=seed, kind=kind)
gp = build_gp(params, prev, S, F_, ep_var_mean=not nomean)
with noise.inject(**to_dev(nz, 'cuda:0')):
    kl_h, kl_u, nll = gp.loss(x.to('cuda:0'), y.to('cuda:0'))
    (2.0 * kl_h + kl_u + 7.0 * nll).backward()
ours = dict(kl_hypers=float(kl_h), kl_u=float(kl_u), nll=float(nll))
g_ours = {k: v.cpu().double() for k, v in grads_of(gp).items()}


def dbl(t):
    if isinstance(t, torch.Tensor) and t.is_floating_point():
        return t.double()
    if isinstance(t, dict):
        return {k: dbl(v) for k, v in t.items()}
    if isinstance(t, (list, tuple)):
        return type(t)(dbl(v) for v in t)
    return t


sc32, g32 = orc.elbo_step(params, prev, x, y, nz, beta=2.0, n_total=7 * B, ep_var_mean=not nomean)
sc64, g64 = orc.elbo_step(dbl(params), dbl(prev), dbl(x), y, dbl(nz), beta=2.0, n_total=7 * B, ep_var_mean=not nomean)
for k in ours:
    ref = sc64[k].item()
    print(f'{k:10s} fp64 {ref:+.8e}   ours rel {abs(ours[k] - ref) / max(abs(ref), 1e-30):.2e}   '
          f'fp32 oracle rel {abs(sc32[k].item() - ref) / max(abs(ref), 1e-30):.2e}')
for k in g_ours:
    print(f'grad {k:12s} ours vs fp64 {rel_l2(g_ours[k], g64[k]):.2e}   fp32 oracle vs fp64 {rel_l2(g32[k].double(), g64[k]):.2e}')
