"""fp32 pivot chain (chol_small3.h, R = float) against the fp64 one: time and accuracy of L, T = L^-1 at n <= 100 on the
matrices the first-task program factorises (K_uu of Cfg2-like inducing points, S_u = Lu Lu^T) and on the unit test's
well-conditioned random SPD matrices.  GPU box only.  Run once per arithmetic (the switch is read once per process):
    python tests/native/chol_f32_experiment.py            # fp64
    VARGP_CHOL_F32_ALONE=1 python tests/native/chol_f32_experiment.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import vargp_oracle as orc  # noqa: E402
from helpers import rel_l2  # noqa: E402
from vargp_amd import ops  # noqa: E402

mode = 'fp32' if os.environ.get('VARGP_CHOL_F32_ALONE') == '1' else 'fp64'


def report(name, A):
    n = A.shape[-1]
    L, T = ops.chol_inv(A.cuda(), 1e-4)
    eye = torch.eye(n, dtype=torch.float64)
    L64 = torch.linalg.cholesky(A.double() + 1e-4 * eye)
    T64 = torch.linalg.solve_triangular(L64, eye.expand_as(L64), upper=False)
    L32 = torch.linalg.cholesky(A + 1e-4 * torch.eye(n))
    T32 = torch.linalg.solve_triangular(L32, torch.eye(n).expand_as(L32), upper=False)
    ld = L.cpu().double().diagonal(dim1=-2, dim2=-1).log().sum(-1)
    ld64, ld32 = L64.diagonal(dim1=-2, dim2=-1).log().sum(-1), L32.double().diagonal(dim1=-2, dim2=-1).log().sum(-1)
    cond = torch.linalg.cond(A.double() + 1e-4 * eye).max().item()
    print('%-5s %-22s n=%3d cond %.1e | L err ours %.2e lapack32 %.2e | T err ours %.2e lapack32 %.2e | logdet abs err ours %.2e '
          'lapack32 %.2e' % (mode, name, n, cond, rel_l2(L.cpu(), L64), rel_l2(L32, L64), rel_l2(T.cpu(), T64), rel_l2(T32, T64),
                             (ld - ld64).abs().max().item(), (ld32 - ld64).abs().max().item()))


def spd(nb, n, seed):
    A = orc.hash_normal((nb, n, n + 8), seed)
    return ((A @ A.mT) / (n + 8) + 0.05 * torch.eye(n, dtype=torch.float64)).float()


for n in (64, 100):
    report('unit-test SPD', spd(3, n, 20 + n))
for kind, ell in (('gauss', 0.5), ('mnist', 0.5), ('mnist', 2.5), ('gauss', 2.0)):
    params, prev, x, y, nz = orc.make_problem(3, 10, 10, 100, 784, 512, seed=60, kind=kind, ell=ell)
    theta = orc.sample_hypers(params['log_mean'], params['log_logvar'], nz['eps_theta'])
    K = orc.rbf_gram(theta, params['z']).reshape(-1, 100, 100)
    report(f'K_uu {kind} ell={ell}', K)
    Lu = orc.vec2tril(params['u_tril_vec'])
    report('S_u', Lu @ Lu.mT)
# near-duplicate inducing points: eigenvalues at the jitter floor
z = orc.hash_normal((4, 100, 8), 3) * 0.3
z[:, 50:] = z[:, :50] + 1e-3 * orc.hash_normal((4, 50, 8), 4)
th = torch.full((1, 9), np.log(0.5), dtype=torch.float64)
report('K_uu near-duplicates', orc.rbf_gram(th, z)[0].float())

A = spd(40, 100, 7).cuda()
for _ in range(5):
    ops.chol_inv(A, 1e-4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ops.chol_inv(A, 1e-4)
e1.record()
torch.cuda.synchronize()
print('%s n=100 batch 40: %.1f us per call (incl. the op wrapper: info fill, allocations)' % (mode, e0.elapsed_time(e1) / 200 * 1e3))
