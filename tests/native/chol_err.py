"""Accuracy of the blocked factorisation L, T = L^-1 at n = 2048 against fp64 LAPACK, next to fp32 LAPACK's own error
(tuning aid: VARGP_CHOL_PANEL2=0 / unset compares the one-level and the two-level algorithm).  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vargp_oracle as orc  # noqa: E402
from helpers import rel_l2  # noqa: E402
from vargp_amd import ops  # noqa: E402

n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 64
z = orc.hash_normal((1, n, D), 5) * np.sqrt(0.25 / D)
th = torch.full((1, D + 1), np.log(0.5), dtype=torch.float64)
A = orc.rbf_gram(th, z)[0].float()
L, T = ops.chol_inv(A.cuda(), 1e-4)
A64 = A.double()
L64 = torch.linalg.cholesky(A64 + 1e-4 * torch.eye(n, dtype=torch.float64))
T64 = torch.linalg.solve_triangular(L64, torch.eye(n, dtype=torch.float64).expand(1, n, n), upper=False)
L32 = torch.linalg.cholesky(A + 1e-4 * torch.eye(n))
T32 = torch.linalg.solve_triangular(L32, torch.eye(n).expand(1, n, n), upper=False)
print('L: ours %.3e  lapack32 %.3e   T: ours %.3e  lapack32 %.3e' % (rel_l2(L.cpu(), L64), rel_l2(L32, L64), rel_l2(T.cpu(), T64), rel_l2(T32, T64)))
res = (L.cpu().double() @ L.cpu().double().mT - (A64 + 1e-4 * torch.eye(n, dtype=torch.float64))).norm() / A64.norm()
res32 = (L32.double() @ L32.double().mT - (A64 + 1e-4 * torch.eye(n, dtype=torch.float64))).norm() / A64.norm()
print('residual |L L^T - A| / |A|: ours %.3e  lapack32 %.3e' % (res.item(), res32.item()))
