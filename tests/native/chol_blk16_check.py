"""The fp32 pivot chain of 64 < n <= 100 (blocked on the matrix core: csrc/chol_blk16.h) through the stand-alone op
(VARGP_CHOL_F32_ALONE=1 selects the fp32 arithmetic there) against fp64 LAPACK, with LAPACK-fp32 beside it; and its time.
GPU box:  VARGP_CHOL_F32_ALONE=1 python tests/native/chol_blk16_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402

torch.manual_seed(0)
ops.set_cholesky_error_mode('defer')
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
for n in (65, 68, 80, 96, 99, 100):
    x = torch.randn(30, n, 40, dtype=torch.float64)
    K = (-(0.5 * torch.cdist(x, x) ** 2 / 40)).exp()                 # RBF kernel matrix of Gaussian points
    A = K.float().cuda()
    L, T = ops.chol_inv(A)
    K64 = A.double().cpu() + 1e-4 * torch.eye(n, dtype=torch.float64)
    L64 = torch.linalg.cholesky(K64)
    T64 = torch.linalg.solve_triangular(L64, torch.eye(n, dtype=torch.float64).expand(30, n, n), upper=False)
    L32 = torch.linalg.cholesky(K64.float())
    T32 = torch.linalg.solve_triangular(L32, torch.eye(n).expand(30, n, n), upper=False)
    print('n = %3d  L %.2e (LAPACK32 %.2e)  T %.2e (LAPACK32 %.2e)  |T L - I| %.2e  upper zero: %s' % (
        n, rel(L.cpu(), L64), rel(L32, L64), rel(T.cpu(), T64), rel(T32, T64),
        (T.cpu().double() @ L.cpu().double() - torch.eye(n, dtype=torch.float64)).abs().max().item(),
        bool((L.triu(1) == 0).all() and (T.triu(1) == 0).all())))
# a matrix that is not positive definite: flagged, NaN out
B = torch.eye(100, device='cuda').repeat(2, 1, 1)
B[1, 50, 50] = -1.0
ops.reset_linalg_errors()
L, T = ops.chol_inv(B)
torch.cuda.synchronize()
print('non-PD: errors counted', ops.linalg_error_count(), ' NaN out:', bool(torch.isnan(L[1]).all()), ' good one intact:', bool((L[0] - torch.eye(100, device='cuda') * (1 + 1e-4) ** 0.5).abs().max() < 1e-6))
A = torch.randn(30, 100, 100, device='cuda')
A = A @ A.mT / 100 + torch.eye(100, device='cuda')
for _ in range(5):
    ops.chol_inv(A)
torch.cuda.synchronize()
_lib.prof_enable(True)
for _ in range(100):
    ops.chol_inv(A)
torch.cuda.synchronize()
ms, cnt = _lib.prof_read('chol_inv_small')
print('chol_inv_small n = 100 x 30: %.1f us per launch (%d)' % (1e3 * ms / cnt, cnt))
