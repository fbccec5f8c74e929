"""n = 2048 x 10 factorisation + inverse (BASELINE config 5's data-independent part) on its own, for rocprofv3 (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vargp_amd import ops  # noqa: E402
torch.manual_seed(0)
n, nb = int(os.environ.get('N', '2048')), 10
x = torch.randn(nb, n, 64, device='cuda')
K = (x @ x.mT) / 64 + torch.eye(n, device='cuda')
ops.set_cholesky_error_mode('defer')
for _ in range(3):
    L, T = ops.chol_inv(K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    L, T = ops.chol_inv(K)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('n = %d x %d: %.3f ms = %.3f of the f32-MFMA peak' % (n, nb, ms, nb * n ** 3 / (ms * 1e-3) / 1e12 / 157.3))
err = ((L @ L.mT) - K - 1e-4 * torch.eye(n, device='cuda')).abs().max().item()
print('max |L L^T - (K + eps I)| = %.2e, |T L - I| = %.2e' % (err, ((T @ L) - torch.eye(n, device='cuda')).abs().max().item()))
