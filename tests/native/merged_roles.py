"""Clean timing of the roles of the first-task forward's merged launch (pivot chains || K_uf product) on VALID data: forward only
(the parameters never change, so a role switched off by VARGP_EXP_MERGED cannot feed NaNs back into the next call), hipEvent
timing of the launch site (vargp_prof_enable / vargp_prof_read).  GPU box:
    for e in 0 1 2; do VARGP_EXP_MERGED=$e python tests/native/merged_roles.py; done      (0: both roles, 1: chains only, 2: product only)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402

ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model('cuda:0')
a = torch.randn(4096, 4096, device='cuda')
for _ in range(30):                     # clocks up
    a @ a
with torch.no_grad():
    for _ in range(20):
        gp.loss(x, y)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(200):
        gp.loss(x, y)
    torch.cuda.synchronize()
for tag in ('t0_pro_kuu', 'chol_rbf_gemm', 't0_qps_gemm'):
    ms, n = _lib.prof_read(tag)
    if n:
        print('VARGP_EXP_MERGED=%s  %-14s %7.2f us  (%d launches)' % (os.environ.get('VARGP_EXP_MERGED', '0'), tag, 1e3 * ms / n, n))
