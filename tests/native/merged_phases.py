"""Timeline of the merged factorisation + K_uf launch of the first-task forward (tuning build of gemm.hip with
-DVARGP_CHOL_PHASES): 100 MHz wall-clock stamps of the phases of the first K_uu chain, the last chain (an S_u matrix), the
first and the last GEMM tile.  GPU box only:  VARGP_HIP_LIB=<a -DVARGP_CHOL_PHASES build of the library: recipe in bm_stamps.py> python tests/native/merged_phases.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402
from vargp_amd.train import ElboTrainer  # noqa: E402

ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model('cuda:0')
tr = ElboTrainer(gp, lr=3e-3, beta=10.0, n_total=12000)
buf = (ctypes.c_ulonglong * 64)()
fn = _lib.lib()._handle if False else ctypes.CDLL(_lib.LIB_PATH).vargp_debug_chol_phases
fn(buf, 39)              # matrix 39 = the last S_u chain (S C + C = 40 matrices)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
fn(buf, 39)
t = list(buf)
t0 = min(v for v in (t[0], t[16], t[32], t[40]) if v)
us = lambda v: (v - t0) / 100.0
names = ['start', 'loaded', 'eliminated', 'staged', 'stores issued', 'ld:norms', 'ld:loads done', 'ld:end', 'ld:K written', 'ld:barrier']
print('K_uu chain 0 :', '  '.join('%s %.1f' % (n, us(t[i])) for i, n in enumerate(names) if t[i]))
print('S_u chain 39 :', '  '.join('%s %.1f' % (n, us(t[16 + i])) for i, n in enumerate(names) if t[16 + i]))
print('first GEMM tile: %.1f .. %.1f   last GEMM tile: %.1f .. %.1f' % (us(t[32]), us(t[33]), us(t[40]), us(t[41])))
