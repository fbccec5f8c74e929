"""Per-phase cycle accounting of t0_bwd_mid_kernel (a -DBM_STAMPS build of the library, VARGP_HIP_LIB): runs a few Cfg2
steps and prints the s_memtime differences between the stamps of workgroup (0, 0), thread 0.  GPU box only."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402
from vargp_amd.train import ElboTrainer  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model(dev)
tr = ElboTrainer(gp, lr=bench.LR, beta=bench.BETA, n_total=bench.N_TOTAL)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == 'tail':     # t0_puu_final_kernel (a -DTAIL_STAMPS build of elbo_t0.hip)
    out = (ctypes.c_ulonglong * 16)()
    fn = _lib.lib().vargp_debug_tail_stamps
    fn.restype, fn.argtypes = None, [ctypes.c_void_p]
    fn(out)
    v = list(out)
    names = {(0, 1): 'issue loads', (1, 3): 'MFMA s=0', (3, 4): 'epilogue s=0', (4, 5): 'MFMA s=1',
             (5, 6): 'epilogue s=1', (6, 7): 'MFMA s=2', (7, 8): 'epilogue s=2', (8, 13): 'gz stores'}
    for (i, j), n in names.items():
        print('%-32s %8d cycles' % (n, v[j] - v[i]))
    print('%-32s %8d cycles' % ('total', v[13] - v[0]))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'mat':      # t0_bwd_mat_body (a -DBMAT_STAMPS build of gemm.hip)
    out = (ctypes.c_ulonglong * 24)()
    fn = _lib.lib().vargp_debug_bmat_stamps
    fn.restype, fn.argtypes = None, [ctypes.c_void_p]
    fn(out)
    names = {(0, 1): 'loads + stage T, gG, L_S', (1, 2): 'gG L_S^T', (2, 3): 'T^T ga', (3, 5): 'gTT loads + barrier',
             (5, 6): 'stage gG2, Lu', (6, 7): 'gG2 Lu^T, T^T gG2 + atomics', (7, 8): 'barrier', (8, 9): 'gT -> X1 + barrier',
             (9, 10): 'w1 = gT T^T', (10, 11): 'S -> X2 + barrier', (11, 12): 'K loads, tmp = T^T S', (12, 13): 'tmp -> X1 + barriers',
             (13, 14): 'gK = tmp T', (14, 15): 'W -> X2 + barrier', (15, 16): 'rows out, row sums'}
    v = list(out)
    for (i, j), n in names.items():
        print('%-32s %8d cycles' % (n, v[j] - v[i]))
    print('%-32s %8d cycles' % ('total', v[16] - v[0]))
    sys.exit(0)
    v = list(out)
    for i, n in enumerate(names):
        print('%-32s %8d cycles' % (n, v[i + 1] - v[i]))
    print('%-32s %8d cycles' % ('total', v[16] - v[0]))
    sys.exit(0)
out = (ctypes.c_ulonglong * 16)()
fn = _lib.lib().vargp_debug_bm_stamps
fn.restype, fn.argtypes = None, [ctypes.c_void_p]
fn(out)
names = ['issue loads', 'wait loads + LDS stores', 'barrier', 'phase 1 (gW, ga)', 'tri product 1 + atomics', 'product 2 (gP)',
         'barrier', 'gP/T/K_uf -> LDS + barrier', 'tri product 3 + atomics', 'product 4 (gK_uf)', 'epilogue W_uf', 'barrier', 'tail']
v = list(out)
for i, n in enumerate(names):
    print('%-32s %8d cycles' % (n, v[i + 1] - v[i]))
print('%-32s %8d cycles' % ('total', v[12] - v[0]))
