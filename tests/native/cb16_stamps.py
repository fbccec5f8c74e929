"""Per-step cycle accounting of the blocked pivot chain (csrc/chol_blk16.h; a -DVARGP_CB16_STAMPS build of chol.hip):
VARGP_HIP_LIB=<that .so> VARGP_CHOL_F32_ALONE=1 python tests/native/cb16_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402
ops.set_cholesky_error_mode('defer')
A = torch.randn(30, 100, 100, device='cuda')
A = A @ A.mT / 100 + torch.eye(100, device='cuda')
for _ in range(3):
    ops.chol_inv(A)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
ctypes.CDLL(_lib.LIB_PATH).vargp_debug_cb16_stamps(buf)
t = [[[buf[(w * 8 + s) * 4 + i] for i in range(4)] for s in range(8)] for w in range(4)]
t0 = min(t[w][0][0] for w in range(4) if t[w][0][0])
for s in range(7):
    print('step %d' % s, '  '.join('w%d: S1@%6d B1@%6d B2@%6d S3 done@%6d' % (w, t[w][s][0] - t0, t[w][s][1] - t0, t[w][s][2] - t0, t[w][s][3] - t0) for w in range(4)))
print('end', [t[w][7][0] - t0 for w in range(4)], 'written', [t[w][7][1] - t0 for w in range(4)])
