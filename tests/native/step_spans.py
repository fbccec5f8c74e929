"""Time line of one Cfg2 step as the GPU sees it (a -DSTEP_SPANS build of the library, VARGP_HIP_LIB): wall-clock (100 MHz) of the
first workgroup's start and the last workgroup's end of each of the eight kernels, in steady state under graph replay, to set
against the dispatch-to-completion durations rocprofv3 reports.  GPU box only.
Build: core.hip, gemm.hip and elbo_t0.hip compiled with -DSTEP_SPANS and linked with the other objects of vargp_amd/csrc/build/
into a second library (recipe: bm_stamps.py)."""
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
tr.capture(x, y)
fns = []
for tu in ('gemm', 't0', 'core'):
    fn = getattr(_lib.lib(), 'vargp_debug_spans_' + tu)
    fn.restype, fn.argtypes = None, [ctypes.c_void_p, ctypes.c_int]
    fns.append(fn)
names = ['pro_kuu', 'chol || K_uf', 'small-column product', 'fwd_fused', 'bwd_mid', 'chains || P_uf', 'puu_final', 'yogi']
acc = [[0.0, 0.0, 0.0] for _ in names]
nrep = 20
role = [0.0, 0.0]
for rep in range(nrep):
    for _ in range(100 if rep == 0 else 10):       # steady state (clocks up) before the step that is measured
        tr.step_graph()
    torch.cuda.synchronize()
    for fn in fns:
        fn(None, 1)
    tr.step_graph()
    torch.cuda.synchronize()
    t = [None] * 12
    for fn in fns:
        out = (ctypes.c_ulonglong * 48)()
        fn(out, 0)
        v = list(out)
        for i in range(12):
            if v[4 * i + 1] != 0:
                t[i] = (v[4 * i], v[4 * i + 1], v[4 * i + 2])
    role[0] += (t[8][1] - t[5][0]) / 100.
    role[1] += (t[9][1] - t[1][0]) / 100.
    t0 = t[0][0]
    for i in range(8):
        nxt = t[i + 1][0] if i + 1 < 8 else t[i][1]
        acc[i][0] += (t[i][0] - t0) / 100.
        acc[i][1] += (t[i][1] - t[i][0]) / 100.
        acc[i][2] += (nxt - t[i][1]) / 100.
print('%-22s %10s %12s %22s' % ('kernel', 'start us', 'busy span us', 'gap to next start us'))
for i, n in enumerate(names):
    print('%-22s %10.2f %12.2f %22.2f' % (n, acc[i][0] / nrep, acc[i][1] / nrep, acc[i][2] / nrep))
print('last matrix chain of the backward ends %.2f us after its kernel\'s start, last pivot chain of the forward %.2f us'
      % (role[0] / nrep, role[1] / nrep))
print('sum of busy spans %.1f us, sum of gaps %.1f us' % (sum(a[1] for a in acc) / nrep, sum(a[2] for a in acc[:-1]) / nrep))

# start / end of every matrix chain of the backward's merged launch in the last measured step (relative to the kernel's start)
fn = _lib.lib().vargp_debug_bmat_ends
fn.restype, fn.argtypes = None, [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 128)()
fn(out)
v = list(out)
k0 = t[5][0]
print('matrix chains (id: start .. end us):')
print('  '.join('%d: %.1f..%.1f' % (i, (v[2 * i] - k0) / 100., (v[2 * i + 1] - k0) / 100.) for i in range(40)))
