"""Per-phase cycle accounting (s_memtime stamps of workgroup 0) of the LDS-resident kernels of the first-task step, in a tuning
build of the library (VARGP_HIP_LIB=<that .so>).  GPU box only.

    python tests/native/bm_stamps.py          t0_bwd_mid_kernel         -DBM_STAMPS    on elbo_t0.hip
    python tests/native/bm_stamps.py ff       t0_fwd_fused_kernel       -DFF_STAMPS    on elbo_t0.hip   (per wave)
    python tests/native/bm_stamps.py tail     t0_puu_final_kernel       -DTAIL_STAMPS  on elbo_t0.hip
    python tests/native/bm_stamps.py mat      t0_bwd_mat_body (+ wall-clock spans of the two roles of its launch)
                                                                        -DBMAT_STAMPS  on gemm.hip      (per wave)
Build recipe (from vargp_amd/csrc, after `make`): compile the one translation unit with the switch and link it with the other
objects of build/ into a second library, e.g.
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFF_STAMPS -c elbo_t0.hip -o /tmp/t0.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvargp_ff.so build/core.o build/gemm.o build/rbf.o build/chol.o \
          build/elbo_ops.o /tmp/t0.o build/elbo_tn.o
The stamps cost nothing in the shipped build (the macros are empty)."""
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
bench.S = int(os.environ.get('VARGP_BM_S', '3'))      # hyper-samples (8: 640 tile workgroups, 2.5 per CU)
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
if len(sys.argv) > 1 and sys.argv[1] == 'ff':       # t0_fwd_fused_kernel (a -DFF_STAMPS build of elbo_t0.hip)
    out = (ctypes.c_ulonglong * 64)()
    fn = _lib.lib().vargp_debug_ff_stamps
    fn.restype, fn.argtypes = None, [ctypes.c_void_p]
    fn(out)
    names = {(0, 1): 'KL loads + sums', (1, 2): 'stage T, G, K_uf tile + barrier', (1, 10): '  all loads issued', (10, 11): '  T in LDS (first wait)',
             (11, 12): '  G, K_uf tile in LDS', (12, 13): '  KL arithmetic', (13, 2): '  barrier', (2, 3): 'P = T K_uf', (3, 4): 'barrier',
             (4, 5): 'P -> LDS, global; column sums + barrier', (5, 6): 'W = G^T P', (6, 7): 'W out, sums', (7, 8): 'column reductions, mu / var',
             (8, 9): 'KL block sum'}
    v = [list(out)[16 * w:16 * w + 16] for w in range(4)]
    print('%-44s %s' % ('cycles per wave', ''.join('%9s' % ('wave %d' % w) for w in range(4))))
    for (i, j), n in names.items():
        print('%-44s %s' % (n, ''.join('%9d' % (v[w][j] - v[w][i]) for w in range(4))))
    print('%-44s %s' % ('total', ''.join('%9d' % (v[w][9] - v[w][0]) for w in range(4))))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'mat':      # t0_bwd_mat_body (a -DBMAT_STAMPS build of gemm.hip)
    out = (ctypes.c_ulonglong * 96)()                # [wave][stamp]
    fn = _lib.lib().vargp_debug_bmat_stamps
    fn.restype, fn.argtypes = None, [ctypes.c_void_p]
    fn(out)
    names = {(0, 1): 'loads + stage gG, L_S', (1, 2): 'gG L_S^T', (2, 3): 'stage T, ga, m + barrier', (3, 5): 'stage gG2, Lu; T^T ga + barrier',
             (5, 6): 'g_u_mean atomics', (6, 7): 'gG2 Lu^T, T^T gG2 + atomics', (7, 8): 'barrier', (8, 9): 'gT -> X1 + barrier',
             (9, 10): 'w1 = gT T^T', (10, 11): 'S -> X2 + barrier', (11, 12): 'K loads, tmp = T^T S', (12, 13): 'tmp -> X1 + barriers',
             (13, 14): 'gK = tmp T', (14, 15): 'W -> X2 + barrier', (15, 16): 'rows out, row sums'}
    v = [list(out)[24 * w:24 * w + 24] for w in range(4)]
    print('%-32s %s' % ('cycles per wave', ''.join('%9s' % ('wave %d' % w) for w in range(4))))
    for (i, j), n in names.items():
        print('%-32s %s' % (n, ''.join('%9d' % (v[w][j] - v[w][i]) for w in range(4))))
    print('%-32s %s' % ('total', ''.join('%9d' % (v[w][16] - v[w][0]) for w in range(4))))
    # wall-clock spans of the two roles over ONE more step (100 MHz ticks -> us)
    sp = _lib.lib().vargp_debug_bmat_span
    sp.restype, sp.argtypes = None, [ctypes.c_void_p, ctypes.c_int]
    span = (ctypes.c_ulonglong * 8)()
    tr.capture(x, y)
    for _ in range(300):                              # steady state (clocks up) before the step that is measured
        tr.step_graph()
    torch.cuda.synchronize()
    sp(span, 1)
    tr.step_graph()
    torch.cuda.synchronize()
    sp(span, 0)
    t = list(span)
    t0 = min(t[0], t[4])
    for r, n in ((0, 'matrix chains'), (1, 'P_uf tiles')):
        print('%-14s first start %6.2f us, last start %6.2f, last end %6.2f' % (n, (t[4 * r] - t0) / 100., (t[4 * r + 2] - t0) / 100.,
                                                                                  (t[4 * r + 1] - t0) / 100.))
    su = v[3][17:22]
    print('S_u role (first class, wave 0): samples %d, stage gL / T_S / L_S %d, L_S^T gL and S -> X2 %d, tail %d, total %d cycles'
          % (su[1] - su[0], su[2] - su[1], su[3] - su[2], su[4] - su[3], su[4] - su[0]))
    sys.exit(0)
if os.environ.get('VARGP_BM_MULTI'):     # t0_bwd_mid_multi_kernel: second tile of the stamped workgroup (+ its set-up and whole life), per wave
    out = (ctypes.c_ulonglong * 96)()
    fn = _lib.lib().vargp_debug_bmm_stamps
    fn.restype, fn.argtypes = None, [ctypes.c_void_p]
    fn(out)
    v = [list(out)[24 * w:24 * w + 24] for w in range(4)]
    names = {(20, 21): 'set-up: first tile loads issued, G / T / a staged', (0, 1): 'tile top',
             (1, 15): 'P, W -> LDS', (15, 2): 'K_uf load issue, gmu / gvar + barrier', (2, 3): 'phase 1 (gW, ga) + barrier',
             (3, 4): 'phase 2: accG += , gP', (4, 5): 'barrier', (5, 6): 'phase 3: gP / K_uf -> LDS, next P / W loads + barrier',
             (6, 7): 'phase 4: accT +=, gK_uf', (7, 16): 'W_uf = gK_uf o K_uf', (16, 17): '  barrier', (17, 8): '  W_uf -> LDS, next likelihood loads + barrier',
             (8, 9): 'W_uf rows out, row / column sums + barrier', (9, 10): 'c_uf atomics', (0, 10): 'second tile, total',
             (20, 22): 'workgroup, total'}
    print('%-56s %s' % ('cycles per wave', ''.join('%9s' % ('wave %d' % w) for w in range(4))))
    for (i, j), n in names.items():
        print('%-56s %s' % (n, ''.join('%9d' % (v[w][j] - v[w][i]) for w in range(4))))
    sys.exit(0)
out = (ctypes.c_ulonglong * 16)()
fn = _lib.lib().vargp_debug_bm_stamps
fn.restype, fn.argtypes = None, [ctypes.c_void_p]
fn(out)
names = {(0, 13): 'issue all loads', (13, 14): 'softmax: wait its loads + evaluate', (14, 15): 'f-groups meet in LDS + barrier',
         (15, 1): 'column sums, nll atomic', (1, 2): 'wait G/P/W, LDS stores + barrier', (2, 3): 'phase 1 (gW, ga) + barrier',
         (3, 4): 'phase 2: tril(P gW^T) atomics, gP', (4, 6): 'barrier', (6, 7): 'phase 3: gP/T/K_uf -> LDS + barrier',
         (7, 8): 'phase 4: tril(gP K_uf^T) atomics, gK_uf', (8, 10): 'tail atomics, W_uf -> LDS', (10, 11): 'barrier',
         (11, 12): 'W_uf rows out, row / column sums'}
v = list(out)
for (i, j), n in names.items():
    print('%-44s %8d cycles' % (n, v[j] - v[i]))
print('%-44s %8d cycles' % ('total', v[12] - v[0]))
