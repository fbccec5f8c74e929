"""Experiment (not a test): do two independent kernel branches of a captured hipGraph overlap on replay?
Run under `rocprofv3 --kernel-trace` and inspect the timestamps of chol_inv_small_kernel vs gemm_kernel."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import vargp_amd
from vargp_amd import ops

dev = 'cuda:0'
vargp_amd.set_cholesky_error_mode('defer')
A = torch.randn(40, 100, 100, device=dev)
A = A @ A.mT + 100 * torch.eye(100, device=dev)
X = torch.randn(8, 1024, 1024, device=dev)
side = torch.cuda.Stream()

def work(fork):
    main = torch.cuda.current_stream()
    if fork:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            L, T = ops.chol_inv(A)
        Y = ops.bgemm(X, X)
        main.wait_stream(side)
    else:
        L, T = ops.chol_inv(A)
        Y = ops.bgemm(X, X)
    return L, Y

for fork in (False, True):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            work(fork)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = work(fork)
    torch.cuda.synchronize()
    for name, fn in (('graph', g.replay), ('eager', lambda: work(fork))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f'fork={fork} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per iteration', flush=True)
