"""Diagnostic: which shapes break hipGraph replay of the fused step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vargp_amd.kernels import RBFKernel
from vargp_amd.likelihoods import MulticlassSoftmax
from vargp_amd.train import ElboTrainer
from vargp_amd.vargp import VARGP
from vargp_amd.synthetic import mnist_like
import vargp_amd
from vargp_amd import ops

dev = 'cuda'
vargp_amd.set_cholesky_error_mode('defer')
fused = os.environ.get('FUSED', '1') == '1'
from vargp_amd import fused as _f
if os.environ.get('KEEP'):
    _f._DEBUG_KEEP = {}
for (C, M, B, D) in [(10, 16, 32, 784)]:
    torch.manual_seed(0)
    xall, yall = mnist_like(4096, D, C, kind='gauss', seed=1)
    z = torch.stack([xall[yall == c][:M] for c in range(C)])
    gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=10), n_var_samples=3).to(dev)
    gp.fused_first_task = fused
    tr = ElboTrainer(gp, lr=3e-3, beta=10.0, n_total=12000)
    x, y = xall[:B].to(dev), yall[:B].to(dev)
    tr.capture(x, y)
    res = []
    for it in range(3):
        out = tr.step_graph(x, y)
        torch.cuda.synchronize()
        res.append({n: int((~torch.isfinite(p.grad)).sum()) for n, p in gp.named_parameters()})
        res[-1]['out'] = [round(o.item(), 2) for o in out]
        if _f._DEBUG_KEEP and it == 1:
            print('after replay 1:', {k: int((~torch.isfinite(v)).sum()) for k, v in _f._DEBUG_KEEP.items() if v.is_floating_point() and int((~torch.isfinite(v)).sum())})
            print('   max abs:', {k: '%.2e' % v.abs().max().item() for k, v in _f._DEBUG_KEEP.items() if v.is_floating_point() and v.numel()})
    print('fused', fused, 'C M B D', (C, M, B, D), 'grads finite per replay', res)
    if _f._DEBUG_KEEP:
        print({k: int((~torch.isfinite(v)).sum()) for k, v in _f._DEBUG_KEEP.items() if v.is_floating_point()})
