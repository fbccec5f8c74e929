"""Diagnostic: toy accuracy trajectory and NaN hunt on the synthetic Split-MNIST surrogate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vargp_amd.datasets import ToyDataset, SplitMNIST
from vargp_amd.train import ElboTrainer
from vargp_amd.train_utils import set_seeds, compute_accuracy
from vargp_amd.vargp import VARGP
import vargp_amd
from vargp_amd import ops

dev = 'cuda'
set_seeds(1)
ds = ToyDataset()
ds.filter_by_class([0, 1])
gp = VARGP.create_clf(ds, M=20, n_f=10, n_var_samples=3).to(dev)
tr = ElboTrainer(gp, lr=1e-2, beta=1.0, n_total=len(ds))
x, y = ds[torch.arange(len(ds))]
x, y = x.to(dev), y.to(dev)
for it in range(3001):
    out = tr.step(x, y)
    if it % 500 == 0:
        print('toy', it, [round(o.item(), 2) for o in out], 'acc', compute_accuracy(ds, gp, device=dev))

set_seeds(2)
vargp_amd.set_cholesky_error_mode('defer')
ds = SplitMNIST('/nonexistent', train=True, synthetic=True, n_synth=3000)
ds.filter_by_class([0, 1])
xa, ya = ds[torch.arange(len(ds))]
d2 = torch.cdist(xa[:200], xa[:200]).pow(2)
print('smnist-syn: n', len(ds), 'same-class d2 median', d2[ya[:200, None] == ya[None, :200]].median().item(),
      'diff-class', d2[ya[:200, None] != ya[None, :200]].median().item())
gp = VARGP.create_clf(ds, M=20, n_f=10, n_var_samples=3).to(dev)
tr = ElboTrainer(gp, lr=3e-3, beta=10.0, n_total=len(ds))
for it in range(40):
    idx = torch.randperm(len(ds))[:512]
    out = tr.step(xa[idx].to(dev), ya[idx].to(dev))
    vals = [o.item() for o in out]
    if it % 5 == 0 or any(v != v for v in vals):
        with torch.no_grad():
            mu, var = gp(xa[:512].to(dev))
        print('smnist', it, [round(v, 2) for v in vals], 'chol failures', ops.linalg_error_count(), 'min var %.3e' % var.min().item(),
              'nan var', torch.isnan(var).any().item(), 'acc', None if any(v != v for v in vals) else compute_accuracy(ds, gp, device=dev))
    if any(v != v for v in vals):
        break
