"""cProfile of the driver's training loop (task 0 of synthetic Split-MNIST at Cfg2 shapes, --graph).  GPU box only."""
import cProfile
import os
import pstats
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'experiments'))
import vargp as drv  # noqa: E402  (experiments/vargp.py)
import vargp_amd  # noqa: E402
from vargp_amd.datasets import SplitMNIST  # noqa: E402

vargp_amd.set_cholesky_error_mode('defer')
torch.manual_seed(4)
ds = SplitMNIST('/nowhere', train=True, synthetic=True, n_synth=36000)
idx = torch.randperm(len(ds))
ds.filter_by_idx(idx[:-6000])
ds.filter_by_class([0, 1])
val = SplitMNIST('/nowhere', train=True, synthetic=True, n_synth=600)
log = drv.JsonlLogger(tempfile.mkdtemp())
args = dict(epochs=30, M=100, lr=3e-3, beta=10.0, batch_size=512, prev_params=[], logger=log, device='cuda', eval_interval=100000,
            graph=True, seed=4)
drv.train(0, ds, val, val, **dict(args, epochs=3))       # warm-up (programs, captures)
pr = cProfile.Profile()
pr.enable()
drv.train(0, ds, val, val, **args)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
import json
print([json.loads(l) for l in open(os.path.join(log.log_dir, 'scalars.jsonl')) if 'steps_per_s' in l])
