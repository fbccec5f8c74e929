"""Which preceding workload makes bench.py's secondary smnist_t1 run slow (mean 2x its median)?  GPU box only."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = argparse.Namespace(eager=False, no_replay=False, no_cpu_baseline=True, comm='allreduce', stress_n=1000000)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
seq = sys.argv[1].split(',')
for name in seq:
    if name == 'dropin':
        r = bench.dropin_workload(args, dev)
        print(name, round(r['value']), round(r['value_defer']))
    else:
        r = bench.run_workload(name, args, dev, 1, 0, False, 30, 3, primary=False, kern_n=50)
        print(name, round(r['value'], 1), round(r['ms_per_step'], 3), round(r['ms_per_step_median'], 3))
