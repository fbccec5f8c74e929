"""cProfile + wall clock of the drop-in loop (reference loop shape through the var_gp alias, eager, 'defer' mode).  GPU box only."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vargp_amd import ops  # noqa: E402
from vargp_amd.optim import Yogi  # noqa: E402

ops.set_cholesky_error_mode(sys.argv[1] if len(sys.argv) > 1 else 'defer')
gp, x, y = bench.make_model('cuda:0')
optim = Yogi(gp.parameters(), lr=3e-3)


def step():
    optim.zero_grad()
    kl_hypers, kl_u, lik = gp.loss(x, y)
    loss = 10.0 * kl_hypers + kl_u + (12000 / x.size(0)) * lik
    loss.backward()
    optim.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('300 steps: host enqueue %.3f ms/step, wall %.3f ms/step' % (1e3 * t_host / 300, 1e3 * t_all / 300))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
