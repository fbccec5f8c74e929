"""Host time of each phase of the drop-in loop, accumulated over 300 steps without synchronising (GPU box only)."""
import os
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
acc = [0.0] * 5
pc = time.perf_counter


def step(rec):
    t0 = pc(); optim.zero_grad()
    t1 = pc(); kl_hypers, kl_u, lik = gp.loss(x, y)
    t2 = pc(); loss = 10.0 * kl_hypers + kl_u + (12000 / x.size(0)) * lik
    t3 = pc(); loss.backward()
    t4 = pc(); optim.step()
    t5 = pc()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d


for _ in range(30):
    step(False)
torch.cuda.synchronize()
t0 = pc()
for _ in range(300):
    step(True)
th = pc() - t0
torch.cuda.synchronize()
print('host %.3f ms/step wall %.3f ms/step; zero_grad %.0f  loss %.0f  combine %.0f  backward %.0f  optim.step %.0f  (us/step)' %
      (1e3 * th / 300, 1e3 * (pc() - t0) / 300, *[1e6 * a / 300 for a in acc]))
