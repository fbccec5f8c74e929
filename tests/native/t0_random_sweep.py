"""Randomised parity sweep of the first-task program against the fp64 oracle (GPU box; not part of pytest: minutes of CPU
oracle time).  `python tests/native/t0_random_sweep.py [n_cases] [seed]` draws shapes inside and around the limits of the
LDS-resident kernels (M <= 104, M % 4, B % 4, D % 4, S <= 8) and prints the worst relative errors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import vargp_oracle as orc  # noqa: E402
from helpers import rel_l2, to_dev  # noqa: E402
from gpu_common import build_gp, grads_of  # noqa: E402
from vargp_amd import noise  # noqa: E402

DEV = 'cuda:0'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst_s, worst_g, bad = 0.0, 0.0, []
    for it in range(n):
        S = int(rng.integers(1, 10))
        C = int(rng.integers(1, 13))
        F_ = int(rng.integers(1, 4))
        M = int(rng.choice([4, 8, 12, 20, 32, 36, 52, 60, 64, 68, 96, 100, 104, 23, 51, 77, 108]))
        D = int(rng.choice([4, 8, 36, 40, 64, 100, 260, 300, 33, 37]))
        B = int(rng.choice([4, 8, 36, 60, 64, 68, 128, 132, 200, 30, 65]))
        params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=100 + it, kind='gauss')
        gp = build_gp(params, prev, S, F_)
        with noise.inject(**to_dev(nz, DEV)):
            kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
            (2.0 * kl_h + kl_u + 7.0 * nll).backward()
        sc, og = orc.elbo_step(params, prev, x, y, nz, beta=2.0, n_total=7 * B)
        es = max(abs(v.item() - sc[k].item()) / max(abs(sc[k].item()), 1e-30)
                 for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)] if sc[k].item() != 0.0)
        eg = max(rel_l2(g.cpu(), og[k]) for k, g in grads_of(gp).items())
        worst_s, worst_g = max(worst_s, es), max(worst_g, eg)
        flag = '' if (es < 1e-4 and eg < 1e-3) else '   <-- above the test tolerances'
        if flag:
            bad.append((S, F_, C, M, D, B))
        print(f'S{S} F{F_} C{C} M{M} D{D} B{B}: scalars {es:.2e}  grads {eg:.2e}{flag}', flush=True)
        gp.release_programs() if hasattr(gp, 'release_programs') else None
    print(f'worst: scalars {worst_s:.2e}  grads {worst_g:.2e}  above tolerance: {bad}')


if __name__ == '__main__':
    main()
