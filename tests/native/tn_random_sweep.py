"""Randomised parity sweep of the block program (models with previous tasks, csrc/elbo_tn.hip) against the fp64 oracle (GPU
box; not part of pytest: minutes of CPU oracle time).  `python tests/native/tn_random_sweep.py [n_cases] [seed]` draws shapes
around the limits that matter to it -- M not a multiple of 4, one to four earlier tasks (two to five panels of the blocked
factorisation, last panel narrower than 50), ragged batches, ep_var_mean on and off -- and prints the worst relative errors."""
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
from vargp_amd import noise, ops  # noqa: E402

DEV = 'cuda:0'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ops.set_cholesky_error_mode('raise')
    worst_s, worst_g, bad, on_block = 0.0, 0.0, [], 0
    for it in range(n):
        S = int(rng.integers(1, 6))
        C = int(rng.integers(1, 7))
        F_ = int(rng.integers(1, 4))
        M = int(rng.choice([8, 20, 30, 33, 36, 52, 60, 64, 100, 104, 120]))
        n_prev = int(rng.integers(1, 5))
        D = int(rng.choice([2, 8, 36, 40, 64, 100, 33]))
        B = int(rng.choice([8, 36, 64, 68, 128, 200, 30, 65]))
        nomean = bool(rng.integers(0, 4) == 0)
        kind = 'wtoy' if D == 2 else 'gauss'
        params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=300 + it, kind=kind)
        gp = build_gp(params, prev, S, F_, ep_var_mean=not nomean)
        on_block += int(bool(gp._use_block_program(B)))
        with noise.inject(**to_dev(nz, DEV)):
            kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
            (2.0 * kl_h + kl_u + 7.0 * nll).backward()
        sc, og = orc.elbo_step(params, prev, x, y, nz, beta=2.0, n_total=7 * B, ep_var_mean=not nomean)
        es = max(abs(float(v) - sc[k].item()) / max(abs(sc[k].item()), 1e-30)
                 for k, v in [('kl_hypers', kl_h), ('kl_u', kl_u), ('nll', nll)] if sc[k].item() != 0.0)
        eg = max(rel_l2(g.cpu(), og[k]) for k, g in grads_of(gp).items())
        worst_s, worst_g = max(worst_s, es), max(worst_g, eg)
        flag = '' if (es < 1e-4 and eg < 1e-3) else '   <-- above the test tolerances'
        if flag:
            bad.append((S, F_, C, M, n_prev, D, B, nomean))
        print(f'S{S} F{F_} C{C} M{M} t{n_prev} D{D} B{B} nomean={int(nomean)} block={int(bool(gp._use_block_program(B)))}: '
              f'scalars {es:.2e}  grads {eg:.2e}{flag}', flush=True)
        gp.release_programs() if hasattr(gp, 'release_programs') else None
    print(f'worst: scalars {worst_s:.2e}  grads {worst_g:.2e}  on the block program: {on_block}/{n}  above tolerance: {bad}')


if __name__ == '__main__':
    main()
