"""CPU, gloo, world 2 and 3: the CLASS-sharded ELBO step of vargp_amd.train.ElboTrainer (train.split_pairs: more ranks than
hyper-samples -> every rank holds one sample and a contiguous range of classes; all-gather of the predictive moments mu, var
(S, C, B) before the softmax likelihood -- which couples the classes, var_gp/likelihoods.py:26-29 -- then each rank's own
backward and the usual sum of the flat [grads | kl_u | nll] buffer) reproduces the single-process step on all S x C pairs.
The local compute is the oracle here (no GPU in this container); tests/test_hip_dist.py runs the same route on the HIP programs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vargp_oracle as orc

F_, C, M, D, B = 3, 4, 6, 5, 16
SEED = 123
NAMES = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _problem(S, n_prev):
    return orc.make_problem(S, F_, C, M, D, B, n_prev=n_prev, seed=19, kind='toy')


def _worker(rank, world, port, S, n_prev, comm, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vargp_amd.train import ElboTrainer, split_pairs
        torch.set_num_threads(1)
        params, prev, x, y, _ = _problem(S, n_prev)
        leaves = {k: torch.nn.Parameter(params[k].clone()) for k in NAMES}
        full = dict(params, **leaves)

        def moments_fn(xb, rect, eps_theta):
            s0, s1, c0, c1 = rect
            sl = dict(full)
            for k in ('z', 'u_mean', 'u_tril_vec'):
                sl[k] = full[k][c0:c1]
            pv = [{k: p[k][c0:c1] for k in p} for p in prev]
            nz = dict(eps_theta=eps_theta)
            if n_prev:           # ep_var_mean=True: the KL does not depend on the u_<t sample
                nz['eps_u'] = torch.zeros(1, s1 - s0, c1 - c0, n_prev * M)
            pmu, pvar, (mu_q, Lq, mu_p, Lp) = orc.forward(sl, pv, xb, nz, want_kl=True)
            kl_u = orc.mvn_kl(mu_q, Lq, mu_p, Lp).sum(-1).mean(0).mean(0)
            kl_h = orc.kl_hypers(full['log_mean'], full['log_logvar'], full['prior_log_mean'], full['prior_log_logvar'])
            return kl_h, kl_u, pmu, pvar

        lik_fn = lambda mu, var, yb, eps_f: orc.softmax_nll(mu, var, yb, eps_f)
        shards = split_pairs(S, C, world)
        tr = ElboTrainer(None, beta=2.0, n_total=64, noise_seed=SEED, params=[leaves[k] for k in NAMES],
                         optimizer=lambda ps: torch.optim.SGD(ps, lr=0.05), comm=comm, shards=shards, grid=(S, C),
                         pair_fns=(moments_fn, lik_fn), pair_dims=(D + 1, F_))
        assert tr.class_split and tr.rect == shards[rank]
        res = []
        for _ in range(steps):
            kl_h, kl_u, nll = tr.step(x, y)
            res.append(dict(kl_h=kl_h.item(), kl_u=kl_u.item(), nll=nll.item(),
                            grads={k: leaves[k].grad.detach().clone().numpy() for k in NAMES}))
        if rank == world - 1:
            out.put((res, {k: leaves[k].detach().clone().numpy() for k in NAMES}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,S,n_prev,comm', [(2, 1, 0, 'allreduce'), (3, 2, 0, 'allreduce'), (3, 1, 1, 'rsag'), (3, 2, 1, 'allreduce')])
def test_class_sharded_step_matches_single_process(world, S, n_prev, comm):
    """world 3, S 2, C 4: ranks 0, 1 hold sample 0's classes {0, 1} and {2, 3}, rank 2 all of sample 1 (uneven rectangles);
    S = 1: every rank a class range of the one sample.  Two SGD steps, so that the second sees the first one's update."""
    from vargp_amd.train import split_pairs
    steps = 2
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, n_prev, comm, steps, out)) for r in range(world)]
    for p in procs:
        p.start()
    got, final = out.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    rects = split_pairs(S, C, world)
    assert sum((r[1] - r[0]) * (r[3] - r[2]) for r in rects) == S * C and any(r[3] - r[2] < C for r in rects)
    params, prev, x, y, _ = _problem(S, n_prev)
    gen = torch.Generator().manual_seed(SEED)
    p = dict(params)
    for k in range(steps):
        nz = dict(eps_theta=torch.randn(S, D + 1, generator=gen), eps_f=torch.randn(S, F_, C, B, generator=gen))
        if n_prev:
            nz['eps_u'] = torch.zeros(1, S, C, n_prev * M)
        sc, grads = orc.elbo_step(p, prev, x, y, nz, beta=2.0, n_total=64)
        np.testing.assert_allclose(got[k]['kl_h'], sc['kl_hypers'].item(), rtol=1e-5)
        np.testing.assert_allclose(got[k]['kl_u'], sc['kl_u'].item(), rtol=2e-5)
        np.testing.assert_allclose(got[k]['nll'], sc['nll'].item(), rtol=2e-5)
        for name, g in grads.items():
            err = np.linalg.norm(got[k]['grads'][name] - g.numpy()) / np.linalg.norm(g.numpy())
            assert err < 1e-4, (k, name, err)
        p = dict(p, **{name: p[name] - 0.05 * grads[name] for name in NAMES})
    for name in NAMES:
        np.testing.assert_allclose(final[name], p[name].numpy(), rtol=1e-4, atol=1e-6)


def test_split_pairs_rectangles():
    from vargp_amd.train import split_pairs
    assert split_pairs(64, 10, 8) == [(8 * r, 8 * r + 8, 0, 10) for r in range(8)]                # BASELINE config 4: whole samples
    assert split_pairs(3, 10, 2) == [(0, 2, 0, 10), (2, 3, 0, 10)]
    cfg2 = split_pairs(3, 10, 8)                                                                  # BASELINE config 2 on 8 GPUs
    assert cfg2 == [(0, 1, 0, 4), (0, 1, 4, 7), (0, 1, 7, 10), (1, 2, 0, 4), (1, 2, 4, 7), (1, 2, 7, 10), (2, 3, 0, 5), (2, 3, 5, 10)]
    assert max((r[1] - r[0]) * (r[3] - r[2]) for r in cfg2) == 5
    for S, Cc, w in [(3, 10, 4), (1, 10, 8), (2, 4, 3), (10, 10, 7), (3, 10, 30)]:
        rs = split_pairs(S, Cc, w)
        seen = set()
        for s0, s1, c0, c1 in rs:
            for s_ in range(s0, s1):
                for c_ in range(c0, c1):
                    assert (s_, c_) not in seen
                    seen.add((s_, c_))
        assert len(rs) == w and len(seen) == S * Cc
