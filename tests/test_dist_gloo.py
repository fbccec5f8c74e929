"""CPU, world_size 2 over gloo: the sample-parallel exchange of vargp_amd.train.ElboTrainer
(noise sharding + one all-reduce of [grads | kl_u | nll]) reproduces the single-process result for
the full sample set.  The local compute is the oracle here (no GPU in this container); on the GPU box
the same trainer drives the HIP model (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vargp_oracle as orc

S_LOCAL, WORLD, F_, C, M, D, B = 2, 2, 3, 4, 6, 5, 16
SEED = 77


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _problem(n_prev):
    return orc.make_problem(S_LOCAL * WORLD, F_, C, M, D, B, n_prev=n_prev, seed=9, kind='toy')


def _worker(rank, port, n_prev, out, counts=None, comm='allreduce'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from vargp_amd import noise
        from vargp_amd.train import ElboTrainer
        torch.set_num_threads(1)
        params, prev, x, y, _ = _problem(n_prev)
        names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
        leaves = {k: torch.nn.Parameter(params[k].clone()) for k in names}
        full = dict(params, **leaves)
        Mt_prev = n_prev * M
        s_loc = S_LOCAL if counts is None else counts[rank]
        s_tot = S_LOCAL * WORLD if counts is None else sum(counts)

        def loss_fn(xb, yb):
            nz = dict(eps_theta=noise.draw('eps_theta', (s_loc, D + 1), 'cpu'))
            if n_prev:
                nz['eps_u'] = noise.draw('eps_u', (s_tot, s_loc, C, Mt_prev), 'cpu', sample_dim=1)
            nz['eps_f'] = noise.draw('eps_f', (s_loc, F_, C, B), 'cpu')
            return orc.loss(full, prev, xb, yb, nz)

        tr = ElboTrainer(None, beta=2.0, n_total=64, noise_seed=SEED, params=[leaves[k] for k in names],
                         loss_fn=loss_fn, optimizer=lambda ps: torch.optim.SGD(ps, lr=0.0), sample_counts=counts, comm=comm)
        assert tr.world == WORLD and tr.rank == rank
        kl_h, kl_u, nll = tr.step(x, y)
        if rank == 0:
            out.put(dict(kl_h=kl_h.item(), kl_u=kl_u.item(), nll=nll.item(),
                         grads={k: leaves[k].grad.detach().clone().numpy() for k in names}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_prev,counts,comm', [(0, None, 'allreduce'), (1, None, 'allreduce'), (0, [2, 1], 'allreduce'),
                                                (1, [1, 2], 'allreduce'), (0, None, 'rsag'), (1, [1, 2], 'rsag')])
def test_sample_parallel_matches_single_process(n_prev, counts, comm):
    """counts: a sample total that the ranks cannot divide evenly (3 samples over 2 ranks): uneven shards, weights S_r / S.
    comm: one all-reduce, or reduce-scatter + all-gather of the same (padded) flat buffer -- the parameter count of the
    problem (odd: 2 scalars + an odd number of gradient entries is not required; the buffer is padded either way)."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, n_prev, out, counts, comm)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single process, all S_LOCAL*WORLD samples, same global noise (same generator seed, same draw order)
    params, prev, x, y, _ = _problem(n_prev)
    gen = torch.Generator().manual_seed(SEED)
    S = S_LOCAL * WORLD if counts is None else sum(counts)
    nz = dict(eps_theta=torch.randn(S, D + 1, generator=gen))
    if n_prev:
        # each rank draws (n_v, world*S_local, ...) and keeps its slice of dim 1; n_v = S_total here
        nz['eps_u'] = torch.randn(S, S, C, n_prev * M, generator=gen)
    nz['eps_f'] = torch.randn(S, F_, C, B, generator=gen)
    sc, grads = orc.elbo_step(params, prev, x, y, nz, beta=2.0, n_total=64)
    np.testing.assert_allclose(got['kl_h'], sc['kl_hypers'].item(), rtol=1e-5)
    np.testing.assert_allclose(got['kl_u'], sc['kl_u'].item(), rtol=2e-5)
    np.testing.assert_allclose(got['nll'], sc['nll'].item(), rtol=2e-5)
    for k, g in grads.items():
        err = np.linalg.norm(got['grads'][k] - g.numpy()) / np.linalg.norm(g.numpy())
        assert err < 1e-4, (k, err)


def _worker_stable(rank, port, comm, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from vargp_amd import noise
        from vargp_amd.train import ElboTrainer
        torch.set_num_threads(1)
        params, prev, x, y, _ = _problem(0)
        names = ['z', 'u_mean', 'u_tril_vec', 'log_mean', 'log_logvar']
        leaves = {k: torch.nn.Parameter(params[k].clone()) for k in names}
        full = dict(params, **leaves)

        def loss_fn(xb, yb):
            nz = dict(eps_theta=noise.draw('eps_theta', (S_LOCAL, D + 1), 'cpu'), eps_f=noise.draw('eps_f', (S_LOCAL, F_, C, B), 'cpu'))
            return orc.loss(full, prev, xb, yb, nz)

        tr = ElboTrainer(None, beta=2.0, n_total=64, noise_seed=SEED, params=[leaves[k] for k in names], loss_fn=loss_fn,
                         optimizer=lambda ps: torch.optim.SGD(ps, lr=1e-3), comm=comm)
        for _ in range(steps):
            tr.step(x, y)
        out.put((rank, {k: leaves[k].detach().clone().numpy() for k in names}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rsag_exchange_is_bit_stable_over_50_steps():
    """Reduce-scatter + all-gather of the flat buffer, 50 SGD steps, two ranks: the ranks' parameters stay BIT-identical to each
    other (every rank applies the same summed buffer: no drift between replicas), a second run reproduces the first bit for
    bit, and with two ranks the result equals the one-all-reduce exchange bit for bit (a + b in either order)."""
    def run(comm):
        ctx = mp.get_context('spawn')
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_stable, args=(r, port, comm, 50, out)) for r in range(WORLD)]
        for p in procs:
            p.start()
        got = dict(out.get(timeout=240) for _ in range(WORLD))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return got
    a, b, c = run('rsag'), run('rsag'), run('allreduce')
    for k in a[0]:
        assert np.array_equal(a[0][k], a[1][k]), k            # rank 0 == rank 1
        assert np.array_equal(a[0][k], b[0][k]), k            # run 1 == run 2
        assert np.array_equal(a[0][k], c[0][k]), k            # rsag == allreduce (two ranks)
        assert np.isfinite(a[0][k]).all()


def test_split_samples_and_uneven_noise_shards():
    from vargp_amd import noise
    from vargp_amd.train import split_samples
    assert split_samples(64, 8) == [8] * 8 and split_samples(64, 3) == [22, 21, 21] and split_samples(3, 2) == [2, 1]
    try:
        counts = [3, 1, 2]
        parts = []
        for r in range(3):
            noise.set_shard(r, 3, 5, 'cpu', counts)
            parts.append(noise.draw('eps_f', (counts[r], 2, 3, 4), 'cpu'))
        gen = torch.Generator().manual_seed(5)
        assert torch.equal(torch.cat(parts), torch.randn(6, 2, 3, 4, generator=gen))
    finally:
        noise.clear_shard()


def test_noise_shards_tile_the_global_draw():
    from vargp_amd import noise
    try:
        parts = []
        for r in range(3):
            noise.set_shard(r, 3, 11, 'cpu')
            parts.append((noise.draw('eps_theta', (2, 7), 'cpu'), noise.draw('eps_f', (2, 3, 4, 5), 'cpu')))
        gen = torch.Generator().manual_seed(11)
        a, b = torch.randn(6, 7, generator=gen), torch.randn(6, 3, 4, 5, generator=gen)
        assert torch.equal(torch.cat([p[0] for p in parts]), a)
        assert torch.equal(torch.cat([p[1] for p in parts]), b)
    finally:
        noise.clear_shard()


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks under torch.distributed.run (a child
    process), brings the process group up (gloo here: no GPU), all-gathers the rank ids and exits 0 with one JSON line --
    the command form the driver uses for the scaling curve, up to (not including) the model build."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--comm', 'rsag'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['dry_run'] and line['ok'] and line['n_gpus'] == 2 and line['ranks_seen'] == [0, 1]
    assert line['comm'] == 'rsag' and line['workload'] == 'smnist'
    # a WORLD_SIZE that contradicts --gpus is an error message and a non-zero exit code, not an assertion trace
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'],
                       env=dict(env, RANK='0', WORLD_SIZE='1'), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'launches its own ranks' in r.stderr
