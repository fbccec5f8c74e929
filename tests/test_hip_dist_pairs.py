"""GPU: the CLASS-sharded step (train.split_pairs: more ranks than hyper-samples) on the native HIP programs, three ranks
sharing cuda:0 with gloo as the transport (on a multi-GPU node the same code runs over RCCL): moments of each rank's
(sample, class) rectangle with `ext_lik` programs -> all-gather of mu, var -> the likelihood of all pairs -> each rank's
backward -> sum of the flat buffer; eager and as three hipGraphs around the two collectives.  Equals the single-process
step on all S x C pairs with the same noise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
F_, B = 4, 64
SEED = 41
# (S, C, M, D, n_prev): D <= 32 direct distance form; the MFMA + merged-factorisation route of the first-task program;
# a model with one earlier task (block program)
CASES = {'direct': (2, 4, 12, 16, 0), 'mfma': (2, 5, 56, 40, 0), 'block': (1, 4, 20, 40, 1)}


def _model(S, C, M, D, n_prev, dev='cuda:0'):
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.vargp import VARGP
    torch.manual_seed(0)
    xall, yall = mnist_like(4096, D, C, kind='gauss', seed=1)
    per = [xall[yall == c] for c in range(C)]
    prev = []
    for t in range(n_prev):
        eye = torch.zeros(M * (M + 1) // 2)
        idx = torch.arange(M)
        eye[idx * (idx + 1) // 2 + idx] = 1.0
        prev.append(dict(z=torch.stack([pc[(t + 1) * M:(t + 2) * M] for pc in per]).to(dev),
                         u_mean=(0.5 * torch.randn(C, M, 1)).to(dev), u_tril_vec=eye.repeat(C, 1).to(dev)))
    z = torch.stack([pc[:M] for pc in per])
    gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=F_), n_var_samples=S, prev_params=prev).to(dev)
    return gp, xall[:B].to(dev), yall[:B].to(dev)


B2 = 36          # the ragged last minibatch of an epoch: a second captured size


def _worker(rank, world, port, case, use_graph, steps, q, two_sizes=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vargp_amd import ops
        from vargp_amd.train import ElboTrainer, split_pairs
        ops.set_cholesky_error_mode('defer')
        S, C, M, D, n_prev = CASES[case]
        gp, x, y = _model(S, C, M, D, n_prev)
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED, shards=split_pairs(S, C, world))
        assert tr.class_split and tr._t0
        if use_graph:
            tr.capture(x, y, warmup=1)                 # (the warm-up is undone by capture itself)
            if two_sizes:
                tr.capture(x[:B2].contiguous(), y[:B2].contiguous(), warmup=1)
        outs = []
        for k in range(steps):
            xb, yb = (x[:B2].contiguous(), y[:B2].contiguous()) if (two_sizes and k % 2 == 1) else (x, y)
            out = tr.step_graph(xb, yb) if use_graph else tr.step(xb, yb)
            outs.append([o.item() for o in out])
        torch.cuda.synchronize()
        assert ops.linalg_error_count() == 0
        if rank == world - 1:
            q.put((outs, {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_class_sharded_two_captured_sizes():
    """Full minibatch and the ragged last one captured side by side, replayed alternately: the un-captured all-gather between
    the graphs must work on the pair buffers of the size being replayed (they are per size; a capture or an eager step of another
    size rebinds them)."""
    test_class_sharded_ranks_equal_single_process('mfma', True, two_sizes=True)


@pytest.mark.parametrize('use_graph', [False, True], ids=['eager', 'three_graphs'])
@pytest.mark.parametrize('case', list(CASES))
def test_class_sharded_ranks_equal_single_process(case, use_graph, two_sizes=False):
    world, steps = 3, (4 if two_sizes else 3)
    S, C, M, D, n_prev = CASES[case]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, use_graph, steps, q, two_sizes)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        outs2, sd2 = q.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)

    # single process, all pairs, the noise the ranks drew: every rank draws the WHOLE eps_theta (S, D+1) and eps_f (S, F, C, B)
    # of a step from a generator seeded alike
    from vargp_amd import noise, ops
    from vargp_amd.train import ElboTrainer
    ops.set_cholesky_error_mode('defer')
    try:
        gp, x, y = _model(S, C, M, D, n_prev)
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, native_noise=False)
        gen = torch.Generator(device='cuda:0')
        gen.manual_seed(SEED)
        outs1 = []
        for k in range(steps):
            Bk = B2 if (two_sizes and k % 2 == 1) else B
            nz = dict(eps_theta=torch.randn(S, D + 1, device='cuda:0', generator=gen),
                      eps_f=torch.randn(S, F_, C, Bk, device='cuda:0', generator=gen))
            with noise.inject(**nz):
                outs1.append([o.item() for o in tr.step(x[:Bk].contiguous(), y[:Bk].contiguous())])
        sd1 = {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}
    finally:
        noise.clear_shard()
        ops.set_cholesky_error_mode('raise')
    np.testing.assert_allclose(np.array(outs2), np.array(outs1), rtol=2e-4)
    for k in sd1:
        err = np.linalg.norm(sd2[k] - sd1[k]) / max(np.linalg.norm(sd1[k]), 1e-30)
        assert err < 1e-4, (k, err)
