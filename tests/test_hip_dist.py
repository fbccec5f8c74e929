"""GPU: the sample-parallel trainer with the real HIP model, two ranks sharing cuda:0 (gloo moves the flat
gradient buffer; on a multi-GPU node the same code runs over RCCL): the 2-rank result — eager and with the
two-graph capture around the all-reduce — equals the single-process result for all 2*S_local samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
S_LOCAL, WORLD, F_, C, M, D, B = 2, 2, 4, 4, 12, 16, 64
SEED = 31


def _model(S, dev='cuda:0'):
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.vargp import VARGP
    torch.manual_seed(0)
    xall, yall = mnist_like(1024, D, C, kind='gauss', seed=1)
    z = torch.stack([xall[yall == c][:M] for c in range(C)])
    gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=F_), n_var_samples=S).to(dev)
    return gp, xall[:B].to(dev), yall[:B].to(dev)


def _run(gp, x, y, use_graph, steps=3):
    from vargp_amd import ops
    from vargp_amd.train import ElboTrainer
    tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED)
    if use_graph:
        import copy
        snap = copy.deepcopy(gp.state_dict())
        tr.capture(x, y, warmup=1)
        gp.load_state_dict(snap)                       # undo the warm-up step
        for grp in tr.optim.param_groups:
            grp['step'].zero_()
        for st in tr.optim.state.values():
            st['exp_avg'].fill_(1e-6)
            st['exp_avg_sq'].fill_(1e-6)
        tr._rng_counter.zero_()                        # restart the shared noise stream as well
    outs = []
    for _ in range(steps):
        out = tr.step_graph(x, y) if use_graph else tr.step(x, y)
        outs.append([o.item() for o in out])
    torch.cuda.synchronize()
    return outs, {k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}


def _worker(rank, port, use_graph, q, rccl=False):
    """rccl: one GPU per rank, the exchange over RCCL (backend 'nccl'); otherwise both ranks share cuda:0 and gloo carries it."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dev = rank if rccl else 0
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group('nccl', rank=rank, world_size=WORLD, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from vargp_amd import ops
        ops.set_cholesky_error_mode('defer')
        gp, x, y = _model(S_LOCAL, f'cuda:{dev}')
        outs, sd = _run(gp, x, y, use_graph)
        if rank == 0:
            q.put((outs, {k: v.numpy() for k, v in sd.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('use_graph,rccl', [(False, False), (True, False),
                                            pytest.param(False, True, id='rccl-2gpu-eager'),
                                            pytest.param(True, True, id='rccl-2gpu-graph')])
def test_two_ranks_equal_single_process(use_graph, rccl):
    if rccl and torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (the RCCL exchange between two devices; the one-GPU box runs the gloo variants)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, use_graph, q, rccl)) for r in range(WORLD)]
    for p in procs:
        p.start()
    outs2, sd2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    # single process, all samples, same global noise stream (the program's counter-based generator: a rank's noise
    # is the slice [rank*S, (rank+1)*S) of the draw this process makes)
    from vargp_amd import noise, ops
    ops.set_cholesky_error_mode('defer')
    try:
        gp, x, y = _model(S_LOCAL * WORLD)
        from vargp_amd.train import ElboTrainer
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED)
        outs1 = [[o.item() for o in tr.step(x, y)] for _ in range(3)]
        sd1 = {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}
    finally:
        noise.clear_shard()
        ops.set_cholesky_error_mode('raise')
    np.testing.assert_allclose(np.array(outs2), np.array(outs1), rtol=2e-4)
    for k in sd1:
        err = np.linalg.norm(sd2[k] - sd1[k]) / max(np.linalg.norm(sd1[k]), 1e-30)
        assert err < 1e-4, (k, err)


def _worker_uneven(rank, port, counts, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(len(counts)))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=len(counts))
    try:
        from vargp_amd.train import ElboTrainer
        gp, x, y = _model(counts[rank])
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED, sample_counts=counts)
        outs = [[o.item() for o in tr.step(x, y)] for _ in range(2)]
        torch.cuda.synchronize()
        if rank == 0:
            q.put((outs, {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_uneven_sample_split_equals_single_process():
    """5 hyper-samples over 2 ranks (3 + 2): weights S_r / S, noise = slices of one global draw."""
    counts = [3, 2]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs2, sd2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from vargp_amd import noise
    from vargp_amd.train import ElboTrainer
    try:
        gp, x, y = _model(sum(counts))
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED)
        outs1 = [[o.item() for o in tr.step(x, y)] for _ in range(2)]
        sd1 = {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}
    finally:
        noise.clear_shard()
    np.testing.assert_allclose(np.array(outs2), np.array(outs1), rtol=2e-4)
    for k in sd1:
        err = np.linalg.norm(sd2[k] - sd1[k]) / max(np.linalg.norm(sd1[k]), 1e-30)
        assert err < 1e-4, (k, err)


def _worker_rccl(port, mode, q):
    """One rank on RCCL ('nccl' backend), the multi-rank path forced (force_exchange): the flat buffer goes through a real
    RCCL all-reduce -- eager, or between the two step graphs."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from vargp_amd import ops
        from vargp_amd.train import ElboTrainer
        ops.set_cholesky_error_mode('defer')
        gp, x, y = _model(S_LOCAL)
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED, sample_counts=[S_LOCAL], force_exchange=True)
        assert tr.multi
        if mode != 'eager':
            tr.capture(x, y, warmup=1)                 # (the warm-up is undone by capture itself)
        outs = []
        for _ in range(3):
            out = tr.step_graph(x, y) if mode != 'eager' else tr.step(x, y)
            outs.append([o.item() for o in out])
        torch.cuda.synchronize()
        q.put((outs, {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['eager', 'two_graphs'])
def test_rccl_single_rank_exchange_equals_plain_step(mode):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(port, mode, q))
    p.start()
    try:
        outs2, sd2 = q.get(timeout=240)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    from vargp_amd import noise, ops
    ops.set_cholesky_error_mode('defer')
    try:
        gp, x, y = _model(S_LOCAL)
        from vargp_amd.train import ElboTrainer
        tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=10 * B, noise_seed=SEED)
        outs1 = [[o.item() for o in tr.step(x, y)] for _ in range(3)]
        sd1 = {k: v.detach().cpu().numpy() for k, v in gp.state_dict().items()}
    finally:
        noise.clear_shard()
        ops.set_cholesky_error_mode('raise')
    np.testing.assert_allclose(np.array(outs2), np.array(outs1), rtol=2e-4)
    for k in sd1:
        err = np.linalg.norm(sd2[k] - sd1[k]) / max(np.linalg.norm(sd1[k]), 1e-30)
        assert err < 1e-4, (k, err)
