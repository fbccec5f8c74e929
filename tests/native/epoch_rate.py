"""Where the driver's device-resident epoch loop loses time against the bare graph replay (GPU box only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vargp_amd  # noqa: E402
from vargp_amd import ops  # noqa: E402
from vargp_amd.train import ElboTrainer  # noqa: E402

ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model('cuda:0')
N = 6000
data = torch.randn(N, 784, device='cuda:0') * 0.02
targets = (torch.arange(N, device='cuda:0') % 10)
tr = ElboTrainer(gp, lr=3e-3, beta=10.0, n_total=N)
tr.capture(x, y)
tr.capture(x[:N % 512].contiguous(), y[:N % 512].contiguous())


def timed(name, fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-60s %8.1f steps/s  (%.3f ms/step)' % (name, steps / dt, 1e3 * dt / steps))


def a():
    for _ in range(360):
        tr.step_graph(x, y)


def a0():
    tr._select_capture(512)
    for _ in range(360):
        tr.step_graph()


def b():
    idx = torch.arange(512, device='cuda:0')
    for _ in range(360):
        tr.step_graph_gather(data, targets, idx)


def c(sync=True, errs=False, ragged=True):
    def run():
        for e in range(30):
            order = torch.randperm(N, device='cuda:0')
            for i in range(0, N if ragged else N - N % 512, 512):
                tr.step_graph_gather(data, targets, order[i:i + 512])
            if sync:
                torch.cuda.synchronize()
            if errs:
                vargp_amd.linalg_error_count()
    return run


def d(sync=True, errs=False, n=N):
    # the epoch-graph route of the driver: gather inside the graph, the epoch's 11 full batches in one launch, ragged tail per step
    dd, tt = data[:n].contiguous(), targets[:n].contiguous()
    tr._select_capture(512)
    tr.capture_epoch(dd, tt)

    def run():
        for e in range(30):
            order = torch.randperm(n, device='cuda:0')
            _, done = tr.run_epoch(order)
            if done * 512 < n:
                tr.step_graph_gather(dd, tt, order[done * 512:])
            if sync:
                torch.cuda.synchronize()
            if errs:
                vargp_amd.linalg_error_count()
    return run


timed('graph replay, no input copy', a0, 360)
timed('step_graph(x, y): copy of a resident batch', a, 360)
timed('step_graph_gather: index_select into the static inputs', b, 360)
timed('30 epochs of 11 full batches, no sync', c(False, False, False), 330)
timed('30 epochs of 11 full + 1 ragged batch, no sync', c(False, False, True), 360)
timed('... + one synchronize per epoch', c(True, False, True), 360)
timed('... + linalg_error_count per epoch', c(True, True, True), 360)
timed('epoch graphs (gather in the graph, 11 steps per launch): 30 epochs of 11 full + 1 ragged, no sync', d(False, False), 360)
timed('... + one synchronize per epoch', d(True, False), 360)
timed('... + linalg_error_count per epoch', d(True, True), 360)
print('info ring entries:', len(ops._info_ring))
