"""One rank's compute share of the CLASS-sharded BASELINE config 2 step (train.split_pairs(3, 10, 8): rectangles of 1 sample x
3..5 classes), timed on one GPU without the two collectives: program forward (ext_lik) -> likelihood of all 30 x 512 moments ->
program backward -> Yogi, replayed from one hipGraph.  Feeds the projection table of DESIGN.md section 8.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from vargp_amd import _lib, ops  # noqa: E402
from vargp_amd.fused import T0Program  # noqa: E402
from vargp_amd.optim import Yogi  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model(dev)
S, C, M, D, B, F_ = 3, 10, 100, 784, 512, 10
kern = gp.kernel
lib, ptr, st = _lib.lib(), _lib.ptr, _lib.stream_ptr
for (s0, s1, c0, c1) in [(0, 1, 0, 3), (0, 1, 0, 4), (0, 1, 0, 5), (0, 1, 0, 10), (0, 2, 0, 10), (0, 3, 0, 10)]:
    Sl, Cl = s1 - s0, c1 - c0
    prog = T0Program(Sl, Cl, M, D, B, F_, dev)
    params = [kern.log_mean, kern.log_logvar, gp.z, gp.u_mean, gp.u_tril_vec]
    grads = [torch.zeros_like(p) for p in params]
    for p, g in zip(params, grads):
        p.grad = g
    opt = Yogi(params, lr=1e-6)
    eps_theta = torch.randn(S, D + 1, device=dev)
    eps_f = torch.randn(S, F_, C, B, device=dev)
    mom = torch.zeros(2, S, C, B, device=dev)
    gfull = torch.zeros(2, S, C, B, device=dev)
    nll = torch.zeros((), device=dev)
    seed = torch.tensor([float(bench.N_TOTAL) / B], device=dev)
    seeds = torch.tensor([bench.BETA * Sl * Cl / (S * C), Sl / S, 0.0], device=dev)
    th = eps_theta[s0:s1].contiguous()

    def step():
        prog.forward(kern.log_mean.detach(), kern.log_logvar.detach(), kern.prior_log_mean, kern.prior_log_logvar,
                     gp.z.detach()[c0:c1], gp.u_mean.detach()[c0:c1], gp.u_tril_vec.detach()[c0:c1], x, y, th, None, ext_lik=True)
        mu, var, gmu, gvar = prog.lik_buffers()
        mom[0, s0:s1, c0:c1].copy_(mu)
        mom[1, s0:s1, c0:c1].copy_(var)
        _lib.check(lib.vargp_softmax_nll_fwd(ptr(mom[0]), ptr(mom[1]), ptr(eps_f), ptr(y), ptr(nll), S, F_, C, B, st()), 'f')
        _lib.check(lib.vargp_softmax_nll_bwd(ptr(mom[0]), ptr(mom[1]), ptr(eps_f), ptr(y), ptr(seed), ptr(gfull[0]), ptr(gfull[1]),
                                             S, F_, C, B, st()), 'b')
        gmu.copy_(gfull[0, s0:s1, c0:c1])
        gvar.copy_(gfull[1, s0:s1, c0:c1])
        prog.backward(seeds, grads[0], grads[1], grads[2][c0:c1], grads[3][c0:c1], grads[4][c0:c1])
        opt.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print('rectangle %d sample(s) x %2d classes (%2d of 30 problems): %.1f us per step without the collectives'
          % (Sl, Cl, Sl * Cl, e0.elapsed_time(e1) / 200 * 1e3))
    # the time line of the last step of a launch (vargp_prof_spans): how long each launch takes when it has almost nothing to do
    acc = {}
    for _ in range(10):
        _lib.prof_spans(1)
        g.replay()
        torch.cuda.synchronize()
        for k, (a, b) in _lib.prof_spans(0).items():
            if ':' not in k:
                acc.setdefault(k, [0.0, 0.0])
                acc[k][0] += a
                acc[k][1] += b - a
    _lib.prof_spans(2)
    print('    spans (us): ' + '  '.join('%s %.1f' % (k, v[1] / 10) for k, v in sorted(acc.items(), key=lambda kv: kv[1][0])))
