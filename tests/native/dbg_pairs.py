import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from test_hip_dist_pairs import _model, CASES, F_, B
from vargp_amd.fused import T0Program, TnProgram
from vargp_amd import ops, noise, _lib
from vargp_amd.train import split_pairs
ops.set_cholesky_error_mode('defer')
lib, ptr, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
for case in CASES:
    S, C, M, D, n_prev = CASES[case]
    gp, x, y = _model(S, C, M, D, n_prev)
    kern = gp.kernel
    eps_theta = torch.randn(S, D + 1, device='cuda:0')
    eps_f = torch.randn(S, F_, C, B, device='cuda:0')
    beta, scale = 2.0, 10.0
    with noise.inject(eps_theta=eps_theta, eps_f=eps_f):
        kl_h, kl_u, nll = gp.loss(x, y)
        (beta * kl_h + kl_u + scale * nll).backward()
    ref = dict(z=gp.z.grad.clone(), u_mean=gp.u_mean.grad.clone(), u_tril_vec=gp.u_tril_vec.grad.clone(),
               log_mean=kern.log_mean.grad.clone(), log_logvar=kern.log_logvar.grad.clone())
    print(case, 'ref scalars', kl_h.item(), kl_u.item(), nll.item())
    rects = split_pairs(S, C, 3)
    progs = []
    mu_f, var_f = torch.zeros(S, C, B, device='cuda:0'), torch.zeros(S, C, B, device='cuda:0')
    klu = 0.0
    for (s0, s1, c0, c1) in rects:
        Sl, Cl = s1 - s0, c1 - c0
        tn = bool(gp._use_block_program(B))
        shape = (Sl, Cl, M, D, B, F_) + ((n_prev + 1,) if tn else ())
        prog = (TnProgram if tn else T0Program)(*shape, x.device, False)
        packed = tuple(t[c0:c1] for t in gp._tn_operands()) if tn else ()
        scal = prog.forward(kern.log_mean.detach(), kern.log_logvar.detach(), kern.prior_log_mean, kern.prior_log_logvar,
                            gp.z.detach()[c0:c1], gp.u_mean.detach()[c0:c1], gp.u_tril_vec.detach()[c0:c1], *packed, x, y,
                            eps_theta[s0:s1].contiguous(), None, ext_lik=True)
        mu, var, gmu, gvar = prog.lik_buffers()
        mu_f[s0:s1, c0:c1] = mu; var_f[s0:s1, c0:c1] = var
        klu += scal[1].item() * Sl / S
        progs.append(prog)
    nll2 = torch.zeros((), device='cuda:0')
    seed = torch.tensor([scale], device='cuda:0')
    g = torch.zeros(2, S, C, B, device='cuda:0')
    _lib.check(lib.vargp_softmax_nll_fwd(ptr(mu_f), ptr(var_f), ptr(eps_f), ptr(y), ptr(nll2), S, F_, C, B, st), 'f')
    _lib.check(lib.vargp_softmax_nll_bwd(ptr(mu_f), ptr(var_f), ptr(eps_f), ptr(y), ptr(seed), ptr(g[0]), ptr(g[1]), S, F_, C, B, st), 'b')
    print(case, 'pair scalars kl_u', klu, 'nll', nll2.item())
    tot = {k: torch.zeros_like(v) for k, v in ref.items()}
    for prog, (s0, s1, c0, c1) in zip(progs, rects):
        Sl, Cl = s1 - s0, c1 - c0
        mu, var, gmu, gvar = prog.lik_buffers()
        gmu.copy_(g[0, s0:s1, c0:c1]); gvar.copy_(g[1, s0:s1, c0:c1])
        seeds = torch.tensor([beta * Sl * Cl / (S * C), Sl / S, 0.0], device='cuda:0')
        loc = {k: torch.zeros_like(v) for k, v in ref.items()}
        prog.backward(seeds, loc['log_mean'], loc['log_logvar'], loc['z'][c0:c1], loc['u_mean'][c0:c1], loc['u_tril_vec'][c0:c1])
        for k in tot:
            tot[k] += loc[k]
    torch.cuda.synchronize()
    for k in ref:
        print('   ', k, 'rel err', ((tot[k] - ref[k]).norm() / ref[k].norm()).item())
