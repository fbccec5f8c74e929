"""Diagnostic (not a test): where does the fp32 noise of the ill-conditioned toy t>0 cases come from?
Compares HIP intermediates with the fp64 oracle and with the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from helpers import load_case, rel_l2, to_dev
from oracle import vargp_oracle as orc
from gpu_common import build_gp, DEV
from vargp_amd import noise, gp_utils

for name in ['toy_t1', 'toy_t2', 'smnist_small_t1', 'smnist_full_t0']:
    g, params, prev, x, y, nz = load_case(name)
    d = lambda o: {k: v.double() for k, v in o.items()}
    p64, prev64, nz64 = d(params), [d(p) for p in prev], d(nz)
    S, F_ = int(g['meta'][0]), int(g['meta'][1])
    gp = build_gp(params, prev, S, F_)
    th32 = orc.sample_hypers(params['log_mean'], params['log_logvar'], nz['eps_theta'])
    th64 = orc.sample_hypers(p64['log_mean'], p64['log_logvar'], nz64['eps_theta'])
    with torch.no_grad(), noise.inject(**to_dev(nz, DEV)):
        th = gp.kernel.sample_hypers(S)
        if prev:
            cq = {}
            mu_lt, S_lt, mu_leq, S_leq, z_leq = gp.compute_q(th, cache=cq)
            o32 = orc.compute_q(th32, params, prev)
            o64 = orc.compute_q(th64, p64, prev64)
            for nm, mine, i in [('mu_leq', mu_leq, 2), ('S_leq', S_leq, 3), ('Lz_lt', cq['Lz_lt'], 5), ('LzKzx', cq['Lz_lt_Kz_lt_z_t'], 6)]:
                print(name, nm, 'hip-vs-64 %.2e' % rel_l2(mine.cpu(), o64[i]), ' orc32-vs-64 %.2e' % rel_l2(o32[i], o64[i]))
            Kzz = gp.kernel.compute(th, z_leq)
            K32, K64 = orc.rbf_gram(th32, o32[4]), orc.rbf_gram(th64, o64[4])
            print(name, 'Kzz_leq max-abs hip-vs-64 %.2e  orc32-vs-64 %.2e' % ((Kzz.cpu().double() - K64).abs().max(), (K32.double() - K64).abs().max()))
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
    sc64 = orc.loss(p64, prev64, x.double(), y, nz64)
    sc32 = orc.loss(params, prev, x, y, nz)
    for i, k in enumerate(['kl_hypers', 'kl_u', 'nll']):
        mine = [kl_h, kl_u, nll][i].item()
        print(name, k, 'hip %.6f ref32 %.6f orc32 %.6f fp64 %.6f | hip-vs-64 %.2e ref32-vs-64 %.2e hip-vs-ref32 %.2e' % (
            mine, float(g[k]), sc32[i].item(), sc64[i].item(), abs(mine - sc64[i].item()) / abs(sc64[i].item()),
            abs(float(g[k]) - sc64[i].item()) / abs(sc64[i].item()), abs(mine - float(g[k])) / abs(float(g[k]))))
