"""GPU: the program's native noise (Philox4x32-10 + Box-Muller inside vargp_elbo_t0_fwd) against a numpy restatement of
the published Philox algorithm (Salmon et al., SC'11), its statistics, the step counter, and the property the
sample-parallel path relies on: a rank's noise is a slice of one global draw."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
S, C, M, D, B, F_ = 3, 4, 10, 36, 50, 5
SEED = 0x1234_5678_9ABC_DEF1


def _philox4x32_10(counter, key):
    """counter (n, 4) uint32, key (2,) uint32 -> (n, 4) uint32."""
    c = counter.astype(np.uint64).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m0, m1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = m0 * c[:, 0], m1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & mask
        c = np.stack([n0, p1 & mask, n2, p0 & mask], axis=1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def _normal_ref(seed, stream, g0, n, step):
    g = np.arange(g0, g0 + n, dtype=np.int64)
    grp = (g >> 2).astype(np.uint64)
    ctr = np.stack([grp & np.uint64(0xFFFFFFFF), grp >> np.uint64(32), np.full(n, stream, np.uint64),
                    np.full(n, step, np.uint64)], axis=1)
    w = _philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    h = ((g & 3) >> 1).astype(np.int64)
    a = np.take_along_axis(w, (2 * h)[:, None], 1)[:, 0].astype(np.float32)
    b = np.take_along_axis(w, (2 * h + 1)[:, None], 1)[:, 0].astype(np.float32)
    u0 = (a + np.float32(1)) * np.float32(2.0 ** -32)
    u1 = b * np.float32(2.0 ** -32)
    r = np.sqrt(-2.0 * np.log(u0.astype(np.float64)))
    ang = 2.0 * np.pi * u1.astype(np.float64)
    return np.where((g & 1) == 0, r * np.cos(ang), r * np.sin(ang))


def _program(s, offset, counter, seed=SEED, f=F_, b=B):
    from vargp_amd.fused import T0Program
    prog = T0Program(s, C, M, D, b, f, DEV)
    prog.set_rng(seed, counter, offset)
    return prog


def _inputs(b=B):
    from oracle import vargp_oracle as orc
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, b, n_prev=0, seed=3, kind='gauss')
    t = lambda v: v.to(DEV).contiguous()
    return [t(params[k]) for k in ('log_mean', 'log_logvar', 'prior_log_mean', 'prior_log_logvar', 'z', 'u_mean',
                                   'u_tril_vec')] + [t(x), t(y)]


def test_native_noise_matches_numpy_philox_and_counter_advances():
    counter = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    prog = _program(S, 0, counter)
    args = _inputs()
    prog.forward(*args, None, None)
    torch.cuda.synchronize()
    assert int(counter.item()) == 8
    et, ef = prog.eps_theta().cpu().numpy().ravel(), prog.eps_f().cpu().numpy().ravel()
    np.testing.assert_allclose(et, _normal_ref(SEED, 0, 0, et.size, 7), atol=2e-5)
    np.testing.assert_allclose(ef, _normal_ref(SEED, 1, 0, ef.size, 7), atol=2e-5)
    prog.forward(*args, None, None)                       # next step: different noise
    ef2 = prog.eps_f().cpu().numpy().ravel()
    assert np.abs(ef2 - ef).max() > 1.0
    np.testing.assert_allclose(ef2, _normal_ref(SEED, 1, 0, ef.size, 8), atol=2e-5)
    counter.fill_(7)                                      # rewinding the counter reproduces the draw
    prog.forward(*args, None, None)
    np.testing.assert_array_equal(prog.eps_f().cpu().numpy().ravel(), ef)


def test_rank_noise_is_a_slice_of_the_global_draw():
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    args = _inputs()
    glob = _program(S, 0, counter)
    glob.forward(*args, None, None)
    gt, gf = glob.eps_theta().clone(), glob.eps_f().clone()
    for r in range(S):                                    # S "ranks" of one sample each (odd sizes: unaligned offsets)
        counter.zero_()
        loc = _program(1, r, counter)
        loc.forward(*args, None, None)
        assert torch.equal(loc.eps_theta()[0], gt[r]) and torch.equal(loc.eps_f()[0], gf[r])


def test_native_noise_statistics():
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    prog = _program(S, 0, counter, f=40, b=500)           # 240k likelihood-noise values
    prog.forward(*_inputs(b=500), None, None)
    e = prog.eps_f().double().flatten()
    n = e.numel()
    assert abs(e.mean().item()) < 5 / n ** 0.5
    assert abs(e.var().item() - 1) < 5 * (2 / n) ** 0.5
    assert abs((e ** 4).mean().item() - 3) < 0.1
    assert abs((e[1:] * e[:-1]).mean().item()) < 5 / n ** 0.5      # neighbours uncorrelated
    assert e.abs().max().item() < 6.7


def test_trainer_graph_equals_eager_with_native_noise():
    """Same seed and counter -> the captured step reproduces the eager step's noise, hence its parameters."""
    import copy
    from vargp_amd import ops
    from vargp_amd.train import ElboTrainer
    from gpu_common import build_gp
    from oracle import vargp_oracle as orc
    params, prev, x, y, nz = orc.make_problem(2, 3, 4, 16, 40, 64, n_prev=0, seed=9, kind='mnist')
    xd, yd = x.to(DEV), y.to(DEV)
    ops.set_cholesky_error_mode('defer')
    try:
        res = []
        for mode in ('eager', 'graph'):
            gp = build_gp(params, prev, 2, 3)
            tr = ElboTrainer(gp, lr=1e-3, beta=2.0, n_total=640, noise_seed=5)
            assert tr.native_noise
            if mode == 'graph':
                snap = copy.deepcopy(gp.state_dict())
                tr.capture(xd, yd, warmup=2)
                gp.load_state_dict(snap)
                tr._rng_counter.zero_()
                for grp in tr.optim.param_groups:
                    grp['step'].zero_()
                for st in tr.optim.state.values():
                    st['exp_avg'].fill_(1e-6)
                    st['exp_avg_sq'].fill_(1e-6)
            outs = [[float(v) for v in (tr.step_graph() if mode == 'graph' else tr.step(xd, yd))] for _ in range(3)]
            torch.cuda.synchronize()
            res.append((outs, {k: v.detach().cpu().clone() for k, v in gp.state_dict().items()}))
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
        for k in res[0][1]:
            assert torch.allclose(res[0][1][k], res[1][1][k], rtol=1e-5, atol=1e-7), k
    finally:
        ops.set_cholesky_error_mode('raise')
