#!/usr/bin/env python
"""ELBO steps/s of the VAR-GP hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
CHILD process (before any GPU call; never an exec), forwards rank 0's one JSON line and exits with the child's code.
`--dry-run` stops after the process group is up (ranks all-gathered, no model, no oracle): rc 0 = the launch path works.
`--comm allreduce|rsag` selects how the flat [grads | kl_u | nll] buffer is summed over the ranks.

N=1 (default line): BASELINE config 2 (BASELINE.json configs[1]) — Split-MNIST task 0 shape, S=3 hyper-samples, F=10, C=10,
M=100, D=784, B=512, synthetic data resident in HBM.  One step = zero_grad + VARGP.loss + combine + backward
+ (N>1: one RCCL all-reduce) + Yogi step, exactly as experiments/vargp.py:29-37 does it.  The same line carries a
`secondary` object: short runs of the other BASELINE configs (Cfg3 Permuted-MNIST tasks 0 / 1 / 4 / 9, Split-MNIST task 1,
Cfg4's 64 samples on one GPU = the base of the multi-GPU curve, Cfg5 stress sweep), each with ms_per_step, roofline.frac
and elbo_rtol_vs_cpu (`--no-secondary` skips them).
N>1 (default, `scaling: "weak"`): every rank evaluates the metric's own config-2 ELBO + gradient on ITS OWN 3 hyper-samples
(same minibatch, rank-sliced global noise), ONE exchange sums the flat [grads | kl_u | nll] buffer, every rank takes the same Yogi
step: an optimizer step over 3N hyper-samples.  The unit of the N=1 line is one ELBO + gradient over 3 hyper-samples x 10 classes;
an N-rank step processes N of them, so `value` = N x optimizer steps / s (`optimizer_steps_per_s` is in the line as well), directly
comparable with the N=1 line.  The same line's `secondary` carries BASELINE config 4 as north_star states it -- a FIXED total of 64
hyper-samples x 10 classes split over the N ranks (`smnist_s64`, STRONG scaling, uneven shards if 64 % N != 0; value = global
steps / s, base = `secondary.smnist_s64` of the N=1 line) -- and config 2's 30 (sample, class) problems class-sharded over the
ranks (`smnist_pairs`).  `--scaling strong` makes config 4 the headline of the N>1 line instead.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

S, F_, C, M, D, B = 3, 10, 10, 100, 784, 512
N_TOTAL, BETA, LR = 12000, 10.0, 3e-3
N_PREV = 0
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_CLOCK_GHZ = 2.4              # the clock that peak is quoted at (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz)
PREHEAT_S = 0.3                   # plain matrix products on scratch tensors before the warm-up steps (reported as `preheat_s`)
DOMINANT_TAG = 'rbf_kuf'       # the K_uf distance GEMM incl. its split-K combine pass (vargp_rbf_gram_fwd)

# Secondary workloads (not the driver's default line): other BASELINE configs, same step definition.
WORKLOADS = {
    'smnist': dict(S=3, M=100, n_prev=0, desc='BASELINE config 2 (Split-MNIST t=0): S3 F10 C10 M100 D784 B512 ELBO step'),
    'smnist_s64': dict(S=64, M=100, n_prev=0, strong=True, desc='BASELINE config 4 (Split-MNIST t=0, 64 hyper-samples x 10 classes), samples split over the ranks'),
    'smnist_pairs': dict(S=3, M=100, n_prev=0, pairs=True, desc='BASELINE config 2 (Split-MNIST t=0, S3 C10 M100 D784 B512): its 30 (hyper-sample, class) problems split over the ranks (train.split_pairs: class-sharded, all-gather of the predictive moments before the softmax)'),
    'smnist_s8': dict(S=8, M=100, n_prev=0, desc="one rank's share of BASELINE config 4 on 8 GPUs (Split-MNIST t=0, 8 of the 64 hyper-samples), no exchange"),
    'smnist_s16': dict(S=16, M=100, n_prev=0, desc="one rank's share of BASELINE config 4 on 4 GPUs (16 of the 64 hyper-samples), no exchange"),
    'smnist_s32': dict(S=32, M=100, n_prev=0, desc="one rank's share of BASELINE config 4 on 2 GPUs (32 of the 64 hyper-samples), no exchange"),
    'smnist_t1': dict(S=3, M=100, n_prev=1, desc='Split-MNIST task 1 (Mt=200), M=100, S=3 (native block-structured program)'),
    'smnist_t4': dict(S=3, M=100, n_prev=4, desc='Split-MNIST task 4 (Mt=500), M=100, S=3 (native block-structured program)'),
    'pmnist_t0': dict(S=10, M=200, n_prev=0, desc='BASELINE config 3 (Permuted-MNIST), task 0: M=200, S=10'),
    'pmnist_t1': dict(S=10, M=200, n_prev=1, desc='Permuted-MNIST task 1 (Mt=400), M=200, S=10'),
    'pmnist_t4': dict(S=10, M=200, n_prev=4, desc='Permuted-MNIST task 4 (Mt=1000), M=200, S=10'),
    'pmnist_t9': dict(S=10, M=200, n_prev=9, desc='Permuted-MNIST task 9, the last of the 10-task sequence (Mt=2000), M=200, S=10'),
}


def make_model(device, seed=0):
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.synthetic import mnist_like
    from vargp_amd.vargp import VARGP
    torch.manual_seed(seed)
    xall, yall = mnist_like(max(4096, 16 * M * (N_PREV + 1)), D, C, kind='gauss', seed=1)
    per_class = [xall[yall == c] for c in range(C)]
    prev = []
    for t in range(N_PREV):       # frozen earlier tasks with the initial variational parameters
        zt = torch.stack([pc[(t + 1) * M:(t + 2) * M] for pc in per_class])
        eye = torch.zeros(M * (M + 1) // 2)
        idx = torch.arange(M)
        eye[idx * (idx + 1) // 2 + idx] = 1.0
        prev.append(dict(z=zt.to(device), u_mean=(0.5 * torch.randn(C, M, 1)).to(device),
                         u_tril_vec=eye.repeat(C, 1).to(device)))
    z = torch.stack([pc[:M] for pc in per_class])
    gp = VARGP(z, RBFKernel(D), MulticlassSoftmax(n_f=F_), n_var_samples=S, prev_params=prev).to(device)
    return gp, xall[:B].to(device), yall[:B].to(device)


def stress(args, device, cpu=True, n=None):
    """BASELINE config 5: N=1e6, D=784, M=2048, C=10, S=1.  One ELBO evaluation WITH its gradient over all N points,
    K_uf built tile by tile in HBM (VARGP.elbo_tiled -> vargp_elbo_tn_begin/_tile/_end: kernel matrix of the inducing
    points and its n=2048 factorisation once, forward + partial backward per tile), then the forward-only predictive sweep.
    `value` = data points per second of the ELBO+gradient sweep.  -> the result dict."""
    from vargp_amd import _lib, ops
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.vargp import VARGP
    n, m, tile = n or args.stress_n, 2048, 8192
    torch.manual_seed(0)
    x = torch.randn(n, D, device=device) * (0.25 / D) ** 0.5
    y = (torch.arange(n, device=device) % C).to(torch.int64)
    z = torch.stack([x[c * m:(c + 1) * m] + 0.01 * torch.randn(m, D, device=device) for c in range(C)])
    gp = VARGP(z.cpu(), RBFKernel(D), MulticlassSoftmax(n_f=F_), n_var_samples=1).to(device)
    ops.set_cholesky_error_mode('defer')
    rtol, rtol_on = stress_elbo_check(gp, x, y)
    # ---- ELBO + gradient sweep ------------------------------------------------------------------------------------
    gp.elbo_tiled(x[:2 * tile], y[:2 * tile], tile, beta=BETA)                      # warm-up (two tiles)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    _lib.prof_read('')
    t0 = time.perf_counter()
    out = gp.elbo_tiled(x, y, tile, beta=BETA)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    kern_ms, kern_n = _lib.prof_read('rbf_kuf_gemm')
    finite = all(torch.isfinite(v).item() for v in out) and all(bool(torch.isfinite(p.grad).all()) for p in gp.parameters())
    flops = 2.0 * C * m * tile * D
    avg_s = kern_ms / max(kern_n, 1) * 1e-3
    # ---- the n = 2048 factorisation (L and T = L^-1 of K_uu + eps I, 10 matrices), timed on its own ------------------
    with torch.no_grad():
        K = gp.kernel.compute(gp.kernel.log_mean.detach().unsqueeze(0), gp.z.detach())
        ops.chol_inv(K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.chol_inv(K)
        e1.record()
        torch.cuda.synchronize()
        chol_ms = e0.elapsed_time(e1) / 3
    chol_flops = C * (m ** 3 / 3.0 + 2.0 * m ** 3 / 3.0)        # factor + explicit inverse of a triangular factor
    # ---- forward-only predictive sweep ---------------------------------------------------------------------------------
    with torch.no_grad():
        gp.predict(x[:2 * tile], tile=tile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        probs = gp.predict(x, tile=tile)
        torch.cuda.synchronize()
        dt_pred = time.perf_counter() - t0
    res = dict(metric='ELBO+gradient sweep, data points/sec (stress)', value=n / dt, unit='points/s', n_gpus=1, steps=1,
               warmup=1, ms_per_step=1e3 * dt, higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32',
               data='synthetic', finite=bool(finite and torch.isfinite(probs).all().item()),
               cholesky_failures=ops.linalg_error_count(),
               config=dict(workload='BASELINE config 5: ELBO + gradient over N=%d points, D=784 M=2048 C=10 S=1, K_uf tiled in '
                                    'HBM (tile %d)' % (n, tile)),
               elbo_rtol_vs_cpu=rtol, elbo_rtol_checked_on=rtol_on,
               elbo=dict(kl_hypers=out[0].item(), kl_u=out[1].item(), nll=out[2].item()),
               predictive_sweep=dict(points_per_s=n / dt_pred, seconds=dt_pred),
               roofline=dict(bound='mfma', kernel='gemm_kernel<RBF> K_uf tile [10*2048 x 784] x [784 x 8192]',
                             achieved=flops / avg_s / 1e12 if kern_n else None, peak=MFMA_F32_PEAK_TFLOPS,
                             unit='TFLOP/s', frac=flops / avg_s / 1e12 / MFMA_F32_PEAK_TFLOPS if kern_n else None,
                             launches=kern_n, avg_us=avg_s * 1e6, traffic=measured_traffic('stress_kuf_tile'),
                             mfma_util=measured_mfma_util('stress_kuf_tile'),
                             counters_from=_latest_profile('traffic')[1] if measured_traffic('stress_kuf_tile') else None),
               roofline_chol=dict(bound='mfma', kernel='blocked Cholesky + inverse factor, n=2048, 10 matrices (register '
                                  'diagonal blocks + MFMA panel / trailing / inverse GEMMs)', achieved=chol_flops / (chol_ms * 1e-3) / 1e12,
                                  peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                                  frac=chol_flops / (chol_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, avg_us=chol_ms * 1e3,
                                  traffic=None))
    if cpu:
        res['cpu_baseline'] = stress_cpu_baseline(gp)
    gp.release_programs()
    return res


def stress_elbo_check(gp, x, y, n_chk=1024, tile=512):
    """ELBO terms of the tiled HIP sweep (two tiles of the same M=2048 C=10 S=1 model, injected noise) against the fp32 CPU
    oracle on the same points: max relative error over (kl_hypers, kl_u, nll summed over the points, total)."""
    from oracle import vargp_oracle as orc
    from vargp_amd import noise
    xs, ys = x[:n_chk].contiguous(), y[:n_chk].contiguous()
    nz = dict(eps_theta=torch.randn(1, D + 1), eps_f=torch.randn(1, F_, C, n_chk))
    with noise.inject(**{k: v.to(x.device) for k, v in nz.items()}):
        got = [float(v) for v in gp.elbo_tiled(xs, ys, tile, beta=BETA)]
    for p in gp.parameters():
        p.grad = None
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    with torch.no_grad():
        ref = [float(v) for v in orc.loss(snapshot(gp), [], xs.cpu(), ys.cpu(), nz)]
    tot = lambda t: BETA * t[0] + t[1] + t[2]
    errs = [abs(a - b) / abs(b) for a, b in zip(got + [tot(got)], ref + [tot(ref)]) if b != 0.0]
    return max(errs), 'sub-sample: the first %d points in tiles of %d (same M=2048 C=10 S=1 model)' % (n_chk, tile)


def stress_cpu_baseline(gp, n_cpu=2048):
    """The oracle (CPU port of the reference algorithm) on a reduced N of the same workload: one loss + backward over
    n_cpu points at M=2048, C=10, S=1; reported as points per second."""
    from oracle import vargp_oracle as orc
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    p = snapshot(gp)
    xs = torch.randn(n_cpu, D) * (0.25 / D) ** 0.5
    ys = (torch.arange(n_cpu) % C).to(torch.int64)
    nz = dict(eps_theta=torch.randn(1, D + 1), eps_f=torch.randn(1, F_, C, n_cpu))
    t0 = time.perf_counter()
    orc.elbo_step(p, [], xs, ys, nz, beta=BETA, n_total=n_cpu)
    dt = time.perf_counter() - t0
    return dict(value=n_cpu / dt, unit='points/s', cores=threads, kind='port',
                sample=f'one ELBO + gradient over {n_cpu} points (M=2048 C=10 S=1 D=784), the factorisations included')


def survey_flops_fwd(S_, C_, M_, n_prev, B_, D_, F__):
    """Algorithmic flop of one forward (symmetric work counted once); a step is 3x this."""
    Mt = M_ * (n_prev + 1)
    n_S = 1 if n_prev == 0 else S_
    fl = (2.0 * S_ * C_ * Mt * B_ * D_ + 1.0 * S_ * C_ * Mt * Mt * D_ + S_ * C_ * Mt ** 3 / 3.0 + n_S * C_ * Mt ** 3 / 3.0
          + 1.0 * S_ * C_ * Mt * Mt * B_ + S_ * C_ * Mt ** 3 / 3.0 + 1.0 * S_ * C_ * Mt * Mt * B_ + 6.0 * S_ * C_ * Mt * B_
          + S_ * C_ * M_ ** 3 / 3.0 + 12.0 * S_ * F__ * C_ * B_)
    for i in range(1, n_prev + 1):          # the linear_joint chain, one fold per earlier task (M< = i M)
        Ml = i * M_
        fl += S_ * C_ * (4.0 / 3.0 * Ml ** 3 + 4.0 * Ml * Ml * M_ + 2.0 * Ml * M_ * M_)
    return fl


def step_timeline(run, period_us, reps=20, burst=4):
    """The step as the GPU sees it under graph replay (vargp_prof_spans: wall-clock stamps written by the kernels themselves,
    the only clock that reaches inside a replayed hipGraph): `burst` replays back to back, the stamps of the LAST one read
    back; averaged over `reps` bursts.  -> list of dict(kernel, start_us (from the step's first kernel), span_us (first
    workgroup's start .. last workgroup's end), gap_before_us (previous kernel's end .. this start; for the first kernel:
    the step period minus the rest), slot_us = gap + span (sums to the step period)), in launch order."""
    from vargp_amd import _lib
    acc, n, clk = {}, 0, {}
    for _ in range(reps):
        _lib.prof_spans(1)
        for _ in range(burst):
            run()
        torch.cuda.synchronize()
        t = _lib.prof_spans(0)
        for k, ghz in _lib.prof_clocks().items():      # shader clock held under each kernel (its workgroup 0: s_memtime over wall clock)
            c = clk.setdefault(k, [0.0, 0])
            c[0] += ghz
            c[1] += 1
        main = sorted(((k, v) for k, v in t.items() if ':' not in k), key=lambda kv: kv[1][0])
        if len(main) < 2:
            continue
        t0 = main[0][1][0]
        prev_end = None
        for k, (a, b) in main:
            e = acc.setdefault(k, [0.0, 0.0, 0.0])
            e[0] += a - t0
            e[1] += b - a
            e[2] += (a - prev_end) if prev_end is not None else 0.0
            prev_end = b
        for k, (a, b) in t.items():
            if ':' in k:
                acc.setdefault(k, [0.0, 0.0, 0.0])[1] += b - t[k.split(':')[0]][0]
                if a:        # (a role that also stamps its LAST workgroup's start)
                    acc[k][0] += a - t[k.split(':')[0]][0]
        n += 1
    _lib.prof_spans(2)
    if not n:
        return []
    rows = [dict(kernel=k, start_us=v[0] / n, span_us=v[1] / n, gap_before_us=v[2] / n) for k, v in acc.items() if ':' not in k]
    rows.sort(key=lambda r: r['start_us'])
    if rows:
        rows[0]['gap_before_us'] = max(0.0, period_us - sum(r['span_us'] + r['gap_before_us'] for r in rows))
    for r in rows:
        r['slot_us'] = r['span_us'] + r['gap_before_us']
        if r['kernel'] in clk and clk[r['kernel']][1]:
            r['clock_ghz'] = clk[r['kernel']][0] / clk[r['kernel']][1]
    marks = {k: v for k, v in acc.items() if ':' in k}
    for r in rows:
        for k, v in marks.items():
            if k.split(':')[0] == r['kernel']:
                r[k.split(':')[1] + '_us'] = v[1] / n    # e.g. chains_end_us: the role's last workgroup, from the kernel's start
                if v[0]:
                    r[k.split(':')[1].replace('_end', '_last_start') + '_us'] = v[0] / n
    return rows


def _latest_profile(kind):
    """profiles/rNN_<kind>.json of the most recent round that has one, or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{kind}.json')))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def measured_traffic(tag):
    """HBM bytes of one launch of the kernel tagged `tag`, from the committed rocprofv3 PMC passes of the same command
    (profiles/README.md: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied); None if absent.
    A bench run cannot read PMC counters itself: the figure is labelled with its source file."""
    data, _ = _latest_profile('traffic')
    try:
        return data[tag]['traffic_bytes']
    except Exception:
        return None


def measured_mfma_util(tag):
    """MFMA utilisation of the kernel tagged `tag` from the committed SQ counter pass (profiles/rNN_mfma.json)."""
    data, _ = _latest_profile('mfma')
    try:
        return data[tag]['mfma_util']
    except Exception:
        return None


def snapshot(gp):
    p = dict(z=gp.z, u_mean=gp.u_mean, u_tril_vec=gp.u_tril_vec, log_mean=gp.kernel.log_mean,
             log_logvar=gp.kernel.log_logvar, prior_log_mean=gp.kernel.prior_log_mean,
             prior_log_logvar=gp.kernel.prior_log_logvar)
    return {k: v.detach().cpu().clone() for k, v in p.items()}


def cpu_baseline(p, x, y, budget_s=15.0, max_steps=30):
    """The oracle (CPU port of the reference algorithm, incl. its full B x B Gram) timed on the host
    cores on the same shapes: zero_grad + loss + backward + a Yogi update (the published algorithm, elementwise on the
    same five tensors), i.e. the same step definition as the GPU side."""
    from oracle import vargp_oracle as orc
    threads = min(os.cpu_count(), 32)     # more threads than this only slows torch's CPU kernels down here
    torch.set_num_threads(threads)
    xc, yc = x.cpu(), y.cpu()
    times, yogi = [], {}
    t_start = time.perf_counter()
    for i in range(max_steps):
        nz = dict(eps_theta=torch.randn(S, D + 1), eps_f=torch.randn(S, F_, C, B))
        t0 = time.perf_counter()
        _, g = orc.elbo_step(p, [], xc, yc, nz, beta=BETA, n_total=N_TOTAL, full_gram=True)
        for k in g:          # Yogi (Zaheer et al. 2018), as vargp_amd/optim.py
            st = yogi.setdefault(k, [torch.full_like(p[k], 1e-6), torch.full_like(p[k], 1e-6)])
            g2 = g[k] * g[k]
            st[0].mul_(0.9).add_(g[k], alpha=0.1)
            st[1].addcmul_(torch.sign(st[1] - g2), g2, value=-0.001)
            b1, b2 = 1 - 0.9 ** (i + 1), 1 - 0.999 ** (i + 1)
            p[k] = p[k] - (LR / b1) * st[0] / ((st[1] / b2).sqrt() + 1e-3)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and i >= 3:
            break
    times = sorted(times[1:]) if len(times) > 1 else times
    med = times[len(times) // 2]
    out = dict(value=1.0 / med, unit='ELBO steps/s', cores=threads, kind='port',
               sample=f'{len(times)} steps (loss + backward + Yogi) of the same Cfg2 workload (S{S} F{F_} C{C} M{M} D{D} B{B}), '
                      f'median; {threads} torch threads (more run slower on this host)')
    # speed calibration of the port against the reference itself (tests/golden/calibrate_cpu.py, run in the build container on
    # identical inputs, where the reference can be imported): port steps/s / reference steps/s
    try:
        with open(os.path.join(ROOT, 'tests', 'golden', 'cpu_calibration.json')) as f:
            cal = json.load(f)
        out['port_vs_reference'] = cal['port_vs_reference']
        out['value_reference_equivalent'] = out['value'] / cal['port_vs_reference']
        out['calibration'] = ('tests/golden/cpu_calibration.json: reference %.2f vs port %.2f steps/s on %d threads of the build '
                              'container (loss + backward)' % (cal['reference_steps_per_s'], cal['port_steps_per_s'], cal['threads']))
    except Exception:
        out['port_vs_reference'] = None
    return out


def elbo_check(gp, x, y):
    """ELBO terms of the HIP path vs the CPU oracle on identical inputs and noise: max relative error over
    (kl_hypers, kl_u, nll, total), evaluated in float64 from the three float32 results.  Large workloads are checked
    on a sub-sample (fewer hyper-samples / minibatch columns, same Mt, same parameters) so that the CPU side stays bounded."""
    from oracle import vargp_oracle as orc
    from vargp_amd import noise
    from vargp_amd.kernels import RBFKernel
    from vargp_amd.likelihoods import MulticlassSoftmax
    from vargp_amd.vargp import VARGP
    Mt = (N_PREV + 1) * M
    big = S * C * Mt ** 3 > 4e10
    Sc, Bc = (min(S, 2), min(B, 128)) if big else (S, B)
    if big:      # same parameters, fewer samples
        g2 = VARGP(gp.z.detach().cpu(), RBFKernel(D), MulticlassSoftmax(n_f=F_), n_var_samples=Sc,
                   prev_params=[{k: p[k].detach().cpu() for k in ('z', 'u_mean', 'u_tril_vec')} for p in gp.prev_params])
        g2.load_state_dict(gp.state_dict())
        gp = g2.to(x.device)
    xs, ys = x[:Bc].contiguous(), y[:Bc].contiguous()
    nz = dict(eps_theta=torch.randn(Sc, D + 1), eps_f=torch.randn(Sc, F_, C, Bc))
    prev = [{k: p[k].detach().cpu() for k in ('z', 'u_mean', 'u_tril_vec')} for p in gp.prev_params]
    if prev:
        nz['eps_u'] = torch.randn(Sc, Sc, C, N_PREV * M)     # KL does not depend on it (ep_var_mean=True)
    with torch.no_grad(), noise.inject(**{k: v.to(x.device) for k, v in nz.items()}):
        got = [float(v) for v in gp.loss(xs, ys)]
    with torch.no_grad():
        ref = [float(v) for v in orc.loss(snapshot(gp), prev, xs.cpu(), ys.cpu(), nz)]
    tot = lambda t: BETA * t[0] + t[1] + (N_TOTAL / Bc) * t[2]
    errs = [abs(a - b) / abs(b) for a, b in zip(got + [tot(got)], ref + [tot(ref)]) if b != 0.0]
    return max(errs), ('sub-sample S=%d B=%d' % (Sc, Bc)) if big else 'full workload'


def dropin_workload(args, device, steps=300, warmup=30):
    """BASELINE config 2 on the DROP-IN route: the reference's own loop shape (experiments/vargp.py:29-37) written against the
    `var_gp` alias package -- `optim.zero_grad(); gp.loss(x, y); combine; loss.backward(); optim.step()`, eager, no trainer, no
    hipGraph, noise drawn by torch -- with this repo's Yogi standing in for torch_optimizer.Yogi.  Timed in the default
    Cholesky error mode ('raise': a host sync per loss call, which is what torch.cholesky's error check costs the reference
    too) and in 'defer' mode (no sync inside the step).  -> dict for the `secondary` object."""
    global S, M, N_PREV
    from var_gp.vargp import VARGP            # noqa: F401  (the alias package: what "import lines only" gives a maintainer)
    from vargp_amd import ops
    from vargp_amd.optim import Yogi
    S, M, N_PREV = 3, 100, 0
    out = dict(workload='BASELINE config 2 through the var_gp alias in the reference loop shape (eager; VARGP.loss returns lazy terms: '
                        'the caller\'s combine + backward() = one program call; program-drawn noise; vargp_amd.optim.Yogi)',
               steps=steps, warmup=warmup)
    for mode in ('raise', 'lazy', 'defer'):
        ops.set_cholesky_error_mode(mode)
        ops.reset_linalg_errors()
        gp, x, y = make_model(device)
        assert isinstance(gp, VARGP)
        optim = Yogi(gp.parameters(), lr=LR)
        N = N_TOTAL

        acc = [0.0] * 5
        pc = time.perf_counter

        def step(rec=False):
            t0_ = pc(); optim.zero_grad()
            t1_ = pc(); kl_hypers, kl_u, lik = gp.loss(x, y)
            t2_ = pc(); loss = BETA * kl_hypers + kl_u + (N / x.size(0)) * lik
            t3_ = pc(); loss.backward()
            t4_ = pc(); optim.step()
            t5_ = pc()
            if rec:
                for i_, d_ in enumerate((t1_ - t0_, t2_ - t1_, t3_ - t2_, t4_ - t3_, t5_ - t4_)):
                    acc[i_] += d_
            return kl_hypers, kl_u, lik

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step(True)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # host time per phase of the caller's loop (no synchronisation inside: what the Python thread spends issuing the step)
        sfx = '' if mode == 'raise' else '_' + mode
        out['host_us' + sfx] = dict(
            total=1e6 * t_host / steps, **{k: 1e6 * a / steps for k, a in zip(('zero_grad', 'loss', 'combine', 'backward', 'optim_step'), acc)})
        out['lazy_terms'] = type(last[0]).__name__ == 'ElboTerm'
        fin = all(bool(torch.isfinite(v)) for v in last)
        out['value' + sfx] = steps / dt
        out['ms_per_step' + sfx] = 1e3 * dt / steps
        out['finite'] = bool(out.get('finite', True) and fin)
        out['programs_cached'] = len(gp._t0_progs)
        gp.release_programs()
        del gp, optim
    ops.set_cholesky_error_mode('defer')
    out['unit'] = 'ELBO steps/s (Cfg2 step, drop-in caller loop)'
    out['cholesky_failures'] = ops.linalg_error_count()
    torch.cuda.empty_cache()
    return out


def epochs_workload(args, device, n_train=12000, epochs=12):
    """BASELINE config 2 as the DRIVER trains it (experiments/vargp.py --graph on a device-resident training set): per epoch one
    on-device permutation, the full minibatches as graphs whose steps gather their own minibatch (ElboTrainer.capture_epoch /
    run_epoch, vargp_gather_minibatch), the ragged last batch through its own captured step, ONE host sync and the deferred Cholesky
    check.  n_train = 12000: a Split-MNIST task (23 full minibatches of 512 + one of 224).  -> dict for the `secondary` object."""
    global S, M, N_PREV
    from vargp_amd import ops
    from vargp_amd.train import ElboTrainer
    S, M, N_PREV = 3, 100, 0
    ops.set_cholesky_error_mode('defer')
    ops.reset_linalg_errors()
    gp, x, y = make_model(device)
    from vargp_amd.synthetic import mnist_like
    xall, yall = mnist_like(n_train, D, C, kind='gauss', seed=1)                # (the data make_model takes its minibatch from)
    data, targets = xall.to(device).contiguous(), yall.to(device).contiguous()
    tr = ElboTrainer(gp, lr=LR, beta=BETA, n_total=n_train)
    tr.capture(data[:B].contiguous(), targets[:B].contiguous())
    ok = tr.capture_epoch(data, targets) is not None
    tail = n_train % B
    if tail:
        tr.capture(data[:tail].contiguous(), targets[:tail].contiguous())

    def epoch():
        order = torch.randperm(n_train, device=device)
        out, done = tr.run_epoch(order)
        if tail:
            out = tr.step_graph_gather(data, targets, order[done * B:])
        failed = ops.linalg_error_count_begin()      # (as the driver: the count rides in front of the epoch's sync)
        torch.cuda.synchronize()
        return out, int(failed)

    epoch()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(epochs):
        out, nb = epoch()
        bad += nb
    dt = time.perf_counter() - t0
    steps = epochs * ((n_train + B - 1) // B)
    vals = [v.item() for v in out]
    return dict(workload='BASELINE config 2 trained the way experiments/vargp.py --graph trains it: %d points resident in HBM, one on-device '
                         'permutation per epoch, the %d full minibatches as graphs of <= 32 steps that gather their own minibatch, the ragged last '
                         'batch, one host sync + the deferred Cholesky check per epoch' % (n_train, n_train // B),
                value=steps / dt, unit='ELBO steps/s', epochs=epochs, steps_per_epoch=(n_train + B - 1) // B, epoch_graphs=bool(ok),
                ms_per_step=1e3 * dt / steps, finite=all(v == v and abs(v) != float('inf') for v in vals), cholesky_failures=int(bad))


# (smnist_s64 / s32 / s16 / s8: one rank's share of BASELINE config 4 at 1 / 2 / 4 / 8 GPUs, measured on this GPU without the
# exchange -- the compute side of the scaling curve DESIGN.md §8 will be held to)
SECONDARY = ['smnist_s64', 'smnist_s32', 'smnist_s16', 'smnist_s8', 'smnist_t1', 'pmnist_t0', 'pmnist_t1', 'pmnist_t4',
             'pmnist_t9']   # + 'stress'


def run_workload(name, args, device, world, rank, use_dist, steps, warmup, primary=True, kern_n=100):
    """One ELBO-step workload: build the model, check it against the CPU oracle, capture, time `steps` steps, time the dominant
    kernels.  -> result dict (rank 0) or None.  primary: the line's top-level fields incl. the CPU baseline."""
    global S, M, N_PREV
    from vargp_amd import _lib, ops
    from vargp_amd.train import ElboTrainer, split_samples
    S, M, N_PREV = (WORKLOADS[name][k] for k in ('S', 'M', 'n_prev'))
    strong = bool(WORKLOADS[name].get('strong'))
    pairs = bool(WORKLOADS[name].get('pairs')) and world > 1      # (one rank: the plain Cfg2 step)
    weak_multi = (not strong) and (not pairs) and world > 1
    counts, s_total = None, None
    if strong:       # a fixed sample total divided over the ranks
        s_total = S
        counts = split_samples(s_total, world)
        S = counts[rank]
    ops.set_cholesky_error_mode('defer')
    ops.reset_linalg_errors()
    gp, x, y = make_model(device)
    rtol, rtol_on = elbo_check(gp, x, y) if rank == 0 else (None, None)
    p0 = snapshot(gp) if rank == 0 and primary else None     # the CPU baseline runs the same (initial) model
    shards = None
    if pairs:        # every rank builds the whole model; the trainer takes its rectangle of the (S, C) grid
        from vargp_amd.train import split_pairs
        shards = split_pairs(S, C, world)
    trainer = ElboTrainer(gp, lr=LR, beta=BETA, n_total=N_TOTAL,
                          sample_counts=None if pairs else ((counts if counts is not None else [S] * world) if use_dist else None),
                          force_exchange=use_dist, comm=args.comm, shards=shards)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.eager
    if use_graph:
        try:
            trainer.capture(x, y)
        except Exception as e:                      # never lose the bench line to a capture problem: run eagerly instead
            if use_dist:                            # (every rank takes the same decision)
                flag = torch.tensor([1.0], device=device)
                dist.all_reduce(flag)
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager steps', file=sys.stderr)
            use_graph = False
        else:
            if use_dist:
                flag = torch.tensor([0.0], device=device)
                dist.all_reduce(flag)
                if flag.item() > 0:                 # some other rank failed to capture
                    use_graph = False
    unroll = args.unroll
    if unroll == 0:          # auto: the largest K <= 10 that divides the number of timed steps, else 4
        unroll = next((k for k in range(10, 1, -1) if steps % k == 0), 4)
    unroll = unroll if (use_graph and not use_dist and unroll > 1) else 1
    if unroll > 1:
        try:
            trainer.capture_unrolled(x, y, unroll)
        except Exception as e:
            print(f'[bench] unrolled capture failed ({type(e).__name__}: {e}); one step per graph', file=sys.stderr)
            unroll = 1
    if use_graph:
        run = trainer.step_graph
    else:
        run = lambda: trainer.step(x, y)
    # The CPU oracle (ELBO check above, cpu_baseline of the previous workload) leaves its OpenMP workers spinning for ~200 ms
    # after their last parallel region; on a box whose CPU quota they exhaust, the launching thread is starved and a short timed
    # region (30 steps = 18 ms) picks up a 20 ms stall (seen as mean = 2 x median on smnist_t1 in the default line only).  Let
    # them go to sleep before the warm-up.  The device is kept busy meanwhile with plain matrix products on scratch tensors (no
    # step of the model): a GPU that idles through those 0.3 s drops its clocks, and a short run (the driver's --steps 20
    # --warmup 5 is 5 ms in all) then measures the ramp, not the training rate (0.214 against 0.204 ms per step).
    if device.type == 'cuda':
        a_ = torch.empty(4096, 4096, device=device).normal_()
        t_ = time.perf_counter()
        while time.perf_counter() - t_ < PREHEAT_S:
            for _ in range(8):
                torch.mm(a_, a_)
            torch.cuda.synchronize()
        del a_
    else:
        time.sleep(0.3)
    for _ in range(warmup):
        run()
    if unroll > 1:
        trainer.step_graph_k()           # (untimed: the K-step graph's first launch uploads it)
    sync()
    # EXACTLY `steps` steps: steps // unroll launches of the K-step graph, the remainder as one-step launches
    plan = [(trainer.step_graph_k, unroll)] * (steps // unroll) + [(run, 1)] * (steps % unroll) if unroll > 1 else [(run, 1)] * steps
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(plan) + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i, (fn, _) in enumerate(plan):
        out = fn()
        evs[i + 1].record()              # per-launch device time (median below); `value` uses the wall clock of all K steps
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) / plan[i][1] for i in range(len(plan)))
    median_ms = per_step[len(per_step) // 2]
    final_loss = [v.item() for v in out]     # read now: the re-launches below include the kernel that resets the accumulators
    finite = all(v == v and abs(v) != float('inf') for v in final_loss)
    # The exchange, measured two ways (outside the timed region): in place -- hipEvents around the collective of `n_comm` more
    # steps (includes waiting for the slowest rank) -- and isolated -- the same collective back to back after a barrier.
    comm_us = comm_iso_us = None
    seen = ranks_seen(world, rank, device) if use_dist else [0]
    if use_dist:
        n_comm = max(5, min(steps, 50))
        trainer.comm_events = []
        for _ in range(n_comm):
            run()
        sync()
        comm_us = sum(a.elapsed_time(b) for a, b in trainer.comm_events) / len(trainer.comm_events) * 1e3
        trainer.comm_events = []
        trainer.flat.zero_()                      # repeated sums of zeros stay finite; the next step overwrites the buffer
        for _ in range(n_comm):
            trainer.exchange()
        sync()
        comm_iso_us = sorted(a.elapsed_time(b) for a, b in trainer.comm_events)[len(trainer.comm_events) // 2] * 1e3
        trainer.comm_events = None
        t = torch.tensor([comm_us, comm_iso_us], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm_us, comm_iso_us = t.tolist()
    # Kernel times, two ways.  (a) IN THE STEP: the kernels of the first-task program stamp the device's wall clock themselves
    # (vargp_prof_spans) -- the only clock that reaches inside a replayed hipGraph; `roofline` is computed from these.
    # (b) ISOLATED: one more (eager) step is run with launch recording on, and the recorded launches -- same kernels, same
    # shapes, the trainer's live buffers -- are re-launched back to back between ONE pair of hipEvents (vargp_prof_remember /
    # vargp_prof_replay): L2-warm, no neighbours, reported as `frac_isolated`; the only figure for the block program's kernels.
    #   first task: "chol_rbf_gemm" = K_uu / S_u factorisations + the K_uf distance GEMM in one launch; "rbf_kuf_bwd_gemm" =
    #   t0_bwdmat_gemm_kernel (P_uf = W_uf x beside the adjoint chains of the factorisations).
    block_prog = bool(gp._use_block_program(B))      # which native program runs this model (vargp.py)
    timeline = []
    if use_graph and not block_prog and not args.no_timeline and not use_dist:       # (several ranks: the exchange sits inside the step)
        try:
            # (the stamps of the LAST step of a launch are read back: of the K-step graph when that is what was timed)
            timeline = step_timeline(trainer.step_graph_k if unroll > 1 else run, 1e3 * median_ms, burst=1 if unroll > 1 else 4)
        except Exception as e:
            print(f'[bench] step timeline failed ({type(e).__name__}: {e})', file=sys.stderr)
    _lib.prof_enable(False)
    _lib.prof_read('')
    _lib.prof_remember(True)
    trainer.step(x, y)
    sync()
    _lib.prof_remember(False)
    kernels = {}
    # (config 2 over several ranks: this rank's part of the S x C problems -- a class range, or whole samples when world <= S)
    share = ((trainer.rect[1] - trainer.rect[0]) * (trainer.rect[3] - trainer.rect[2]) / float(S * C)) if pairs else 1.0
    flops_kuf = share * 2.0 * S * C * M * (N_PREV + 1) * B * D     # one K_uf launch of compute_pf_diag
    Mt = M * (N_PREV + 1)
    if not block_prog:
        candidates = [('chol_rbf_gemm', flops_kuf, 'chol_rbf_gemm_kernel (K_uu/S_u factorisations -- register-resident pivot chains, four '
                       'pivots per barrier -- sharing one launch with the K_uf = rbf(z, x) distance GEMM; flops counted: the GEMM)'),
                      ('rbf_kuf_bwd_gemm', 2.0 * S * C * M * B * D, 't0_bwdmat_gemm_kernel (P_uf = W_uf x of the kernel-matrix '
                       'backward, [C*M x B] x [B x D] per sample, sharing one launch with the per-matrix adjoint chains of '
                       'the factorisations; flops counted: the product)'),
                      ('rbf_kuu_bwd_gemm', 2.0 * S * C * M * M * D, 'gemm_kernel P_uu = W_uu z of the kernel-matrix backward '
                       '([M x M] x [M x D] per (sample, class); more than 4 samples only: otherwise inside the last launch)'),
                      ('t0_pro_kuu', 2.0 * S * C * M * M * D, 't0_pro_kuu_kernel (K-split inner products of K_uu = rbf(z, z) next '
                       'to the prologue roles and the row norms; flops counted: the product)'),
                      ('rbf_kuf_gemm', flops_kuf, 'gemm_kernel<RBF> (K_uf = rbf(z, x))'),
                      ('rbf_kuu_gemm', 2.0 * S * C * M * D * (B + M), 'gemm_pair_kernel<RBF> (K_uu = rbf(z, z) and K_uf = rbf(z, x) '
                       'in one launch)')]
    else:
        # models with previous tasks (csrc/elbo_tn.hip): the K_uf GEMM is cut into slices that run beside the pivot chains
        # of the blocked factorisation, so the roofline objects are the big stand-alone products; ALGORITHMIC flops
        # (symmetric / triangular work counted once), the longest of them is `roofline`
        candidates = [('rbf_kuu_gemm', 1.0 * S * C * Mt * Mt * D, 'gemm_kernel<RBF> K(z_<=t, z_<=t): tiles touching the lower '
                       'triangle, mirrored (S*C*Mt^2*D flop: symmetric work counted once)'),
                      ('tn_p_gemm', 1.0 * S * C * Mt * Mt * B, 'gemm_kernel P = T K_uf (T lower triangular: S*C*Mt^2*B flop)'),
                      ('rbf_kuu_bwd_gemm', 2.0 * S * C * Mt * Mt * D, 'gemm_kernel W_uu z of the kernel-matrix backward '
                       '([Mt x Mt] x [Mt x D] per (sample, class))'),
                      ('rbf_kuf_bwd_gemm', 2.0 * S * C * Mt * B * D, 'gemm_kernel W_uf x of the kernel-matrix backward '
                       '([C*Mt x B] x [B x D] per sample)'),
                      ('tn_chol_bwd3', S * C * Mt * Mt * Mt / 3.0, 'gemm_kernel gK = T^T tril(Smat T), lower triangle mirrored '
                       '(both factors lower triangular: S*C*Mt^3/3 flop)')]
    for tag, fl, desc in candidates:
        if args.no_replay:
            break
        try:
            kernels[tag] = (_lib.prof_replay(tag, kern_n), fl, desc)
        except _lib.VargpHipError:
            pass
    if 'rbf_kuu_bwd_gemm' in kernels and 'rbf_kuf_bwd_gemm' not in kernels:
        # shapes outside the LDS-resident backward run both W.Y products in ONE pair launch (recorded under the first tag)
        us, _, _ = kernels['rbf_kuu_bwd_gemm']
        kernels['rbf_kuu_bwd_gemm'] = (us, 2.0 * S * C * Mt * D * (Mt + B), 'gemm_pair_kernel (both W.Y products of the '
                                       'kernel-matrix backward in one launch)')
    if not kernels:
        kernels['none'] = (float('nan'), 0.0, 'not timed (--no-replay)')
    if not block_prog:
        dom = next(t for t in ('chol_rbf_gemm', 'rbf_kuf_gemm', 'rbf_kuu_gemm', 'none') if t in kernels)
    else:
        dom = max(kernels, key=lambda t: kernels[t][0] if kernels[t][0] == kernels[t][0] else -1.0)
    kern_us, dominant_flops, dominant_desc = kernels[dom]
    iso_us = kern_us
    # in-step figures: algorithmic flop of every launch of the step (SURVEY 8d terms: symmetric / triangular work once)
    tl_info = {'t0_pro_kuu': ('t0_pro_kuu', 1.0 * S * C * M * M * D), 'chol_rbf_gemm': ('chol_rbf_gemm', flops_kuf),
               'gemm_kernel': (None, 1.0 * S * C * M * M * (M + 2)), 't0_fwd_fused': (None, 2.0 * S * C * M * M * B),
               't0_bwd_mid': (None, 4.0 * S * C * M * M * B), 't0_bwdmat_gemm': ('rbf_kuf_bwd_gemm', 2.0 * S * C * M * B * D),
               't0_puu_final': (None, 2.0 * S * C * M * M * D), 'yogi_multi': (None, 0.0)}
    for r in timeline:
        tag, fl = tl_info.get(r['kernel'], (None, 0.0))
        r['flop'] = fl = fl * (share if r['kernel'] != 'chol_rbf_gemm' else 1.0)
        # slot = previous kernel's last workgroup end .. this kernel's last workgroup end: the wall time of the step that belongs
        # to this launch (launch latency and the previous kernel's cache write-back included; the slots sum to the step period).
        # rocprofv3's dispatch-to-completion duration lies between span and slot.
        r['frac'] = fl / (r['slot_us'] * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS if fl > 0 and r['slot_us'] > 0 else None
        r['isolated_us'] = kernels[tag][0] if tag in kernels else None
    dom_row = None
    if timeline:
        # the roofline object describes the LONGEST launch of the replayed step (by its in-step span)
        dom_row = max((r for r in timeline if r['flop'] > 0), key=lambda r: r['slot_us'])
        tag = tl_info[dom_row['kernel']][0]
        if tag in kernels:
            dom = tag
            iso_us, dominant_flops, dominant_desc = kernels[tag]
        else:
            dom, iso_us, dominant_flops = dom_row['kernel'], float('nan'), dom_row['flop']
            dominant_desc = dom_row['kernel'] + '_kernel (algorithmic flop of its products, SURVEY 8d)'
        kern_us = dom_row['slot_us']
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    errs = ops.linalg_error_count()
    res = None
    if rank == 0:
        # strong / class-sharded: GLOBAL steps per second of the fixed job.  weak: every rank processes one config-2-sized ELBO +
        # gradient (its own S hyper-samples) per step -- the units all ranks processed per second
        opt_steps = steps / dt
        value = opt_steps * world if weak_multi else opt_steps
        avg_s = kern_us * 1e-6
        achieved = dominant_flops / avg_s / 1e12 if kern_us == kern_us and kern_us > 0 else None
        cfg2 = name == 'smnist'
        if pairs:
            unit = 'ELBO steps/s (global steps of the Cfg2 ELBO, %d (sample, class) problems over %d ranks)' % (S * C, world)
        elif strong:
            unit = 'ELBO steps/s (global steps of the %d-sample ELBO)' % s_total
        elif weak_multi:
            unit = ('ELBO steps/s (ELBO + gradient evaluations over %d hyper-samples x %d classes, summed over the %d ranks: '
                    '%d x the optimizer steps/s of the %d-sample step)' % (S, C, world, world, S * world))
        else:
            unit = 'ELBO steps/s (%s step: S=%d hyper-samples)' % ('Cfg2' if cfg2 else name, S)
        res = dict(metric='ELBO steps/sec', value=value, unit=unit,
                   n_gpus=world, steps=steps, warmup=warmup, ms_per_step=1e3 * dt / steps,
                   ms_per_step_median=median_ms, optimizer_steps_per_s=opt_steps,
                   higher_is_better=True, scaling='strong' if (strong or pairs) and world > 1 else 'weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload=WORKLOADS[name]['desc'], S_per_gpu=counts if strong else (None if pairs else S), Mt=M * (N_PREV + 1),
                               S_total=s_total if strong else (S if pairs else S * world), F=F_, C=C, M=M, D=D, B=B, N=N_TOTAL, beta=BETA,
                               shards=shards, shard_weights=([round((r[1] - r[0]) * (r[3] - r[2]) / float(S * C), 6) for r in shards] if shards
                                                             else ([c / float(s_total) for c in counts] if strong else None)),
                               optimizer='yogi', parallelism=(f'(sample, class)-parallel x{world}' if pairs else f'sample-parallel x{world}'),
                               steps_per_graph_launch=unroll,
                               launch=('hipGraph replay' + ((' (3 graphs around the all-gather and the all-reduce)' if (pairs and trainer.class_split) else ' (2 graphs around the all-reduce)') if use_dist else '')) if use_graph
                               else 'eager'),
                   elbo_rtol_vs_cpu=rtol, elbo_rtol_checked_on=rtol_on, finite=bool(finite), cholesky_failures=errs,
                   ranks_seen=seen, comm=args.comm if use_dist else None,
                   allreduce_us=comm_us, allreduce_us_isolated=comm_iso_us,
                   allreduce_bytes=trainer.flat.numel() * 4 if use_dist else None,
                   allgather_bytes=(trainer.max_pairs * 2 * B * 4 * world) if (pairs and trainer.class_split) else None,
                   final_loss=dict(kl_hypers=final_loss[0], kl_u=final_loss[1], nll=final_loss[2]),
                   roofline=dict(bound='mfma', kernel=dominant_desc,
                                 achieved=achieved, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                                 frac=(achieved / MFMA_F32_PEAK_TFLOPS) if achieved else None,
                                 launches=kern_n, avg_us=avg_s * 1e6,
                                 timing=('in the step: wall-clock stamps written by the kernels inside the replayed hipGraph '
                                         '(vargp_prof_spans); avg_us = the launch\'s slot = previous kernel\'s last workgroup end .. this '
                                         'kernel\'s last workgroup end (slots sum to ms_per_step); span_us = first workgroup start .. last '
                                         'workgroup end; last of 4 back-to-back replays, mean of 20') if dom_row
                                 else 'isolated: 100 back-to-back re-launches between one hipEvent pair',
                                 span_us=dom_row['span_us'] if dom_row else None,
                                 # the shader clock MEASURED under this launch (s_memtime against the 100 MHz wall clock, workgroup
                                 # 0 of the launch, inside the replayed graph); below the 2.4 GHz the peak is quoted at, the
                                 # fraction of the roof the chip offered at that clock rides along
                                 clock_ghz=dom_row.get('clock_ghz') if dom_row else None,
                                 frac_at_held_clock=((achieved / MFMA_F32_PEAK_TFLOPS) * PEAK_CLOCK_GHZ / dom_row['clock_ghz'])
                                 if (dom_row and achieved and dom_row.get('clock_ghz') and dom_row['clock_ghz'] < PEAK_CLOCK_GHZ) else None,
                                 frac_span=(dominant_flops / (dom_row['span_us'] * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS) if dom_row else None,
                                 frac_isolated=(dominant_flops / (iso_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS)
                                 if iso_us == iso_us and iso_us > 0 else None,
                                 isolated_us=iso_us if iso_us == iso_us else None,
                                 traffic=measured_traffic(dom) if cfg2 else None,
                                 counters_from=_latest_profile('traffic')[1] if cfg2 else None,
                                 counters_commit=(_latest_profile('traffic')[0] or {}).get('commit') if cfg2 else None))
        # whole step against the same roof: SURVEY 8(d) algorithmic flop of the step / wall time of the step
        step_flop = 3.0 * survey_flops_fwd(s_total if strong else (S if pairs else S * world), C, M, N_PREV, B, D, F_)
        res['step_flop'] = step_flop
        res['step_frac'] = step_flop / (dt / steps) / 1e12 / MFMA_F32_PEAK_TFLOPS / world
        res['preheat_s'] = PREHEAT_S if device.type == 'cuda' else 0.0
        if timeline:
            res['timeline'] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in timeline]
            res['timeline_sum_us'] = dict(spans=sum(r['span_us'] for r in timeline), gaps=sum(r['gap_before_us'] for r in timeline))
        if N_PREV > 0:
            res['roofline_others'] = [dict(kernel=dsc, avg_us=us, achieved=fl / (us * 1e-6) / 1e12,
                                           frac=fl / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS)
                                      for t, (us, fl, dsc) in kernels.items() if t != dom and us == us and fl > 0]
        gtag = next((t for t in ('rbf_kuf_bwd_gemm', 'rbf_kuu_bwd_gemm') if t in kernels), None)
        if gtag == dom and 'chol_rbf_gemm' in kernels:
            gtag = 'chol_rbf_gemm'                     # (the backward's launch is `roofline` itself: report the forward's here)
        if N_PREV == 0 and gtag and dom != gtag:       # the other GEMM-carrying launch of the step
            us2, fl2, desc2 = kernels[gtag]
            res['roofline_gemm'] = dict(bound='mfma', kernel=desc2, achieved=fl2 / (us2 * 1e-6) / 1e12,
                                        peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                                        frac=fl2 / (us2 * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, launches=kern_n, avg_us=us2,
                                        timing='isolated: back-to-back re-launches between one hipEvent pair',
                                        traffic=measured_traffic(gtag) if cfg2 else None)
            if not block_prog:
                res['roofline_others'] = [dict(kernel=dsc, avg_us=us, achieved=fl / (us * 1e-6) / 1e12,
                                               frac=fl / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS)
                                          for t, (us, fl, dsc) in kernels.items() if t not in (dom, gtag) and us == us and fl > 0]
        if world == 1 and primary and not args.no_cpu_baseline and cfg2:
            res['cpu_baseline'] = cpu_baseline(p0, x, y)
    # hand the workspaces back before the next workload (12 GB at Permuted-MNIST task 9)
    del trainer, run, out
    gp.release_programs()
    del gp
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def secondary_summary(res):
    """The fields of a secondary workload's result that go into the default line."""
    keep = ('value', 'unit', 'ms_per_step', 'ms_per_step_median', 'steps', 'warmup', 'elbo_rtol_vs_cpu', 'elbo_rtol_checked_on',
            'finite', 'cholesky_failures', 'predictive_sweep', 'roofline_chol')
    out = {k: res[k] for k in keep if k in res}
    out['workload'] = res['config']['workload']
    if 'launch' in res['config']:
        out['launch'] = res['config']['launch']
    r = res['roofline']
    out['roofline'] = dict(kernel=r['kernel'][:90], frac=r['frac'], achieved=r['achieved'], unit=r['unit'], avg_us=r['avg_us'])
    if res.get('roofline_others'):
        out['roofline_others'] = [dict(kernel=o['kernel'][:60], frac=o['frac'], avg_us=o['avg_us']) for o in res['roofline_others']]
    return out


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: run the N ranks under torch.distributed.run as a child process and
    forward its output (rank 0 prints the one JSON line) and exit code.  Nothing here touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this image
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    print('[bench] --gpus %d without a launcher: starting %s' % (n, ' '.join(cmd)), file=sys.stderr)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith('{')]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)                     # anything that is not THE line goes to stderr
    if js:
        print(js[-1])
    sys.stdout.flush()
    sys.exit(proc.returncode if proc.returncode != 0 or js else 1)


def ranks_seen(world, rank, device):
    """All-gather of the rank ids: what the line reports as `ranks_seen` (every rank really took part)."""
    if world == 1 or not dist.is_initialized():
        return [rank]
    mine = torch.tensor([rank], dtype=torch.int64, device=device)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [int(t.item()) for t in got]


def dry_run(args, world, rank, local):
    """Bring the process group up exactly as a real run would (RCCL when every rank has its own GPU, gloo otherwise),
    all-gather the rank ids, print the line and leave: no model, no oracle, no kernel."""
    n_dev = torch.cuda.device_count()          # does not initialise the GPU
    backend = 'nccl' if n_dev >= world and n_dev > 0 else 'gloo'
    device = torch.device('cpu')
    if backend == 'nccl':
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group(backend, **(dict(device_id=device) if backend == 'nccl' else {}))
    seen = ranks_seen(world, rank, device)
    if world > 1:
        buf = torch.ones(world * 3 + 1, device=device)          # one exchange of each kind on a toy buffer
        dist.all_reduce(buf)
        ok = bool((buf == world).all().item())
        pad = torch.ones(world * 4, device=device)
        shard = torch.zeros(4, device=device)
        dist.reduce_scatter_tensor(shard, pad)
        dist.all_gather_into_tensor(pad, shard)
        ok = ok and bool((pad == world).all().item())
        dist.barrier()
    else:
        ok = True
    if rank == 0:
        print(json.dumps(dict(dry_run=True, ok=ok, n_gpus=world, ranks_seen=seen, backend=backend, comm=args.comm,
                              workload=args.workload or ('smnist_s64' if world > 1 and args.scaling == 'strong' else 'smnist'),
                              devices_visible=n_dev)))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if ok and sorted(seen) == list(range(world)) else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='do not replay the step from a captured hipGraph')
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS) + ['stress'],
                    help='default: the BASELINE metric workload (Cfg2), 3 hyper-samples per GPU; the others are secondary '
                         'measurements (several GPUs: smnist_s64 = BASELINE config 4, a fixed 64 samples split over the ranks)')
    ap.add_argument('--scaling', default=None, choices=['strong', 'weak'],
                    help='several GPUs only: weak (default) = Cfg2 with 3 hyper-samples per rank; strong = smnist_s64 (Cfg4) as the headline')
    ap.add_argument('--no-secondary', action='store_true',
                    help='one GPU, default workload: skip the short runs of the other BASELINE configs (`secondary`)')
    ap.add_argument('--secondary-budget', type=float, default=240.0,
                    help='seconds after which no further secondary workload is started')
    ap.add_argument('--stress-n', type=int, default=1000000)
    ap.add_argument('--unroll', type=int, default=0,
                    help='one GPU, hipGraph mode: steps per graph launch (ElboTrainer.capture_unrolled); 1 = one step per launch, '
                         '0 (default) = the largest K <= 10 dividing --steps')
    ap.add_argument('--no-timeline', action='store_true',
                    help='skip the in-step kernel time line (vargp_prof_spans: extra graph replays after the timed region)')
    ap.add_argument('--no-replay', action='store_true',
                    help='skip the back-to-back re-launches that time the dominant kernels (keeps a rocprof trace clean)')
    ap.add_argument('--comm', default='allreduce', choices=['allreduce', 'rsag'],
                    help='several GPUs: sum the flat [grads | kl_u | nll] buffer with one all-reduce, or with a reduce-scatter + '
                         'all-gather of the same buffer (every shard over its own xGMI link)')
    ap.add_argument('--dry-run', action='store_true',
                    help='start the ranks, create the process group, all-gather the rank ids, print a line and exit')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        self_launch(args.gpus)                   # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it launches its '
                 f'own ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}')
    if args.dry_run:
        dry_run(args, world, rank, local)        # does not return
    default_line = args.workload is None
    if args.workload is None:
        args.workload = 'smnist_s64' if (world > 1 and args.scaling == 'strong') else 'smnist'
    # VARGP_BENCH_ONE_GPU=1 (smoke test on a one-GPU box): all ranks share GPU 0 and gloo carries the exchanges -- the whole
    # multi-rank path of this file (uneven shards, class-sharded secondary, two / three graphs) without a multi-GPU node
    one_gpu = os.environ.get('VARGP_BENCH_ONE_GPU', '0') == '1'
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # VARGP_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL group, two-graph capture around the all-reduce, barriers)
    # with a single rank -- a smoke test of that path on a one-GPU box
    use_dist = world > 1 or os.environ.get('VARGP_BENCH_FORCE_DIST', '0') == '1'
    if use_dist:
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    if args.workload == 'stress':
        print(json.dumps(stress(args, device, cpu=not args.no_cpu_baseline)))
        return
    # (the ranks are counted right after the group is up, so that a failing run still reports who took part)
    seen0 = ranks_seen(world, rank, device) if use_dist else [0]
    try:
        res = run_workload(args.workload, args, device, world, rank, use_dist, args.steps, args.warmup, primary=True)
    except Exception as e:
        if rank == 0:
            print(json.dumps(dict(metric='ELBO steps/sec', value=None, unit='ELBO steps/s', n_gpus=world, steps=args.steps,
                                  warmup=args.warmup, ranks_seen=seen0, error=f'{type(e).__name__}: {e}'[:500],
                                  config=dict(workload=args.workload))))
            sys.stdout.flush()
        raise
    if world > 1 and default_line and not args.no_secondary and os.environ.get('VARGP_BENCH_PAIRS', '1') != '0':
        # several GPUs, besides the headline: (1) BASELINE config 4 as north_star states it -- a FIXED 64 hyper-samples x 10 classes,
        # whole samples per rank (strong scaling; skipped when it IS the headline); (2) the metric's config 2 -- 3 samples x 10
        # classes -- class-sharded over the same ranks (train.split_pairs).  Every rank runs them; a failure is reported inside the
        # line, the headline stays.
        keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'elbo_rtol_vs_cpu', 'finite', 'cholesky_failures',
                'allreduce_us', 'allreduce_us_isolated', 'allreduce_bytes', 'allgather_bytes', 'ranks_seen', 'step_frac')
        sec = {}
        for wname in (['smnist_s64'] if args.workload != 'smnist_s64' else []) + ['smnist_pairs']:
            try:
                r2 = run_workload(wname, args, device, world, rank, use_dist, max(20, min(args.steps, 100)), 5, primary=False,
                                  kern_n=20)
                if rank == 0:
                    sec[wname] = dict({k: r2[k] for k in keep if k in r2}, workload=r2['config']['workload'],
                                      S_per_gpu=r2['config']['S_per_gpu'], shards=r2['config']['shards'],
                                      shard_weights=r2['config']['shard_weights'], launch=r2['config']['launch'])
            except Exception as e:
                if rank == 0:
                    sec[wname] = dict(error=f'{type(e).__name__}: {e}'[:300])
                torch.cuda.empty_cache()
        if rank == 0:
            res['secondary'] = sec
            # The headline of an N-rank line is WEAK scaling (per-GPU work fixed: `value` = N x the optimizer steps/s, as its unit
            # says): so that nobody reads it as the step rate of a fixed job, the two fixed jobs' global step rates ride at the top
            # level under their own names -- these are the STRONG-scaling figures comparable with secondary.smnist_s64 /
            # the headline of the N = 1 line.
            res['optimizer_steps_per_s_headline'] = res.get('optimizer_steps_per_s')
            res['value_strong_fixed_64_samples'] = sec.get('smnist_s64', {}).get('value')
            res['value_strong_cfg2_30_problems'] = sec.get('smnist_pairs', {}).get('value')
    if rank == 0 and world == 1 and default_line and not args.no_secondary and not use_dist:
        # short, driver-timed runs of the other BASELINE configs in the same line (their own step counts are stated)
        t_start = time.perf_counter()
        sec = {}
        for name in ['smnist_dropin', 'smnist_epochs'] + SECONDARY + ['stress']:
            if time.perf_counter() - t_start > args.secondary_budget:
                sec[name] = dict(skipped='secondary budget of %.0f s used up' % args.secondary_budget)
                continue
            try:
                if name == 'smnist_dropin':
                    sec[name] = dropin_workload(args, device)
                    sec[name]['vs_trainer'] = sec[name]['value'] / res['value']
                    sec[name]['vs_trainer_lazy'] = sec[name]['value_lazy'] / res['value']
                    sec[name]['vs_trainer_defer'] = sec[name]['value_defer'] / res['value']
                    continue
                if name == 'smnist_epochs':
                    sec[name] = epochs_workload(args, device)
                    sec[name]['vs_trainer'] = sec[name]['value'] / res['value']
                    continue
                if name == 'stress':
                    r = stress(args, device, cpu=False)
                else:
                    big = WORKLOADS[name]['n_prev'] >= 4
                    r = run_workload(name, args, device, 1, 0, False, 10 if big else 30, 3, primary=False,
                                     kern_n=10 if big else 50)
                sec[name] = secondary_summary(r)
            except Exception as e:           # a secondary workload never costs the headline line
                sec[name] = dict(error=f'{type(e).__name__}: {e}'[:300])
                torch.cuda.empty_cache()
        res['secondary'] = sec
        res['secondary_seconds'] = time.perf_counter() - t_start
        # a failing secondary workload never costs the headline line (exit code stays 0), but it is counted at the top level
        bad = [n for n, r in sec.items() if 'error' in r or r.get('finite') is False or (r.get('cholesky_failures') or 0) > 0]
        res['secondary_failed'] = len(bad)
        res['secondary_failed_names'] = bad
    if rank == 0:
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
