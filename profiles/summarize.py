"""Turn the raw rocprofv3 output of one round (under gpurun_out/, scratch: produced by `bash profiles/collect.sh <tag>` on
the GPU box) into the committed summaries here.

    python profiles/summarize.py r03

Outputs (profiles/<tag>_*):
  bench.json                       the default `python bench.py` line of this round
  lines.json                       the bench lines of the secondary workloads (other BASELINE configs)
  bench_{graph,eager}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench (graph replay / eager)
  replay_kernel_stats.csv          same for the run that includes bench.py's back-to-back re-launches (the roofline timing)
  top_kernels.md                   per-step table from the eager run (exact launches per step)
  pmc_<counter>_<kernel>.csv       PMC rows of the headline kernels (evidence behind traffic.json / mfma.json)
  traffic.json                     HBM traffic per launch: 1024 * (2 * FETCH_SIZE + WRITE_SIZE)  [FETCH_SIZE in KB reports half
                                   the bytes of wide coalesced reads on gfx950, MI355X_MICROARCH.md HBM section]
  mfma.json                        MFMA utilisation per launch = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE)
                                   (GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 SIMDs: calibrated on the 4096^3 GEMM, where the
                                   formula gives 0.90 against 141 TFLOP/s = 0.90 of the f32-MFMA peak by the clock), and the
                                   wave-cycle split ACTIVE / WAIT_INST (issue stalls) / WAIT (parked at waitcnt or barrier)
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles')
S, C, M, D, B = 3, 10, 100, 784, 512
# headline kernels of the default workload: tag -> (kernel-name prefix, algorithmic bytes per launch, flops per launch)
KERNELS = {
    # factorisations: read K_uu/S_u, write L and T ((S*C + C) matrices); GEMM: read z and x once, write K_uf
    'chol_rbf_gemm': ('void vargp::chol_rbf_gemm_kernel',
                      4 * (3 * (S * C + C) * M * M + C * M * D + B * D + S * C * M * B), 2.0 * S * C * M * B * D),
    # P_uf = W_uf x: read W_uf (S*C*M*B) and x once, write P_uf (S*C*M*D); adjoint chains in the same launch: per (s, c) read
    # T, the tile kernel's gT, K_uu and write W_uu (4 M^2); per class read T_s and gG_s of every sample, L_S, T_S, write gS_u
    'rbf_kuf_bwd_gemm': ('void vargp::t0_bwdmat_gemm_kernel',
                         4 * (S * C * M * B + B * D + S * C * M * D + 4 * S * C * M * M + (2 * S + 3) * C * M * M),
                         2.0 * S * C * M * B * D),
    # tile kernel of the backward: read QP-side operands per (s, c, 64-column tile), write W_uf; everything else is atomics
    't0_bwd_mid': ('void vargp::t0_bwd_mid_kernel', 4 * (2 * S * C * M * B + 3 * S * C * M * M + 3 * S * C * M * M),
                   8.0 * S * C * M * M * B),
    # front launch: read z once for the product and once for the norms, x once; write the two partial Gram matrices per (s, c)
    't0_pro_kuu': ('void vargp::t0_pro_kuu_kernel', 4 * (2 * C * M * D + B * D + 2 * S * C * M * M), 2.0 * S * C * M * M * D),
    # last launch: read W_uu, P_uf, z; write gz
    't0_puu_final': ('void vargp::t0_puu_final_kernel', 4 * (S * C * M * M + S * C * M * D + 2 * C * M * D + B * D),
                     2.0 * S * C * M * M * D),
    't0_fwd_fused': ('void vargp::t0_fwd_fused_kernel', 4 * (S * C * M * B + 2 * S * C * M * M + 2 * S * C * M * B),
                     4.0 * S * C * M * M * B),
    # small-column product QP[:, :NR] = T RK[:, :NR] (a, G, G2): read T and the NR = 204 small columns, write them
    'gemm_kernel': ('void vargp::gemm_kernel<64, 64, 64, true, false, true, false, true>', 4 * (S * C * M * M + 2 * S * C * M * 204),
                    1.0 * S * C * M * M * 204),
    # optimiser: p, g, m, v read; p, m, v written
    'yogi_multi': ('void vargp::yogi_multi_kernel', 4 * 7 * (C * M * D + C * M + C * M * (M + 1) // 2 + 2 * (D + 1)), 0.0),
}
# ALGORITHMIC flop of each launch of the step (SURVEY 8d terms, symmetric / triangular work once; bench.py: tl_info) and the
# name bench.py's `timeline` uses for it
STEP_FLOP = {'t0_pro_kuu': 1.0 * S * C * M * M * D, 'chol_rbf_gemm': 2.0 * S * C * M * B * D, 'gemm_kernel': 1.0 * S * C * M * M * (M + 2),
             't0_fwd_fused': 2.0 * S * C * M * M * B, 't0_bwd_mid': 4.0 * S * C * M * M * B, 'rbf_kuf_bwd_gemm': 2.0 * S * C * M * B * D,
             't0_puu_final': 2.0 * S * C * M * M * D, 'yogi_multi': 0.0}
TIMELINE_NAME = {'rbf_kuf_bwd_gemm': 't0_bwdmat_gemm'}
PEAK_TF = 157.3
SECONDARY = ['smnist_s64', 'smnist_t1', 'smnist_t4', 'pmnist_t0', 'pmnist_t1', 'pmnist_t4', 'pmnist_t9', 'stress']
# kernels of the N = 1e6 sweep (M = 2048, C = 10, S = 1, tile 8192), counters from the short sweep of collect.sh:
# tag -> (kernel-name prefix, grid size in threads, algorithmic bytes, flops)
MS, TL = 2048, 8192
STRESS_KERNELS = {
    # K_uf tile: read z (C*M x D) and the pre-scaled minibatch tile once, write K_uf
    'stress_kuf_tile': ('void vargp::gemm_kernel<128, 128, 16, true, true, true, true, false>', 10 * (MS // 128) * (TL // 128) * 256,
                        4 * (C * MS * D + TL * D + C * MS * TL), 2.0 * C * MS * TL * D),
    # P = T K_uf (T lower triangular): read T and K_uf, write P
    'stress_p_gemm': ('void vargp::gemm_kernel<128, 64, 32, true, false, true, false, true>', 10 * (MS // 128) * (TL // 64) * 256,
                      4 * (C * MS * MS // 2 + 2 * C * MS * TL), 1.0 * C * MS * MS * TL),
}


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    return None


def pmc_rows(path, prefix):
    bare = prefix[5:] if prefix.startswith('void ') else prefix       # non-template kernels are listed without the return type
    return [r for r in csv.DictReader(open(path)) if r['Kernel_Name'].startswith(prefix) or r['Kernel_Name'].startswith(bare)]


def avg(rows, counter):
    v = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == counter]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def keep(rows, name):
    if not rows:
        return
    with open(os.path.join(OUT, name), 'w', newline='') as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:200])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    G = os.path.join(ROOT, 'gpurun_out')
    src = lambda d, f: os.path.join(G, f'{tag}_{d}', f)
    shutil.copy(src('graph', 'p_kernel_stats.csv'), os.path.join(OUT, f'{tag}_bench_graph_kernel_stats.csv'))
    shutil.copy(src('eager', 'p_kernel_stats.csv'), os.path.join(OUT, f'{tag}_bench_eager_kernel_stats.csv'))
    shutil.copy(src('replay', 'p_kernel_stats.csv'), os.path.join(OUT, f'{tag}_replay_kernel_stats.csv'))
    line = last_json(os.path.join(G, f'{tag}_line_smnist.log'))
    json.dump(line, open(os.path.join(OUT, f'{tag}_bench.json'), 'w'), indent=1)
    lines = {w: last_json(os.path.join(G, f'{tag}_line_{w}.log')) for w in SECONDARY
             if os.path.exists(os.path.join(G, f'{tag}_line_{w}.log'))}
    if not lines:        # from round 3 on the default line carries them itself
        lines = (line or {}).get('secondary', {})
    json.dump(lines, open(os.path.join(OUT, f'{tag}_lines.json'), 'w'), indent=1)

    traffic, mfma = {}, {}
    for name, (prefix, algo, flops) in KERNELS.items():
        fr, wr, sq = (pmc_rows(src(d, 'p_counter_collection.csv'), prefix) for d in ('fetch', 'write', 'sq'))
        keep([r for r in fr if r['Counter_Name'] == 'FETCH_SIZE'], f'{tag}_pmc_FETCH_SIZE_{name}.csv')
        keep([r for r in wr if r['Counter_Name'] == 'WRITE_SIZE'], f'{tag}_pmc_WRITE_SIZE_{name}.csv')
        keep(sq, f'{tag}_pmc_SQ_{name}.csv')
        f_kb, nf = avg(fr, 'FETCH_SIZE')
        w_kb, nw = avg(wr, 'WRITE_SIZE')
        if f_kb is not None and w_kb is not None:
            traffic[name] = dict(kernel=prefix, FETCH_SIZE_KB=f_kb, WRITE_SIZE_KB=w_kb, dispatches=[nf, nw], fetch_correction=2.0,
                                 note='gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, '
                                      'HBM / rocprofv3); WRITE_SIZE exact',
                                 traffic_bytes=1024.0 * (2.0 * f_kb + w_kb), algorithmic_bytes=algo)
        busy, n = avg(sq, 'SQ_VALU_MFMA_BUSY_CYCLES')
        gui, _ = avg(sq, 'GRBM_GUI_ACTIVE')
        if busy is not None and gui:
            wave, _ = avg(sq, 'SQ_WAVE_CYCLES')
            act, _ = avg(sq, 'SQ_ACTIVE_INST_ANY')
            wi, _ = avg(sq, 'SQ_WAIT_INST_ANY')
            wa, _ = avg(sq, 'SQ_WAIT_ANY')
            mops, _ = avg(sq, 'SQ_INSTS_VALU_MFMA_MOPS_F32')
            mfma[name] = dict(kernel=prefix, dispatches=n, SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=gui,
                              mfma_util=busy / (128.0 * gui), mfma_flops_executed=(mops * 512.0) if mops else None,
                              algorithmic_flops=flops,
                              wave_cycle_split=dict(active=act / wave, wait_inst=wi / wave, wait=wa / wave) if wave else None,
                              note='mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE): busy cycles summed over 1024 '
                                   'SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs; calibrated on the 4096^3 GEMM (0.90)')
    if os.path.exists(src('stfetch', 'p_counter_collection.csv')):
        for name, (prefix, grid, algo, flops) in STRESS_KERNELS.items():
            sel = lambda d: [r for r in pmc_rows(src(d, 'p_counter_collection.csv'), prefix) if int(r['Grid_Size']) == grid]
            fr, wr, sq = sel('stfetch'), sel('stwrite'), sel('stsq')
            f_kb, nf = avg(fr, 'FETCH_SIZE')
            w_kb, nw = avg(wr, 'WRITE_SIZE')
            if f_kb is not None and w_kb is not None:
                traffic[name] = dict(kernel=prefix, grid_threads=grid, FETCH_SIZE_KB=f_kb, WRITE_SIZE_KB=w_kb, dispatches=[nf, nw],
                                     fetch_correction=2.0, traffic_bytes=1024.0 * (2.0 * f_kb + w_kb), algorithmic_bytes=algo)
            busy, n = avg(sq, 'SQ_VALU_MFMA_BUSY_CYCLES')
            gui, _ = avg(sq, 'GRBM_GUI_ACTIVE')
            if busy is not None and gui:
                mops, _ = avg(sq, 'SQ_INSTS_VALU_MFMA_MOPS_F32')
                mfma[name] = dict(kernel=prefix, dispatches=n, mfma_util=busy / (128.0 * gui),
                                  mfma_flops_executed=(mops * 512.0) if mops else None, algorithmic_flops=flops)
    cal = os.path.join(G, f'{tag}_sqcal', 'p_counter_collection.csv')
    if os.path.exists(cal):
        rows = pmc_rows(cal, 'void vargp::gemm_kernel')
        busy, n = avg(rows, 'SQ_VALU_MFMA_BUSY_CYCLES')
        gui, _ = avg(rows, 'GRBM_GUI_ACTIVE')
        mfma['calibration_gemm_4096'] = dict(dispatches=n, mfma_util=busy / (128.0 * gui),
                                             timing=[l for l in open(os.path.join(G, f'{tag}_sqcal.log')).read().splitlines()
                                                     if l.startswith('gemm4k')])
    import subprocess
    sha = subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(['git', 'status', '--porcelain', '--', 'vargp_amd', 'bench.py'], cwd=ROOT, capture_output=True,
                                text=True).stdout.strip())
    traffic['commit'] = mfma['commit'] = sha + ('+dirty' if dirty else '')     # the tree the counters were collected on
    json.dump(traffic, open(os.path.join(OUT, f'{tag}_traffic.json'), 'w'), indent=1)
    json.dump(mfma, open(os.path.join(OUT, f'{tag}_mfma.json'), 'w'), indent=1)

    # per-step table from the eager run (no re-launches in it: --no-replay)
    rows = list(csv.DictReader(open(src('eager', 'p_kernel_stats.csv'))))
    steps = 50 + 5 + 1 + 1          # timed + warm-up + the recording step + the ELBO check
    out, tot, nl = [], 0.0, 0.0
    for r in rows:
        calls, a = int(r['Calls']), float(r['AverageNs']) / 1e3
        if calls < steps // 2:
            continue
        if r['Name'].startswith('Cijk_'):       # bench.py's device warm-up between set-up and the W warm-up steps (torch.mm on
            continue                            # scratch tensors, hipBLASLt): not part of a step
        per = calls / steps
        tot += calls * a / steps
        nl += per
        if len(out) < 26:
            out.append(f"| `{r['Name'][:78]}` | {per:.1f} | {a:.1f} | {calls * a / steps:.1f} |")
    with open(os.path.join(OUT, f'{tag}_top_kernels.md'), 'w') as g:
        g.write('| kernel | launches/step | avg µs | µs/step |\n|---|---|---|---|\n' + '\n'.join(out) + '\n\n')
        g.write(f'Sum of kernel time: {tot:.0f} µs per step over {nl:.0f} launches (eager run, {steps} program runs).\n')
    # the step under GRAPH REPLAY (the mode the metric is timed in), launch by launch: rocprofv3's dispatch-to-completion average
    # of a run without re-launches and without stamps (collect.sh: --no-replay --no-timeline), next to the in-step span / slot
    # the kernels stamp themselves (bench line: `timeline`), the algorithmic flop, and PMC traffic against algorithmic bytes
    grows = {r['Name']: r for r in csv.DictReader(open(src('graph', 'p_kernel_stats.csv')))}
    tl = {r['kernel']: r for r in (line or {}).get('timeline', [])}
    tab, tot_us = [], 0.0
    for name, (prefix, algo, _) in KERNELS.items():
        bare = prefix[5:] if prefix.startswith('void ') else prefix
        r = next((v for k, v in grows.items() if k.startswith(prefix) or k.startswith(bare)), None)
        if r is None:
            continue
        us = float(r['AverageNs']) / 1e3
        tot_us += us
        fl = STEP_FLOP.get(name, 0.0)
        t = tl.get(TIMELINE_NAME.get(name, name), {})
        tr = traffic.get(name, {}).get('traffic_bytes')
        tab.append((t.get('start_us', 1e9), f"| `{bare[7:45]}` | {us:.1f} | {t.get('span_us', float('nan')):.1f} | {t.get('slot_us', float('nan')):.1f} | "
                    f"{fl / 1e9:.2f} | {(fl / (us * 1e-6) / 1e12 / PEAK_TF) if fl else float('nan'):.3f} | "
                    f"{algo / 1e6:.1f} | {(tr / 1e6) if tr else float('nan'):.1f} | {(tr / algo) if tr else float('nan'):.2f} |"))
    with open(os.path.join(OUT, f'{tag}_top_kernels.md'), 'a') as g:
        g.write('\nThe same step under graph replay, in launch order (rocprofv3 average of `bench.py --no-replay --no-timeline`; span / slot: '
                'the in-step wall-clock stamps of the bench line, `timeline`):\n\n'
                '| kernel | rocprof µs | span µs | slot µs | algorithmic GFLOP | frac of f32-MFMA peak (rocprof µs) | algorithmic MB | '
                'PMC traffic MB | traffic / algorithmic |\n|---|---|---|---|---|---|---|---|---|\n')
        g.write('\n'.join(row for _, row in sorted(tab)) + '\n\n')
        g.write(f'Sum of the rocprofv3 averages: {tot_us:.1f} µs; step (bench line): {1e3 * (line or {}).get("ms_per_step", float("nan")):.1f} µs; '
                f'whole step {(line or {}).get("step_flop", 0) / 1e9:.2f} GFLOP = {(line or {}).get("step_frac", float("nan")):.3f} of the peak.\n')
    print(json.dumps(dict(traffic=traffic, mfma=mfma), indent=1)[:3000])
    print(open(os.path.join(OUT, f'{tag}_top_kernels.md')).read())
    for w, l in lines.items():
        print(w, l and l.get('value'))


if __name__ == '__main__':
    main()
