"""Prepend the round's summary (bench line, per-kernel table from the committed rocprofv3 summaries) to profiles/README.md:
    python profiles/make_readme.py r04
Run after `python profiles/summarize.py r04`.  Idempotent: an existing section of the same round is replaced."""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
b = json.load(open(f'{HERE}/{tag}_bench.json'))
m = json.load(open(f'{HERE}/{tag}_mfma.json'))
t = json.load(open(f'{HERE}/{tag}_traffic.json'))
g = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(f'{HERE}/{tag}_bench_graph_kernel_stats.csv'))}
e = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(f'{HERE}/{tag}_bench_eager_kernel_stats.csv'))}
top = open(f'{HERE}/{tag}_top_kernels.md').read().strip().splitlines()[-1]


def avg(d, sub):
    v = [x for k, x in d.items() if sub in k]
    return v[0] if v else float('nan')


ROWS = [('chol_rbf_gemm', 'chol_rbf_gemm_kernel', 'roofline', 2.408e9,
         '`chol_rbf_gemm_kernel<25,2,128,32,false,float,512>` — factorisations of K_uu+εI (30, from K-split partial Gram matrices) and '
         'S_u+εI (10): fp32 chains, four pivots per barrier ‖ K_uf distance GEMM (x pre-scaled by the norm role, eight waves per 128×64 tile)'),
        ('rbf_kuf_bwd_gemm', 't0_bwdmat_gemm_kernel', 'roofline_gemm', 2.408e9,
         '`t0_bwdmat_gemm_kernel` — per-matrix adjoint chains (40 workgroups) ‖ P_uf = W_uf·x (216 persistent workgroups, three tiles each)'),
        ('t0_bwd_mid', 't0_bwd_mid_kernel', None, 1.23e9, '`t0_bwd_mid_kernel` — backward middle per (s, c, 64-column tile), incl. the likelihood of the tile'),
        ('t0_fwd_fused', 't0_fwd_fused_kernel', None, 0.61e9, '`t0_fwd_fused_kernel` — forward middle per (s, c, 64-column tile)'),
        ('t0_pro_kuu', 't0_pro_kuu_kernel', None, 0.47e9, '`t0_pro_kuu_kernel<32>` — prologue ‖ row norms (+ x∘σ⁻²) ‖ K-split K_uu inner products'),
        ('t0_puu_final', 't0_puu_final_kernel', None, 0.47e9, '`t0_puu_final_kernel<3>` — P_uu = W_uu·z + finalisation (ḡz, ḡθ) + packed-vector gradient')]
out = f'''# profiles — round {int(tag[1:])} (MI355X, ROCm 7.2)

Raw inputs: `bash profiles/collect.sh {tag}` on the GPU box (rocprofv3 runs of `bench.py`, one counter pass per PMC group, the
default bench line with its `secondary` workloads); summaries: `python profiles/summarize.py {tag}`, this section:
`python profiles/make_readme.py {tag}` (`gpurun_out/` is scratch).  Files of earlier rounds are kept for comparison; their text
follows below.  `{tag}_traffic.json` / `{tag}_mfma.json` carry `commit` = the tree the counters were collected on
(`{t.get('commit', '?')[:12]}`); `bench.py` reports it as `roofline.counters_commit`.  File set as in round 3: `{tag}_bench.json`,
`{tag}_lines.json`, `{tag}_bench_{{graph,eager}}_kernel_stats.csv`, `{tag}_replay_kernel_stats.csv`, `{tag}_top_kernels.md`,
`{tag}_pmc_*`, `{tag}_traffic.json`, `{tag}_mfma.json`.

Bench line: **{b['value']:.0f} ELBO steps/s** ({b['ms_per_step']:.4f} ms/step), ELBO rtol vs CPU oracle {b['elbo_rtol_vs_cpu']:.1e},
CPU baseline {b['cpu_baseline']['value']:.1f} steps/s on {b['cpu_baseline']['cores']} threads (port / reference = {b['cpu_baseline'].get('port_vs_reference') or float('nan'):.2f}).
{top}

| kernel | bench (live, back to back) | rocprof avg (eager / graph) | flop / launch | achieved (on the live or graph time) | MFMA util (PMC) | wave cycles active / issue-stalled / parked | traffic (PMC) vs algorithmic |
|---|---|---|---|---|---|---|---|
'''
for tagk, sub, rl, fl, desc in ROWS:
    ea, gr = avg(e, sub), avg(g, sub)
    live = b[rl]['avg_us'] if rl and b.get(rl) else None
    us = live if live else gr
    w = m[tagk]['wave_cycle_split']
    out += (f"| {desc} | {('%.1f µs' % live) if live else '–'} | {ea:.1f} / {gr:.1f} µs | {fl:.3g} | {fl / us / 1e6:.1f} TFLOP/s = "
            f"{fl / us / 1e6 / 157.3:.2f} | {m[tagk]['mfma_util']:.2f} | {100 * w['active']:.0f} % / {100 * w['wait_inst']:.0f} % / "
            f"{100 * w['wait']:.0f} % | {t[tagk]['traffic_bytes'] / 1e6:.1f} MB vs {t[tagk]['algorithmic_bytes'] / 1e6:.1f} MB |\n")
sec = b.get('secondary', {})
if sec:
    out += '\nSecondary (steps/s unless noted, all in the default line): ' + ' · '.join(
        '%s %.0f' % (k, v['value']) for k, v in sec.items() if 'value' in v and k not in ('stress', 'smnist_dropin'))
    if 'stress' in sec and 'value' in sec['stress']:
        st = sec['stress']
        out += (' · stress %.0fk points/s (n = 2048 × 10 factorisation + inverse %.2f ms = %.2f of peak; K_uf tile GEMM %.2f of peak)'
                % (st['value'] / 1e3, st['roofline_chol']['avg_us'] / 1e3, st['roofline_chol']['frac'], st['roofline']['frac']))
    if 'smnist_dropin' in sec and 'value' in sec['smnist_dropin']:
        out += ' · drop-in loop %.0f (raise) / %.0f (defer) steps/s' % (sec['smnist_dropin']['value'], sec['smnist_dropin']['value_defer'])
    out += '.\n'
out += 'MFMA-utilisation formula calibrated on the 4096³ GEMM: %.2f.\n\n---\n\n' % m['calibration_gemm_4096']['mfma_util']
path = f'{HERE}/README.md'
old = open(path).read()
marker = f'# profiles — round {int(tag[1:])} '
if old.startswith(marker):
    old = old[old.index('\n---\n\n') + 6:]
open(path, 'w').write(out + old)
print(out)
