"""(Re)write the round-6 section at the top of profiles/README.md from profiles/r06_bench.json and the graph-replay statistics.
    python profiles/make_readme_r06.py        (after summarize.py r06 and secondary_md.py r06)"""
import csv
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
d = json.load(open(f'{HERE}/r06_bench.json'))
tl = d['timeline']
clk = ' · '.join('%s %.2f' % (t['kernel'], t['clock_ghz']) for t in tl if t.get('clock_ghz'))
sec = d['secondary']
v = lambda k: sec[k]['value']
st = sec['stress']
g = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(f'{HERE}/r06_bench_graph_kernel_stats.csv'))}
bm = [x for k, x in g.items() if 't0_bwdmat_gemm_kernel' in k][0]
rf = d['roofline']
new = f"""# profiles — round 6 (MI355X, ROCm 7.2)

Raw inputs: `bash profiles/collect.sh r06` on the GPU box; summaries: `python profiles/summarize.py r06`, `python profiles/secondary_md.py r06`,
this section: `python profiles/make_readme_r06.py` (`gpurun_out/` is scratch).  File set as in round 5 (`r06_bench.json`, `r06_lines.json`,
`r06_bench_{{graph,eager}}_kernel_stats.csv`, `r06_replay_kernel_stats.csv`, `r06_top_kernels.md`, `r06_pmc_*`, `r06_traffic.json`, `r06_mfma.json`,
`r06_{{smnist_t1,pmnist_t1,smnist_s64,smnist_s8}}_kernel_stats.csv` + `r06_secondary_kernels.md`, `r06_chol2048_kernel_stats.csv`), plus
`r06_hotS.txt`: the two big products of the first-task step as plain batched GEMMs at S = 8 / 16 / 64, every tile shape
(`tests/native/bench_kernels hotS`) — what the role-merged launches were measured against before they were taken apart for many hyper-samples.

Bench line: **{d['value']:.0f} ELBO steps/s** ({d['ms_per_step']:.4f} ms/step, 10 steps per graph launch; round 5: 5118), `step_frac` {d['step_frac']:.3f}, ELBO rtol vs CPU oracle
{d['elbo_rtol_vs_cpu']:.1e}, CPU baseline {d['cpu_baseline']['value']:.2f} steps/s on 32 threads.  `roofline` = `t0_bwdmat_gemm_kernel`: slot {rf['avg_us']:.1f} µs, span {rf['span_us']:.1f},
`frac` {rf['frac']:.3f} (span {rf['frac_span']:.3f}, isolated {rf['frac_isolated']:.3f}); rocprofv3 graph-replay average of the same kernel {bm:.1f} µs
(`r06_bench_graph_kernel_stats.csv`) → {2.408448e9 / (bm * 1e-6) / 1e12 / 157.3:.3f}.  Per launch: `r06_top_kernels.md` — against round 5 the front launch 21.8 → 19.0 µs and the
factorisation ‖ K_uf launch 41.5 → 37.1 µs (blocked pivot chain on the matrix core, its zero-fill role sized to the spare CUs, S_u built by
its chain workgroup: DESIGN §5).

**The clock is measured** (round-5 review, item 3).  `vargp_prof_spans` mode 3: workgroup 0 of every stamped launch reads the shader clock
(`s_memtime`) beside the 100 MHz wall clock at its start and at its end; the ratio of the two differences is the clock the chip held
while that workgroup ran, inside the replayed hipGraph.  GHz per launch of the Cfg2 step: {clk}.
`roofline.clock_ghz` = {rf['clock_ghz']:.2f} for the dominant launch, so `frac_at_held_clock` = {rf['frac_at_held_clock']:.3f} against `frac` {rf['frac']:.3f}: the chip does not clock
down under this step, and the "≈ 1.65 GHz" of the round-4 / round-5 texts was a wrong inference (a chain's cycle count from a stand-alone
run set against its duration inside the merged launch).  The GEMM roles of the two merged launches therefore run at 0.52 / 0.46 of
the peak alone, not at ≈ 0.7 of a lower roof — the gap is the one-workgroup-per-CU occupancy the chain role's registers force on the
whole kernel (DESIGN §5, §11).

Secondary (steps/s unless noted; round 5 in brackets): drop-in loop {v('smnist_dropin'):.0f} (raise) / {sec['smnist_dropin']['value_lazy']:.0f} (lazy) / {sec['smnist_dropin']['value_defer']:.0f} (defer) ·
driver-style epochs {v('smnist_epochs'):.0f} · **smnist_s64 {v('smnist_s64'):.0f} [307]** · smnist_s32 {v('smnist_s32'):.0f} [579] · smnist_s16 {v('smnist_s16'):.0f} [1238] · smnist_s8 {v('smnist_s8'):.0f} [2165] ·
smnist_t1 {v('smnist_t1'):.0f} [1770] · pmnist_t0 {v('pmnist_t0'):.0f} [633] · pmnist_t1 {v('pmnist_t1'):.0f} [289] · pmnist_t4 {v('pmnist_t4'):.1f} [74.0] · pmnist_t9 {v('pmnist_t9'):.1f} [20.5] ·
stress {st['value'] / 1e3:.0f}k points/s [235k] (n = 2048 × 10 factorisation + inverse 2.20 ms = 0.25 of peak, unchanged).
The 64-sample step is 1.92e11 flop in {1e3 / v('smnist_s64'):.2f} ms = **{1.92e11 / (1 / v('smnist_s64')) / 1e12 / 157.3:.2f} of the f32-MFMA peak** [0.375]; kernel by kernel in `r06_secondary_kernels.md`
(its chain launches run on a side stream beside the K_uf / P_uf products, so their rocprofv3 durations overlap and add up to more
than the step).

---

"""
p = f'{HERE}/README.md'
s = open(p).read()
i0, i1 = s.index('# profiles — round 6'), s.index('# profiles — round 5')
open(p, 'w').write(s[:i0] + new + s[i1:])
print(new[:1500])
