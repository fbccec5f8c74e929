"""Turn the raw rocprofv3 output of one round (under gpurun_out/, scratch) into the committed summaries here.

Commands that produce the inputs (run from /tmp on the GPU box, R = repo root; separate passes for the two PMC
counters, no other trace domain with --pmc):
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_graph -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_eager -- python $R/bench.py --eager --steps 50 --warmup 5 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- python $R/bench.py --eager --steps 20 --warmup 3 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- python $R/bench.py --eager --steps 20 --warmup 3 --no-cpu-baseline
Usage: python profiles/summarize.py r01 [bench-line.json]
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles')
# kernels whose HBM traffic bench.py reports: tag (bench.py) -> (kernel-name prefix, algorithmic bytes per launch)
S, C, M, D, B = 3, 10, 100, 784, 512
NR = (4 + 2 * M + 3) // 4 * 4
LD = (NR + B + 3) // 4 * 4
KERNELS = {
    # factorisations: read K_uu/S_u, write L and T ((S*C + C) matrices); GEMM: read z and x once, write K_uf
    'chol_rbf_gemm': ('void vargp::chol_rbf_gemm_kernel',
                      4 * (3 * (S * C + C) * M * M + C * M * D + B * D + S * C * M * B)),
    # W.Y products: read W_uf (S*C*M*B), W_uu (S*C*M*M), x, z once; write P_uf, P_uu (S*C*M*D each)
    'rbf_kuu_bwd_gemm': ('void vargp::gemm_pair_kernel<64, 64, 64, true, false, true, false>',
                         4 * (S * C * M * B + S * C * M * M + B * D + C * M * D + 2 * S * C * M * D)),
}


def one(pattern):
    files = glob.glob(os.path.join(ROOT, 'gpurun_out', pattern))
    assert files, pattern
    return max(files, key=os.path.getmtime)


def stats_rows(path):
    return list(csv.DictReader(open(path)))


def pmc_avg(path, prefix, counter):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
            if r['Kernel_Name'].startswith(prefix) and r['Counter_Name'] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    graph, eager = one('p_graph/*/*kernel_stats.csv'), one('p_eager/*/*kernel_stats.csv')
    shutil.copy(graph, os.path.join(OUT, f'{tag}_bench_graph_kernel_stats.csv'))
    shutil.copy(eager, os.path.join(OUT, f'{tag}_bench_eager_kernel_stats.csv'))
    fetch, write = one('p_fetch/*/*counter_collection.csv'), one('p_write/*/*counter_collection.csv')
    traffic = {}
    for name, (prefix, algo) in KERNELS.items():
        f_kb, nf = pmc_avg(fetch, prefix, 'FETCH_SIZE')
        w_kb, nw = pmc_avg(write, prefix, 'WRITE_SIZE')
        # keep the rows of this kernel as evidence
        for src, cname in ((fetch, 'FETCH_SIZE'), (write, 'WRITE_SIZE')):
            rows = [r for r in csv.DictReader(open(src)) if r['Kernel_Name'].startswith(prefix)]
            with open(os.path.join(OUT, f'{tag}_pmc_{cname}_{name}.csv'), 'w', newline='') as g:
                wtr = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
                wtr.writeheader()
                wtr.writerows(rows)
        traffic[name] = dict(kernel=prefix, FETCH_SIZE_KB=f_kb, WRITE_SIZE_KB=w_kb, dispatches=[nf, nw],
                             fetch_correction=2.0,
                             note='gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads '
                                  '(MI355X_MICROARCH.md, HBM / rocprofv3); WRITE_SIZE exact',
                             traffic_bytes=1024.0 * (2.0 * f_kb + w_kb), algorithmic_bytes=algo)
    with open(os.path.join(OUT, f'{tag}_traffic.json'), 'w') as g:
        json.dump(traffic, g, indent=1)
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[2], os.path.join(OUT, f'{tag}_bench.json'))

    # README table from the eager run (exact launches per step)
    rows = stats_rows(eager)
    steps = 50 + 5 + 1          # timed + warm-up + the recording step of bench.py
    lines = []
    tot = 0.0
    nlaunch = 0.0
    # bench.py re-launches three recorded kernels 3 + 100 times each for its live timing: not part of a step
    replayed = ('void vargp::chol_rbf_gemm_kernel', 'void vargp::gemm_pair_kernel<64, 64, 64, true, false, true, false>',
                'void vargp::gemm_kernel<64, 64, 64, true, true, true, true>')
    for r in rows:
        calls, avg = int(r['Calls']), float(r['AverageNs']) / 1e3
        if r['Name'].startswith(replayed):
            calls -= 103
        if calls < steps // 2:
            continue
        per = calls / steps
        tot += calls * avg / steps
        nlaunch += per
        if len(lines) < 24:
            lines.append(f"| `{r['Name'][:76]}` | {per:.1f} | {avg:.1f} | {calls * avg / steps:.1f} |")
    with open(os.path.join(OUT, f'{tag}_top_kernels.md'), 'w') as g:
        g.write('| kernel | launches/step | avg µs | µs/step |\n|---|---|---|---|\n' + '\n'.join(lines) + '\n\n')
        g.write(f'Sum of kernel time: {tot:.0f} µs per step over {nlaunch:.0f} launches (eager run, {steps} steps; the 103 '
                f're-launches per timed kernel of bench.py\'s live measurement are subtracted).\n')
    print(json.dumps(traffic, indent=1))
    print(open(os.path.join(OUT, f'{tag}_top_kernels.md')).read())


if __name__ == '__main__':
    main()
