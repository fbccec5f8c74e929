"""profiles/<tag>_secondary_kernels.md from the per-workload rocprofv3 statistics profiles/<tag>_<workload>_kernel_stats.csv
(`bash tests/native/prof_t1.sh smnist_t1 pmnist_t1 smnist_s64 smnist_s8` on the GPU box, then the csv files copied here).

    python profiles/secondary_md.py r06
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
WORK = [('smnist_t1', 'Split-MNIST t = 1 (S3 C10 M100, Mt = 200; block program)'),
        ('pmnist_t1', 'Permuted-MNIST t = 1 (S10 C10 M200, Mt = 400; block program)'),
        ('smnist_s64', 'BASELINE config 4 on one GPU (S64 C10 M100; first-task program, launches apart, Gram built by the chain workgroups)'),
        ('smnist_s8', "one rank's share of config 4 on 8 GPUs (S8; first-task program, launches apart)")]
SKIP = ('at::native', '__amd_rocclr', 'Cijk_')


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.replace('vargp::', '')
    m = re.match(r'([A-Za-z0-9_]+(<[^(]*>)?)\(', name)
    return m.group(1) if m else name[:80]


out = ['# Per-kernel rocprofv3 statistics of the secondary workloads (%s)' % tag, '',
       '`bash tests/native/prof_t1.sh smnist_t1 pmnist_t1 smnist_s64 smnist_s8` (`bench.py --workload W --no-replay --steps 200 --warmup 20` under',
       '`rocprofv3 --kernel-trace --stats`; hipGraph replay).  µs/step = calls × average ÷ program runs (the call count of the optimiser',
       'kernel); kernels of the set-up (`at::native::*`, the pre-heat products `Cijk_*`, copies) are left out.  Launches on a side stream',
       '(many hyper-samples: chains beside the product) overlap, so there the column adds up to MORE than the step.', '']
for w, desc in WORK:
    f = os.path.join(ROOT, '%s_%s_kernel_stats.csv' % (tag, w))
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if not any(s in r['Name'] for s in SKIP)]
    runs = max(int(r['Calls']) for r in rows if 'yogi_multi' in r['Name'])
    tab = []
    for r in rows:
        per = int(r['Calls']) / runs
        if per < 0.5:
            continue
        avg = float(r['AverageNs']) / 1e3
        tab.append((short(r['Name']), per, avg, per * avg))
    tab.sort(key=lambda t: -t[3])
    tot = sum(t[3] for t in tab)
    out += ['## %s: %s' % (w, desc), '', '%d launches, %.0f µs of kernel time per step.' % (round(sum(t[1] for t in tab)), tot), '',
            '| kernel | launches / step | average µs | µs / step |', '|---|---|---|---|']
    out += ['| `%s` | %.0f | %.1f | %.1f |' % t for t in tab]
    out.append('')
open(os.path.join(ROOT, '%s_secondary_kernels.md' % tag), 'w').write('\n'.join(out))
print('\n'.join(out[:60]))
