#!/bin/bash
# per-kernel counters of one eager Cfg2 bench run: every kernel of the step, counter groups given as quoted arguments
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  rm -rf /tmp/pmk
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmk -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --eager --no-replay --steps 6 --warmup 2 > /tmp/pmk.log 2>&1
  f=$(find /tmp/pmk -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmk.log; continue; fi
  python3 - "$f" <<'PY'
import csv,sys,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('vargp::','')[:34]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    if any(t in k for t in ('t0_','chol_rbf','yogi','gemm_kernel')):
        print('%-36s' % k, '  '.join('%s=%d' % (c.replace('SQ_',''), sum(v)/len(v)) for c,v in sorted(d.items())), 'n=%d' % len(next(iter(d.values()))))
PY
done
