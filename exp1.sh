#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { echo "== $W $*"; env "$@" python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('elbo_rtol_vs_cpu'))"; }
for W in pmnist_t1 pmnist_t4; do
  b VARGP_X=0
  b VARGP_TN_WYTILE=1
  b VARGP_TN_WYTILE=10
  b VARGP_TN_WYTILE=11
done
