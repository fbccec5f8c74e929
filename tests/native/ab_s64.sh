#!/bin/bash
# A/B of the first-task program at many hyper-samples: the LDS-resident route (VARGP_T0_UNITS tile units allowed) against the
# default routing.  Run on the GPU box from the repo root: bash tests/native/ab_s64.sh [workload ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in ${@:-smnist_s32 smnist_s64}; do
  for U in 2048 16384; do
    echo "== $W VARGP_T0_UNITS=$U"
    VARGP_T0_UNITS=$U python3 $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-timeline --no-replay --steps 60 --warmup 10 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','elbo_rtol_vs_cpu','finite','cholesky_failures','error')})"
  done
done
