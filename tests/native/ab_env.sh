#!/bin/bash
# A/B of tuning switches on bench workloads.  Usage: bash tests/native/ab_env.sh "<workloads>" "<env set 1>" "<env set 2>" ...
# (an env set is a space-separated list of VAR=value; "-" = no variables).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=$1; shift
for W in $WL; do
  for E in "$@"; do
    [ "$E" = "-" ] && E=""
    echo -n "== $W [$E] "
    env $E python3 $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-timeline --no-replay --steps ${STEPS:-100} --warmup 10 2>&1 | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print({k:d.get(k) for k in ('value','ms_per_step','elbo_rtol_vs_cpu','error') if d.get(k) is not None})
except Exception as e: print('FAILED', e)"
  done
done
