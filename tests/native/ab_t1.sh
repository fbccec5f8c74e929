#!/bin/bash
# A/B of the later-task workloads under an environment switch: bash tests/native/ab_t1.sh VAR=VALUE [workload ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
SW=$1; shift
for W in ${@:-smnist_t1 pmnist_t1}; do
  for rep in 1 2; do
    for mode in base "$SW"; do
      if [ "$mode" = base ]; then E=""; else E="$SW"; fi
      env $E python3 $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-timeline --steps 300 --warmup 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W', '$mode', round(d['value'],1), d['elbo_rtol_vs_cpu'])"
    done
  done
done
