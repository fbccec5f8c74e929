#!/bin/bash
# A/B of the BASELINE config-2 line under environment switches: bash tests/native/ab_cfg2.sh VAR=VALUE [VAR=VALUE ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for mode in base "$@"; do
    if [ "$mode" = base ]; then E=""; else E="$mode"; fi
    env $E python3 $R/bench.py --no-cpu-baseline --no-secondary --no-timeline --steps 400 --warmup 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', '$mode', round(d['value'],1), d['elbo_rtol_vs_cpu'])"
  done
done
