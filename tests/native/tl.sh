#!/bin/bash
# The Cfg2 bench line's per-launch time line under the given environment.  Usage: bash tests/native/tl.sh "<env set>" ...   ("-" = none)
R=${GRAFT_REPO_ROOT:-/root/repo}
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  echo "== [$E]"
  env $E python3 $R/bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 20 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.1f  ms %.4f  rtol %.2g' % (d['value'], d['ms_per_step'], d.get('elbo_rtol_vs_cpu') or -1))
for t in d.get('timeline') or []:
    print('  %-16s span %6.2f gap %5.2f slot %6.2f  %s' % (t['kernel'], t['span_us'], t['gap_before_us'], t['slot_us'], {k: v for k, v in t.items() if k.endswith('_end_us') or k.endswith('start_us') and k != 'start_us'}))"
done
