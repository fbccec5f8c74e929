#!/bin/bash
# counters of one kernel (name substring $1) in an eager bench.py run; counter groups from $2 on (quoted strings)
KN=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  rm -rf /tmp/pmk
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmk -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --eager --no-replay --steps 6 --warmup 2 > /tmp/pmk.log 2>&1
  f=$(find /tmp/pmk -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KN" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k,v in acc.items()}, 'launches', [len(v) for v in acc.values()][:1])
PY
done
