#!/bin/bash
# rocprofv3 average of the kernels matching $2 for workload $1 under the current environment: bash tests/native/prof_one.sh smnist_s64 puu_final
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/p1_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $T -o p -- python3 $R/bench.py --workload $1 --no-cpu-baseline --no-secondary --no-timeline --no-replay --steps 100 --warmup 10 > $T.log 2>&1
f=$(find $T -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$2" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2].lower() in r['Name'].lower():
        print('%-80s calls %6s avg %9.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3))
PY
rm -rf $T $T.log
