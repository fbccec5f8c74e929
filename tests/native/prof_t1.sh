#!/bin/bash
# rocprofv3 per-kernel statistics of the later-task workloads (block program under hipGraph replay).
# Run on the GPU box from the repo root: bash tests/native/prof_t1.sh [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
for W in ${@:-smnist_t1 pmnist_t1}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1_$W -o p -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-timeline --no-replay --steps 200 --warmup 20 > $OUT/t1_$W.log 2>&1
  tail -1 $OUT/t1_$W.log | cut -c1-300
  f=$(find $OUT/t1_$W -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total us', tot/1e3)
for r in rows[:40]:
    print('%-90s calls %6s avg %9.1f us  %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
done
