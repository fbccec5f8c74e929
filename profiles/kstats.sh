#!/bin/bash
# Per-kernel average durations of one bench.py invocation (eager, no replays): `bash profiles/kstats.sh <tag> [bench args]`
# on the GPU box.  Writes gpurun_out/<tag>/ (scratch) and prints the top of the stats table.
TAG=${1:-ks}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --eager --no-replay --steps 30 --warmup 5 "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:32]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), '%8.1f'%(float(r['AverageNs'])/1e3))
PY
