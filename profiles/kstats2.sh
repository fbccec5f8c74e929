#!/bin/bash
# as kstats.sh with a step count: `bash profiles/kstats2.sh <tag> <steps> <warmup> [bench args]`; prints launches per step too
TAG=${1:-ks}; ST=${2:-6}; WU=${3:-2}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --eager --no-replay --steps $ST --warmup $WU "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" $((ST+WU+1)) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2])
tot=0
for r in rows[:34]:
    per=int(r['Calls'])/n
    us=float(r['AverageNs'])/1e3
    tot+=per*us
    print(r['Name'][:86].ljust(86), '%6.1f/step'%per, '%9.1f us'%us, '%9.1f us/step'%(per*us))
print('listed total us/step: %.0f' % tot)
PY
