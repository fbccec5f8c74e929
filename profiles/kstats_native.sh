#!/bin/bash
# Per-kernel durations of the stand-alone n = 2048 x 10 factorisation + inverse (tests/native/bench_kernels chol2048):
# `bash profiles/kstats_native.sh <tag> <bench_kernels args>` on the GPU box.
TAG=${1:-kn}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o p -- $R/tests/native/bench_kernels "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), '%9.1f'%(float(r['AverageNs'])/1e3), '%9.1f'%(float(r['TotalDurationNs'])/1e3))
PY
tail -3 $R/gpurun_out/$TAG.log
