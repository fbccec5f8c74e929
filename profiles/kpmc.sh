#!/bin/bash
# SQ counters of one kernel (name substring $2) in an eager bench.py run: `bash profiles/kpmc.sh <tag> <kernel> [bench args]`.
TAG=${1:-pm}; KN=${2:-t0_bwd_mid}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --eager --no-replay --steps 6 --warmup 2 "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(find $R/gpurun_out/$TAG -name "*counter_collection.csv" | head -1)
python3 - "$f" "$KN" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print('%-28s %14.0f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
