cd tests/native && make bench_kernels >/dev/null 2>&1; cd ../..
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oiE "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_SMEM|WAIT_IFETCH|INST_CYCLES)[A-Z_0-9]*)\b" | sort -u | tr '\n' ' '; echo
for mode in 0 1; do
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH" ; do
    rm -rf /tmp/pm; VARGP_CHOL_F32_ALONE=$mode rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -o p -- $R/tests/native/bench_kernels chol 20 > /tmp/pm.log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    python3 - "$f" "f32=$mode" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'small3' in r['Kernel_Name'] and '25' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('no csv', e); print(open('/tmp/pm.log').read()[-1500:])
print(sys.argv[2], {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'n=', [len(v) for v in acc.values()][:1])
PY
  done
done
