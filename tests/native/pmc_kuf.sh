#!/bin/bash
# PMC passes over the K_uf GEMM role alone (VARGP_EXP_MERGED=2) at S = 8, eager: where do the non-MFMA cycles of the slab loop go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_kuf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
export VARGP_EXP_MERGED=2
B="python3 $R/bench.py --workload smnist_s8 --no-cpu-baseline --no-secondary --eager --no-replay --no-timeline --steps 6 --warmup 2"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TA_TCP_STATE_READ"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- $B > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  echo "== pass $i: $set"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0.0,0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_rbf_gemm' in r['Kernel_Name']:
        a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,(v,n) in acc.items(): print('   %-36s %16.0f per launch (%d launches)' % (k, v/n, n))
PY
  [ -z "$f" ] && tail -3 $OUT/p$i.log
done
grep -i -E "LDS|MFMA" $OUT/counters.txt | grep -i -o -E "\b(SQ|TCP|TCC)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' | cut -c1-3000
