#!/bin/bash
# env-knob experiments on the two GEMM-carrying launches at S = 8 (spans from the in-graph stamps)
run() { echo "== $1"; env $1 python bench.py --workload smnist_s8 --steps 100 --warmup 20 --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  steps/s %.1f  ms %.4f' % (d['value'], d['ms_per_step']))
for t in d.get('timeline',[]):
    if t['kernel'] in ('chol_rbf_gemm','t0_bwdmat_gemm','t0_pro_kuu','t0_puu_final'): print('   %-16s span %7.2f slot %7.2f' % (t['kernel'], t['span_us'], t['slot_us']), {k:round(v,1) for k,v in t.items() if k.endswith('end_us')})
"; }
run "X=1"
run "VARGP_MERGED_PAD=0"
run "VARGP_EXP_MERGED=2"
run "VARGP_EXP_MERGED=2 VARGP_MERGED_PAD=0"
run "VARGP_EXP_BWDMAT=2"
run "VARGP_GEMM_PERSIST=0"
run "VARGP_GEMM_PERSIST=3"
run "VARGP_MERGED_TILE=3"
