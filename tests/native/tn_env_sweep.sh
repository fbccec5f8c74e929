R=${GRAFT_REPO_ROOT:-/root/repo}
for W in pmnist_t1 smnist_t1; do
for E in "X=1" "VARGP_TN_PAIRFWD=0" "VARGP_TN_PAIRBWD=0" "VARGP_TN_KSYM=0" "VARGP_GEMM_TRIC64=0" "VARGP_CHOL_F32=0"; do
  env $E python3 $R/bench.py --workload $W --no-cpu-baseline --no-secondary --no-timeline --steps 300 --warmup 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W', '$E', round(d['value'],1))"
done; done
