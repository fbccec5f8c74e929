#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 1 0; do
  export VARGP_GEMM_PERSIST=$m
  echo "persist=$m"
  bash $R/profiles/kstats.sh persist_$m --steps 20 2>&1 | grep -E "bwdmat|bwd_mid|puu_final|chol_rbf|fwd_fused|pro_kuu|yogi"
done
