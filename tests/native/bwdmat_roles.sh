#!/bin/bash
# Per-role timing of the backward merged launch (VARGP_EXP_BWDMAT: 0 both roles, 1 matrix chains only, 2 P_uf tiles only; results of 1 / 2 are incomplete).  GPU box: bash tests/native/bwdmat_roles.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 0 1 2; do
  export VARGP_EXP_BWDMAT=$m
  bash $R/profiles/kstats.sh roles_$m --steps 20 2>&1 | grep -E "bwdmat|bwd_mid|puu_final" 
done
