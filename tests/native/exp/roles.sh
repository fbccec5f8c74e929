#!/bin/bash
# per-role timing of the backward's merged launch (tuning builds / env switches; results of modes 1, 2 are incomplete)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 0 1 2; do
  export VARGP_EXP_BWDMAT=$m
  bash $R/profiles/kstats.sh roles_$m --steps 20 2>&1 | grep -E "bwdmat|bwd_mid|puu_final" 
done
