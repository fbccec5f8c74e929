#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export VARGP_EXP_BWDMAT=1
for v in "" NO_GLU; do
  if [ -n "$v" ]; then export VARGP_HIP_LIB=$R/tests/native/exp/libvargp_$v.so; fi
  echo "variant=$v"
  bash $R/profiles/kstats.sh noglu_$v --steps 20 2>&1 | grep -E "bwdmat"
done
