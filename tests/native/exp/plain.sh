#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export VARGP_HIP_LIB=$R/tests/native/exp/libvargp_plain.so
bash $R/profiles/kstats.sh plain --steps 20 2>&1 | grep -E "bwd_mid|bwdmat"
