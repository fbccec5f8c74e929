#!/bin/bash
# timing ablations of the small Cholesky kernel (see VARGP_CHOL_ABL in csrc/chol.hip); run on the GPU box
set -e
cd "$(dirname "$0")/../.."
make -s -C tests/native
for abl in ${ABLS:-0 1 2 4}; do
  d=/tmp/abl$abl; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DVARGP_CHOL_ABL=$abl $EXTRA -c vargp_amd/csrc/chol.hip -o $d/chol.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libvargp_hip.so $d/chol.o vargp_amd/csrc/build/core.o \
      vargp_amd/csrc/build/gemm.o vargp_amd/csrc/build/rbf.o vargp_amd/csrc/build/elbo_ops.o vargp_amd/csrc/build/elbo_t0.o
  echo "ABL=$abl"; LD_LIBRARY_PATH=$d tests/native/bench_kernels chol 50 | tail -1
done
