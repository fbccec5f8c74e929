#!/bin/bash
# Per-phase cycle accounting of the n > 50 Cholesky kernel (chol_small3.h): builds chol.hip with -DVARGP_CHOL_STAMPS
# into a scratch copy of the library and runs the micro-benchmark against it.  Run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
make -s -C tests/native
d=/tmp/chol_stamps; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DVARGP_CHOL_STAMPS -c vargp_amd/csrc/chol.hip -o $d/chol.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libvargp_hip.so $d/chol.o vargp_amd/csrc/build/core.o \
    vargp_amd/csrc/build/gemm.o vargp_amd/csrc/build/rbf.o vargp_amd/csrc/build/elbo_ops.o vargp_amd/csrc/build/elbo_t0.o vargp_amd/csrc/build/elbo_tn.o
LD_LIBRARY_PATH=$d tests/native/bench_kernels chol 50
