#!/bin/bash
# Compile-time parameter sweep of the first-task program's glue kernels (run on the GPU box):
#   tests/native/sweep_t0.sh VARGP_W_ROWS 4 8 16
# builds elbo_t0.hip with -D<NAME>=<value> into a scratch copy of the library and runs bench.py against it.
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
for v in "$@"; do
  d=/tmp/sweep_${name}_$v; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -D$name=$v -c vargp_amd/csrc/elbo_t0.hip -o $d/elbo_t0.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libvargp_hip.so $d/elbo_t0.o vargp_amd/csrc/build/core.o \
      vargp_amd/csrc/build/gemm.o vargp_amd/csrc/build/rbf.o vargp_amd/csrc/build/elbo_ops.o vargp_amd/csrc/build/chol.o
  echo -n "$name=$v  "
  VARGP_HIP_LIB=$d/libvargp_hip.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"
done
