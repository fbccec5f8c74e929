#!/bin/bash
# per-phase cycles of t0_bwd_mid_kernel / t0_puu_final_kernel (workgroup 0) in -DBM_STAMPS / -DTAIL_STAMPS builds (GPU box)
cd vargp_amd/csrc
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DBM_STAMPS -DTAIL_STAMPS -c elbo_t0.hip -o /tmp/t0_bm.o 2>&1 | grep -E "error"
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvargp_bm.so build/core.o build/gemm.o build/rbf.o build/chol.o build/elbo_ops.o /tmp/t0_bm.o build/elbo_tn.o
cd ../..
echo "== t0_bwd_mid"; VARGP_HIP_LIB=/tmp/libvargp_bm.so python tests/native/bm_stamps.py 2>&1 | tail -14
echo "== t0_puu_final"; VARGP_HIP_LIB=/tmp/libvargp_bm.so python tests/native/bm_stamps.py tail 2>&1 | tail -10
