#!/bin/bash
# per-phase cycles of the multi-tile t0_bwd_mid_multi_kernel at S = 8 (second tile of workgroup 100)
cd vargp_amd/csrc
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DBM_STAMPS -DBM_STAMP_BLOCK=100u -c elbo_t0.hip -o /tmp/t0_st.o 2>&1 | grep -E "error"
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvargp_st.so build/core.o build/gemm.o build/rbf.o build/chol.o build/elbo_ops.o /tmp/t0_st.o build/elbo_tn.o
cd ../..
echo "== S=8 multi-tile t0_bwd_mid"; VARGP_BM_MULTI=1 VARGP_BM_S=8 VARGP_HIP_LIB=/tmp/libvargp_st.so python tests/native/bm_stamps.py 2>&1 | tail -24
