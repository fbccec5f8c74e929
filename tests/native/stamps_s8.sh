#!/bin/bash
# per-phase cycles of t0_bwd_mid_kernel and t0_fwd_fused_kernel on the THROUGHPUT-bound shape (S = 8: 640 tile workgroups on 256 CUs),
# stamped by a workgroup of the first round (0) and one of the last (600)
cd vargp_amd/csrc
for blk in 0 600; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DBM_STAMPS -DFF_STAMPS -DBM_STAMP_BLOCK=${blk}u -DFF_STAMP_BLOCK=${blk}u -c elbo_t0.hip -o /tmp/t0_st.o 2>&1 | grep -E "error"
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvargp_st.so build/core.o build/gemm.o build/rbf.o build/chol.o build/elbo_ops.o /tmp/t0_st.o build/elbo_tn.o
  for S in 3 8; do
    [ $S = 3 ] && [ $blk = 600 ] && continue
    echo "== S=$S block $blk: t0_bwd_mid"; (cd ../..; VARGP_BM_S=$S VARGP_HIP_LIB=/tmp/libvargp_st.so python tests/native/bm_stamps.py 2>&1 | tail -14)
    echo "== S=$S block $blk: t0_fwd_fused"; (cd ../..; VARGP_BM_S=$S VARGP_HIP_LIB=/tmp/libvargp_st.so python tests/native/bm_stamps.py ff 2>&1 | tail -16)
  done
done
