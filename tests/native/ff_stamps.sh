#!/bin/bash
# per-phase cycles of t0_fwd_fused_kernel in a -DFF_STAMPS build (GPU box): workgroup 0 of the Cfg2 step, then -- on an 8-sample
# step (640 workgroups: 2.5 per CU) -- the first and one of the LAST workgroups (which finds the kernel's code in the I-cache)
cd vargp_amd/csrc
for blk in 0 600; do
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFF_STAMPS -DFF_STAMP_BLOCK=${blk}u -c elbo_t0.hip -o /tmp/t0_$blk.o 2>&1 | grep -E "error"
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvargp_ff_$blk.so build/core.o build/gemm.o build/rbf.o build/chol.o build/elbo_ops.o /tmp/t0_$blk.o build/elbo_tn.o
done
cd ../..
echo "== S=3, workgroup 0"; VARGP_HIP_LIB=/tmp/libvargp_ff_0.so python tests/native/bm_stamps.py ff 2>&1 | tail -16 | head -9
echo "== S=8, workgroup 0"; VARGP_BM_S=8 VARGP_HIP_LIB=/tmp/libvargp_ff_0.so python tests/native/bm_stamps.py ff 2>&1 | tail -16 | head -9
echo "== S=8, workgroup 600 (third round on its CU)"; VARGP_BM_S=8 VARGP_HIP_LIB=/tmp/libvargp_ff_600.so python tests/native/bm_stamps.py ff 2>&1 | tail -16 | head -9
