// Multi-tile form of t0_bwd_mid_kernel (t0_bwd_mid.h) for THROUGHPUT-bound shapes -- more (s, c, 64-column tile) units than the
// chip has CUs (BASELINE config 4's per-rank shares: S = 8, 16 hyper-samples x 10 classes x 8 tiles).  A workgroup walks
// ntile / nparts consecutive tiles of ONE (s, c):
//   * G and T are both LDS-resident for the whole workgroup (45 + 45 KB) and staged once, not once per tile (80 of the 190 KB a
//     single-tile workgroup pulls in);
//   * the two M x M products gG += tril(P gW^T), gT += tril(gP K_uf^T) keep their accumulators in registers ACROSS the tiles: one
//     round of float atomics per workgroup (nparts per (s, c)) instead of one per tile (ntile per (s, c)) -- and none of them
//     inside the MFMA loops;
//   * ga, gkd, the row sums r_uf and the gtheta share accumulate in registers likewise;
//   * the next tile's P / W tiles are requested before the current tile's last product and land under it, its K_uf tile (needed
//     in phase 3 only) at its own top; the barriers inside the tile loop wait for LDS operations only (bmm_lds_barrier), so
//     neither these loads nor the last tile's round of atomics are drained at a barrier;
//   * the likelihood is NOT evaluated here (the single-tile kernel can: every (s, c, tile) workgroup then redoes the softmax over
//     all classes of its columns -- C-fold redundant vector work, 11k cycles per tile, which a throughput-bound launch cannot
//     hide): the forward runs its own softmax launch for these shapes and this kernel reads gmu / gvar.
// Same arithmetic per tile as the single-tile kernel (same MFMA block tables: t0_bwd_mid.h); sums over tiles are taken in
// registers instead of by atomics, i.e. in a fixed order.
#pragma once
#include "t0_bwd_mid.h"

namespace vargp {

#ifdef BM_STAMPS      // per-wave stamps (lane 0 of each wave of workgroup BM_STAMP_BLOCK): its second tile, its set-up and its whole life
__device__ unsigned long long g_bmm_stamps[4][24];
extern "C" void vargp_debug_bmm_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bmm_stamps), sizeof(g_bmm_stamps)); }
#define BMM_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == (BM_STAMP_BLOCK) && (tile == tile0 + 1 || (i) >= 20)) g_bmm_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BMM_STAMP(i) do { } while (0)
#endif

constexpr size_t kBwdMidMultiLdsBytes =
    sizeof(float) * (2 * kBmKP * kBmSA + 2 * kBmKP * kBmST + 128 /*a*/ + 64 + 64 /*gmu, gvar*/ + 64 /*column sums*/ + 8);

// acc[u] += X[rows of block rb(u)] Y[rows of block cb(u)]^T over the 64 columns of two M x 64 LDS tiles (no clearing: the blocks
// accumulate over the tiles of the workgroup)
template <int WV>
__device__ __forceinline__ void bmm_tri_mfma_acc(const float* __restrict__ sX, const float* __restrict__ sY, int li, int lh,
                                                 bm_f32x16 (&acc)[3]) {
  using W = BmWave<WV>;
  const float* xr[3];
  const float* yr[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    xr[u] = sX + min(32 * W::rb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
    yr[u] = sY + min(32 * W::cb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
  }
  // fragments one k-group ahead of their MFMAs
  float4 nx[3], ny[3];
#pragma unroll
  for (int u = 0; u < W::NB; ++u) { nx[u] = bm_frag_kc(xr[u], 0); ny[u] = bm_frag_kc(yr[u], 0); }
  bm_for<0, 8>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    float4 cx[3], cy[3];
#pragma unroll
    for (int u = 0; u < W::NB; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; }
    if constexpr (g + 1 < 8) {
#pragma unroll
      for (int u = 0; u < W::NB; ++u) { nx[u] = bm_frag_kc(xr[u], 8 * (g + 1)); ny[u] = bm_frag_kc(yr[u], 8 * (g + 1)); }
    }
#pragma unroll
    for (int u = 0; u < W::NB; ++u) bm_mfma4(acc[u], cx[u], cy[u]);
  });
}

// phase 2 of wave WV:  accG += tril(P gW^T),  accP = a gmu^T - 2 P gvar + G gW
template <int WV>
__device__ __forceinline__ void bmm_phase2(const float* __restrict__ sG, const float* __restrict__ sP, const float* __restrict__ sW,
                                           const float* __restrict__ sa, const float* __restrict__ sgm,
                                           const float* __restrict__ sgv, int li, int lh, bm_f32x16 (&accP)[2],
                                           bm_f32x16 (&accG)[3]) {
  using W = BmWave<WV>;
  bmm_tri_mfma_acc<WV>(sP, sW, li, lh, accG);
  constexpr int R[2] = {W::R0, W::R1};
  const int n = 32 * W::CBH + li;
  {
    const float gmn = sgm[n], gvn = sgv[n];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * R[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int mc = min(m, kBmKP - 1);
        accP[u][r] = m < kBmKP ? fmaf(sa[mc], gmn, -2.f * sP[mc * kBmST + n] * gvn) : 0.f;
      }
  }
  constexpr int G0 = bm_min(kBmKP, 32 * W::R0 + 32) / 8, G1 = bm_min(kBmKP, 32 * W::R1 + 32) / 8;     // G lower: k <= row; G0 < G1
  const float* arow0 = sG + min(32 * W::R0 + li, kBmKP - 1) * kBmSA + 4 * lh;
  const float* arow1 = sG + min(32 * W::R1 + li, kBmKP - 1) * kBmSA + 4 * lh;
  const float* bcol = sW + (4 * lh) * kBmST + n;
  float4 nb = bm_frag_km(bcol, 0, kBmST), na0 = bm_frag_kc(arow0, 0), na1 = bm_frag_kc(arow1, 0);
  bm_for<0, G1>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 bb = nb, a0 = na0, a1 = na1;
    if constexpr (g + 1 < G1) {                        // next group's fragments: in flight under this group's MFMAs
      nb = bm_frag_km(bcol, 8 * (g + 1), kBmST);
      na1 = bm_frag_kc(arow1, 8 * (g + 1));
      if constexpr (g + 1 < G0) na0 = bm_frag_kc(arow0, 8 * (g + 1));
    }
    if constexpr (g < G0) {
      // the two blocks' MFMAs alternate (independent accumulators)
      accP[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bb.x, accP[1], 0, 0, 0);
      accP[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bb.x, accP[0], 0, 0, 0);
      accP[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bb.y, accP[1], 0, 0, 0);
      accP[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bb.y, accP[0], 0, 0, 0);
      accP[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bb.z, accP[1], 0, 0, 0);
      accP[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bb.z, accP[0], 0, 0, 0);
      accP[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bb.w, accP[1], 0, 0, 0);
      accP[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bb.w, accP[0], 0, 0, 0);
    } else {
      bm_mfma4(accP[1], a1, bb);
    }
  });
}

// phase 4 of wave WV:  accT += tril(gP K_uf^T),  accK = T^T gP  ([0]: row block R0, [1]: row block R1)
template <int WV>
__device__ __forceinline__ void bmm_phase4(const float* __restrict__ sT, const float* __restrict__ sP, const float* __restrict__ sW,
                                           int li, int lh, bm_f32x16 (&accK)[2], bm_f32x16 (&accT)[3]) {
  using W = BmWave<WV>;
  bmm_tri_mfma_acc<WV>(sP, sW, li, lh, accT);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) accK[u][r] = 0.f;
  constexpr int GS0 = 4 * W::R0, GS1 = 4 * W::R1, GE = kBmKP / 8;       // T lower: (T^T gP)[m] sums k >= m
  const int n = 32 * W::CBH + li;
  const float* acol0 = sT + (4 * lh) * kBmSA + min(32 * W::R0 + li, kBmKP - 1);
  const float* acol1 = sT + (4 * lh) * kBmSA + min(32 * W::R1 + li, kBmKP - 1);
  const float* bcol = sP + (4 * lh) * kBmST + n;
  float4 nb = bm_frag_km(bcol, 8 * GS0, kBmST), na0 = bm_frag_km(acol0, 8 * GS0, kBmSA), na1 = na0;
  if constexpr (GS1 == GS0) na1 = bm_frag_km(acol1, 8 * GS0, kBmSA);
  bm_for<GS0, GE>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 bb = nb, a0 = na0, a1 = na1;
    if constexpr (g + 1 < GE) {
      nb = bm_frag_km(bcol, 8 * (g + 1), kBmST);
      na0 = bm_frag_km(acol0, 8 * (g + 1), kBmSA);
      if constexpr (g + 1 >= GS1) na1 = bm_frag_km(acol1, 8 * (g + 1), kBmSA);
    }
    if constexpr (g >= GS1) {
      accK[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bb.x, accK[0], 0, 0, 0);
      accK[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bb.x, accK[1], 0, 0, 0);
      accK[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bb.y, accK[0], 0, 0, 0);
      accK[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bb.y, accK[1], 0, 0, 0);
      accK[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bb.z, accK[0], 0, 0, 0);
      accK[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bb.z, accK[1], 0, 0, 0);
      accK[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bb.w, accK[0], 0, 0, 0);
      accK[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bb.w, accK[1], 0, 0, 0);
    } else {
      bm_mfma4(accK[0], a0, bb);
    }
  });
}

// every block register of a wave's M x M accumulators out: dst[row * ldd + col] += acc  (col <= row < M; float atomics:
// nparts workgroups per (s, c) add into the same block, which the forward has cleared)
template <int WV>
__device__ __forceinline__ void bmm_flush_tri(const bm_f32x16 (&acc)[3], float* __restrict__ dst, int ldd, int M, int li, int lh) {
  using W = BmWave<WV>;
  bm_for<0, 16 * W::NB>([&](auto ai) { bm_tri_atomic<WV, decltype(ai)::value>(acc, dst, ldd, M, li, lh); });
}
__device__ __forceinline__ void bmm_flush(const int wave, const bm_f32x16 (&acc)[3], float* __restrict__ dst, int ldd, int M,
                                          int li, int lh) {
  // (M made opaque: as a loop invariant every one of the ~400 uniform tests `32 rb + rl < M` of the four wave variants is
  //  hoisted out of the tile loop and kept in scalar registers across it -- 300+ spilled SGPRs, then spilled vector registers)
  asm volatile("" : "+s"(M));
  if (wave == 0) bmm_flush_tri<0>(acc, dst, ldd, M, li, lh);
  else if (wave == 1) bmm_flush_tri<1>(acc, dst, ldd, M, li, lh);
  else if (wave == 2) bmm_flush_tri<2>(acc, dst, ldd, M, li, lh);
  else bmm_flush_tri<3>(acc, dst, ldd, M, li, lh);
}

__global__ __launch_bounds__(256) void t0_bwd_mid_multi_kernel(const float* __restrict__ TT, const float* __restrict__ QP,
                                                               const float* __restrict__ Wf, const float* __restrict__ RK,
                                                               const float* __restrict__ gmu, const float* __restrict__ gvar,
                                                               const float* __restrict__ gscale, float* __restrict__ gQP,
                                                               float* __restrict__ gTT, float* __restrict__ gRK,
                                                               float* __restrict__ gkd, float* __restrict__ r_uf,
                                                               float* __restrict__ c_uf, float* __restrict__ gtheta, int S, int C,
                                                               int M, int B, int D, int NR, int LD, int ntile, int nparts,
                                                               float* __restrict__ zero_out, int zero_n) {
  extern __shared__ __attribute__((aligned(16))) float bm_lds[];
  STEP_SPAN(t0, 4);
  if (blockIdx.x == 0 && zero_out)
    for (int i = threadIdx.x; i < zero_n; i += 256) zero_out[i] = 0.f;
  float* sG = bm_lds;                               // [KP][SA]   G[m][k] (row-major)
  float* sT = sG + kBmKP * kBmSA;                   // [KP][SA]   T[k][m] (row-major)
  float* sP = sT + kBmKP * kBmSA;                   // [KP][ST]   P tile [m][n], later gP, later W_uf
  float* sW = sP + kBmKP * kBmST;                   // [KP][ST]   W tile -> gW, later the K_uf tile
  float* sa = sW + kBmKP * kBmST;                   // [128]      a = Lz^-1 m
  float* sgm = sa + 128;                            // [64]       seed * gmu of the tile
  float* sgv = sgm + 64;                            // [64]       seed * gvar
  float* scs = sgv + 64;                            // [64]       column sums of W_uf
  const int tid_k = threadIdx.x, tid = tid_k, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware as the single-tile kernel: XCD x works through the matrices b = x, x + 8, ..., all workgroups of one
  // before the next (grid = 8 ceil(SC / 8) nparts; the surplus exits)
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int64_t b = (int64_t)(idx / nparts) * 8 + xcd;
  if (b >= (int64_t)S * C) return;
  const int part = idx % nparts;
  const int tile0 = (part * ntile) / nparts, tile1 = ((part + 1) * ntile) / nparts;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  const float* Qb = QP + b * MLD;
  const float* Tb = TT + b * MM;
  const float* Wb = Wf + b * (int64_t)M * B;
  const float* Kb = RK + b * MLD + NR;
  const float gs = gscale ? gscale[0] : 1.f;
  const int s = (int)(b / C);
  constexpr int NA_ = kBmNA, NT_ = kBmNT;
  { [[maybe_unused]] const int tile = -9; BMM_STAMP(20); }

  // ---- per-tile loads into registers (clamped indices; padding is selected in when the values are stored)
  float4 rp[NT_], rw[NT_], rk[NT_];
  float gmv, gvv;
  // (two groups with different live ranges: the likelihood's inputs are consumed at the top of the tile, the three tiles
  //  during it -- requested together they would hold 160 registers across the last product of the previous tile)
  auto load_tiles = [&](const int n0, const int tid) {
    const int ncl = min(n0 + (tid & 63), B - 1);
    gmv = gmu[b * B + ncl]; gvv = gvar[b * B + ncl];
    bm_load_tile(Qb + NR, LD, M, n0, B, tid, rp);
    bm_load_tile(Wb, B, M, n0, B, tid, rw);
  };

  // ---- once per workgroup: the first tile's inputs (its likelihood loads first: vmcnt retires in order), then G, T, a
  gmv = gvv = 0.f;
  load_tiles(tile0 * 64, tid);
  {
    float4 rg[NA_], rt[NA_];
    bm_load_mat<true>(Qb + 4, LD, M, tid, rg);        // G = T L_S: lower triangular
    bm_load_mat<true>(Tb, M, M, tid, rt);
    const float av = tid < 128 ? Qb[(int64_t)min(tid, M - 1) * LD] : 0.f;
    bm_store_mat<true>(sG, rg, M, tid);
    bm_store_mat<true>(sT, rt, M, tid);
    if (tid < 128) sa[tid] = tid < M ? av : 0.f;
  }
  { [[maybe_unused]] const int tile = -9; BMM_STAMP(21); }

  bm_f32x16 accG[3], accT[3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accG[u][r] = 0.f; accT[u][r] = 0.f; }
  float ga_acc = 0.f, ruf_acc = 0.f, gkd_acc = 0.f, gth_acc = 0.f;      // per thread (m, h) / wave 0: sums over the tiles

  for (int tile = tile0; tile < tile1; ++tile) {
    const int n0 = tile * 64;
    // (the thread's indices are made opaque per iteration: as loop invariants the compiler hoists the hundreds of LDS / global
    //  addresses derived from them out of the loop and spills them -- 1.7 KB of scratch per lane)
    int tid = tid_k;
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(M), "+s"(B), "+s"(LD), "+s"(NR), "+s"(C));      // (likewise the uniform shape parameters)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int cbh = wave & 1;
    const int rbs[2] = {(wave >> 1) ? 1 : 0, (wave >> 1) ? 2 : 3};
    BMM_STAMP(0);
    // ---- phase 0: P, W, gmu, gvar into LDS -------------------------------------------------------------------------------------
    BMM_STAMP(1);
    bm_store_tile(sP, rp, M, n0, B, tid);
    bm_store_tile(sW, rw, M, n0, B, tid);
    BMM_STAMP(15);
    bm_load_tile(Kb, LD, M, n0, B, tid, rk);          // this tile's K_uf: needed in phase 3, lands under phases 1 and 2
    if (tid < 64) {
      const bool ok = n0 + tid < B;
      sgm[tid] = ok ? gs * gmv : 0.f;
      sgv[tid] = ok ? gs * gvv : 0.f;
      scs[tid] = 0.f;
    }
    bmm_lds_barrier();
    BMM_STAMP(2);
    // ---- phase 1: gW = 2 W gvar in place; ga += P gmu; gkd += sum gvar ---------------------------------------------------------
#pragma unroll
    for (int u = 0; u < NT_; ++u) {
      const int e = tid + 256 * u;
      if (e < kBmKP * 16) {
        const int m = e >> 4, n = (e & 15) * 4;
        float4 w = *reinterpret_cast<const float4*>(&sW[m * kBmST + n]);
        const float4 g = *reinterpret_cast<const float4*>(&sgv[n]);
        w.x *= 2.f * g.x; w.y *= 2.f * g.y; w.z *= 2.f * g.z; w.w *= 2.f * g.w;
        *reinterpret_cast<float4*>(&sW[m * kBmST + n]) = w;
      }
    }
    {
      const int m = tid >> 1, h = tid & 1;               // thread (m, h): row m, columns [32 h, 32 h + 32)
      const float* pr = sP + min(m, kBmKP - 1) * kBmST + 32 * h;
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 p = *reinterpret_cast<const float4*>(pr + 4 * i);
        const float4 g = *reinterpret_cast<const float4*>(&sgm[32 * h + 4 * i]);
        acc0 = fmaf(p.x, g.x, acc0); acc1 = fmaf(p.y, g.y, acc1); acc0 = fmaf(p.z, g.z, acc0); acc1 = fmaf(p.w, g.w, acc1);
      }
      ga_acc += acc0 + acc1;
      if (wave == 0) gkd_acc += sgv[lane];
    }
    bmm_lds_barrier();
    BMM_STAMP(3);
    // ---- phase 2: accG += tril(P gW^T),  gP = a gmu^T - 2 P gvar + G gW -----------------------------------------------------------
    bm_f32x16 accP[2];
    if (wave == 0) bmm_phase2<0>(sG, sP, sW, sa, sgm, sgv, li, lh, accP, accG);
    else if (wave == 1) bmm_phase2<1>(sG, sP, sW, sa, sgm, sgv, li, lh, accP, accG);
    else if (wave == 2) bmm_phase2<2>(sG, sP, sW, sa, sgm, sgv, li, lh, accP, accG);
    else bmm_phase2<3>(sG, sP, sW, sa, sgm, sgv, li, lh, accP, accG);
    BMM_STAMP(4);
    // (last tile: the G block is complete -- its round of atomics / stores drains under phases 3 and 4)
    if (tile + 1 == tile1) bmm_flush(wave, accG, gQP + b * MLD + 4, LD, M, li, lh);
    bmm_lds_barrier();                                  // everybody is done with P and gW
    BMM_STAMP(5);
    // ---- phase 3: gP into P's place, the K_uf tile into gW's place; then the NEXT tile's loads (they land under phase 4) -----------
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < kBmKP) sP[m * kBmST + 32 * cbh + li] = m < M ? accP[u][r] : 0.f;
      }
    bm_store_tile(sW, rk, M, n0, B, tid);
    if (tile + 1 < tile1) load_tiles(n0 + 64, tid);
    bmm_lds_barrier();
    BMM_STAMP(6);
    // ---- phase 4: accT += tril(gP K_uf^T),  gK_uf = T^T gP,  W_uf = gK_uf o K_uf ---------------------------------------------------
    {
      bm_f32x16 acc[2];      // [0]: row block rbs[0], [1]: row block rbs[1]
      if (wave == 0) bmm_phase4<0>(sT, sP, sW, li, lh, acc, accT);
      else if (wave == 1) bmm_phase4<1>(sT, sP, sW, li, lh, acc, accT);
      else if (wave == 2) bmm_phase4<2>(sT, sP, sW, li, lh, acc, accT);
      else bmm_phase4<3>(sT, sP, sW, li, lh, acc, accT);
      BMM_STAMP(7);
      if (tile + 1 == tile1) bmm_flush(wave, accT, gTT + b * MM, M, M, li, lh);      // drains under the epilogue
      const int n = 32 * cbh + li;
      float wv[2][16];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
          wv[u][r] = m < M ? acc[u][r] * sW[min(m, kBmKP - 1) * kBmST + n] : 0.f;      // K_uf is zero in the columns past B
        }
      BMM_STAMP(16);
      bmm_lds_barrier();                                // everybody is done with gP as an operand
      BMM_STAMP(17);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < kBmKP) sP[m * kBmST + n] = wv[u][r];
        }
    }
    bmm_lds_barrier();
    BMM_STAMP(8);
    {
      // W_uf rows out (coalesced float4), column sums, row sums
      float* Gout = gRK + b * MLD + NR + n0;
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
      const int n4 = (tid & 15) * 4;                    // the same four columns in every round (256 % 16 == 0)
      float4 wv4[kBmNT];
#pragma unroll
      for (int u = 0; u < kBmNT; ++u) {
        const int e = min(tid + 256 * u, kBmKP * 16 - 1);
        wv4[u] = *reinterpret_cast<const float4*>(&sP[(e >> 4) * kBmST + n4]);
      }
#pragma unroll
      for (int u = 0; u < kBmNT; ++u) {
        const int e = tid + 256 * u, m = e >> 4;
        if (e < kBmKP * 16) { cs.x += wv4[u].x; cs.y += wv4[u].y; cs.z += wv4[u].z; cs.w += wv4[u].w; }
        if (m < M && n0 + n4 < B) *reinterpret_cast<float4*>(&Gout[(int64_t)m * LD + n4]) = wv4[u];
      }
      atomicAdd(&scs[n4], cs.x); atomicAdd(&scs[n4 + 1], cs.y); atomicAdd(&scs[n4 + 2], cs.z); atomicAdd(&scs[n4 + 3], cs.w);
      const int m = tid >> 1, h = tid & 1;
      const float* pr = sP + min(m, kBmKP - 1) * kBmST + 32 * h;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(pr + 4 * i);
        a0 += v.x + v.z; a1 += v.y + v.w;
      }
      ruf_acc += a0 + a1;
    }
    bmm_lds_barrier();
    BMM_STAMP(9);
    if (tid < 64) {
      const float cv = scs[tid];
      if (n0 + tid < B) atomicAdd(&c_uf[(int64_t)s * B + n0 + tid], cv);
      gth_acc += cv;                                    // (wave 0: summed over its lanes at the end)
    }
    BMM_STAMP(10);
    // (no barrier here: the next iteration's first LDS writes -- the likelihood's scratch in P's place, then P / W -- touch what
    //  everybody stopped reading before the barrier above, and scs is cleared by the threads that have just read it)
  }

  // ---- the vector sums over this workgroup's tiles (the M x M blocks went out inside the last tile) ----------------------------
  {
    const int m = tid >> 1, h = tid & 1;
    ga_acc += __shfl_xor(ga_acc, 1, 64);
    ruf_acc += __shfl_xor(ruf_acc, 1, 64);
    if (h == 0 && m < M) {
      atomicAdd(&gQP[b * MLD + (int64_t)m * LD], ga_acc);
      atomicAdd(&r_uf[b * M + m], ruf_acc);            // b * M + m == s * C * M + c * M + m
    }
    if (wave == 0) {
      const float tv = wave_sum(gkd_acc);
      const float tot = wave_sum(gth_acc);
      if (lane == 0) {
        atomicAdd(&gkd[b], tv);
        atomicAdd(&gtheta[(int64_t)s * (D + 1) + D], 2.f * tot);
      }
    }
  }
  { [[maybe_unused]] const int tile = -9; BMM_STAMP(22); }
}

}  // namespace vargp
