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

// ---- f32 MFMA 16x16x4 blocking: M <= 104 is SEVEN 16-row blocks (112 rows: 1.25x padding work for M = 100) instead of four 32-row
// blocks (128 rows: 1.64x).  Lane (l16, q) = (lane & 15, lane >> 4); the four MFMAs of one k-group of 16 take k = 16 g + 4 q + j,
// j < 4, from lane group q (one b128 read per lane for a K-contiguous operand, four b32 reads for a k-major one); accumulator
// register r of a block holds row 16 rb + 4 q + r, column 16 cb + l16.  Rows / inner indices 104 .. 111 do not exist in LDS:
// row indices are clamped (the results of those rows are never stored), inner indices are masked to zero.
typedef float bm_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kB16NB = 7;       // 16-row blocks of an M-long dimension
__device__ __forceinline__ void b16_mfma4(bm_f32x4& acc, const float4 a, const float4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}
__device__ __forceinline__ float4 b16_mask(const float4 v, const bool ok) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
// The 28 blocks of the lower triangle of an M x M result, seven per wave: wave WV holds block row 6 - WV (7 - WV blocks) and, for
// WV > 0, block row WV - 1 (WV blocks) -- the column blocks of the second row are a subset of the first's, so one k-group needs at
// most eight fragment reads for its 28 MFMAs.
template <int WV> struct B16Tri {
  static constexpr int RA = 6 - WV, NA = 7 - WV, RB = WV > 0 ? WV - 1 : 0, NBk = WV;
  __host__ __device__ static constexpr int rb(int u) { return u < NA ? RA : RB; }
  __host__ __device__ static constexpr int cb(int u) { return u < NA ? u : u - NA; }
};
// acc[u] += X[rows of block rb(u)] Y[rows of block cb(u)]^T over the 64 columns of two M x 64 LDS tiles ([row][col], stride kBmST)
template <int WV>
__device__ __forceinline__ void b16_tri_acc(const float* __restrict__ sX, const float* __restrict__ sY, int l16, int q,
                                            bm_f32x4 (&acc)[kB16NB]) {
  using W = B16Tri<WV>;
  const float* xa = sX + min(16 * W::RA + l16, kBmKP - 1) * kBmST + 4 * q;
  const float* xb = sX + min(16 * W::RB + l16, kBmKP - 1) * kBmST + 4 * q;
  const float* yr[W::NA];
#pragma unroll
  for (int c = 0; c < W::NA; ++c) yr[c] = sY + min(16 * c + l16, kBmKP - 1) * kBmST + 4 * q;
  float4 nxa = bm_frag_kc(xa, 0), nxb = nxa, ny[W::NA];
  if constexpr (W::NBk > 0) nxb = bm_frag_kc(xb, 0);
#pragma unroll
  for (int c = 0; c < W::NA; ++c) ny[c] = bm_frag_kc(yr[c], 0);
  bm_for<0, 4>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 cxa = nxa, cxb = nxb;
    float4 cy[W::NA];
#pragma unroll
    for (int c = 0; c < W::NA; ++c) cy[c] = ny[c];
    if constexpr (g + 1 < 4) {                         // next group's fragments: in flight under this group's MFMAs
      nxa = bm_frag_kc(xa, 16 * (g + 1));
      if constexpr (W::NBk > 0) nxb = bm_frag_kc(xb, 16 * (g + 1));
#pragma unroll
      for (int c = 0; c < W::NA; ++c) ny[c] = bm_frag_kc(yr[c], 16 * (g + 1));
    }
#pragma unroll
    for (int c = 0; c < W::NA; ++c) b16_mfma4(acc[c], cxa, cy[c]);
#pragma unroll
    for (int c = 0; c < W::NBk; ++c) b16_mfma4(acc[W::NA + c], cxb, cy[c]);
  });
}
// An M x 64 result, 16 columns per wave (column block CW = wave), all seven row blocks:
//   LOWER:  acc[i] += sum_{k <= row} A[row][k] B[k][n]      A = sA[row][k] (lower triangular, K-contiguous rows, stride SA)
//   !LOWER: acc[i] += sum_{k >= row} A[k][row] B[k][n]      A = sA[k][row] (the transpose of a lower triangular matrix, k-major)
// B = sB[k][n] (k-major, stride kBmST).  Row block i takes part in the k-groups g <= i (LOWER) / g >= i.
template <bool LOWER, int SA = kBmSA>
__device__ __forceinline__ void b16_prod(const float* __restrict__ sA, const float* __restrict__ sB, int cw, int l16, int q,
                                         bm_f32x4 (&acc)[kB16NB]) {
  const float* ap[kB16NB];
#pragma unroll
  for (int i = 0; i < kB16NB; ++i)
    ap[i] = LOWER ? sA + min(16 * i + l16, kBmKP - 1) * SA + 4 * q : sA + (4 * q) * SA + min(16 * i + l16, kBmKP - 1);
  const float* bcol = sB + (4 * q) * kBmST + 16 * cw + l16;
  auto afrag = [&](const int i, const int g) -> float4 {
    // (the last k-group holds k = 96 .. 111: lane groups 2, 3 are past the 104 rows / columns that exist)
    if (LOWER) return g == 6 ? b16_mask(bm_frag_kc(ap[i], 16 * g - (q >= 2 ? 8 : 0)), q < 2) : bm_frag_kc(ap[i], 16 * g);
    return g == 6 ? b16_mask(bm_frag_km(ap[i], 16 * g - (q >= 2 ? 8 : 0), SA), q < 2) : bm_frag_km(ap[i], 16 * g, SA);
  };
  auto bfrag = [&](const int g) -> float4 {
    return g == 6 ? b16_mask(bm_frag_km(bcol, 16 * g - (q >= 2 ? 8 : 0), kBmST), q < 2) : bm_frag_km(bcol, 16 * g, kBmST);
  };
  constexpr int G0 = 0, G1 = kB16NB;
  float4 nb = bfrag(G0), na[kB16NB];
#pragma unroll
  for (int i = 0; i < kB16NB; ++i) na[i] = (LOWER ? i >= G0 : i <= G0) ? afrag(i, G0) : make_float4(0.f, 0.f, 0.f, 0.f);
  bm_for<G0, G1>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 bb = nb;
    float4 ca[kB16NB];
#pragma unroll
    for (int i = 0; i < kB16NB; ++i) ca[i] = na[i];
    if constexpr (g + 1 < G1) {
      nb = bfrag(g + 1);
#pragma unroll
      for (int i = 0; i < kB16NB; ++i)
        if (LOWER ? i >= g + 1 : i <= g + 1) na[i] = afrag(i, g + 1);
    }
#pragma unroll
    for (int i = 0; i < kB16NB; ++i)
      if (LOWER ? i >= g : i <= g) b16_mfma4(acc[i], ca[i], bb);
  });
}
// register A (flat index u * 4 + r) of a wave's lower-triangle blocks, added to dst[row * ldd + col] (col <= row < M)
template <int WV, int A>
__device__ __forceinline__ void b16_tri_atomic(const bm_f32x4 (&acc)[kB16NB], float* __restrict__ dst, int ldd, int M, int l16, int q) {
  using W = B16Tri<WV>;
  constexpr int u = A / 4, r = A % 4, rb = W::rb(u), cb = W::cb(u);
  if (16 * rb + r < M) {                                // (uniform) the register holds at least one row < M
    const int row = 16 * rb + 4 * q + r, col = 16 * cb + l16;
    if (row < M && (rb != cb || col <= row)) atomicAdd(&dst[(int64_t)row * ldd + col], acc[u][r]);
  }
}
template <int WV>
__device__ __forceinline__ void b16_flush_tri(const bm_f32x4 (&acc)[kB16NB], float* __restrict__ dst, int ldd, int M, int l16, int q) {
  bm_for<0, 4 * kB16NB>([&](auto ai) { b16_tri_atomic<WV, decltype(ai)::value>(acc, dst, ldd, M, l16, q); });
}
__device__ __forceinline__ void b16_flush(const int wave, const bm_f32x4 (&acc)[kB16NB], float* __restrict__ dst, int ldd, int M,
                                          int l16, int q) {
  // (M made opaque: as a loop invariant every uniform test `16 rb + r < M` of the four wave variants is hoisted out of the tile
  //  loop and kept in scalar registers across it)
  asm volatile("" : "+s"(M));
  if (wave == 0) b16_flush_tri<0>(acc, dst, ldd, M, l16, q);
  else if (wave == 1) b16_flush_tri<1>(acc, dst, ldd, M, l16, q);
  else if (wave == 2) b16_flush_tri<2>(acc, dst, ldd, M, l16, q);
  else b16_flush_tri<3>(acc, dst, ldd, M, l16, q);
}
__device__ __forceinline__ void b16_tri(const int wave, const float* __restrict__ sX, const float* __restrict__ sY, int l16, int q,
                                        bm_f32x4 (&acc)[kB16NB]) {
  if (wave == 0) b16_tri_acc<0>(sX, sY, l16, q, acc);
  else if (wave == 1) b16_tri_acc<1>(sX, sY, l16, q, acc);
  else if (wave == 2) b16_tri_acc<2>(sX, sY, l16, q, acc);
  else b16_tri_acc<3>(sX, sY, l16, q, acc);
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

  bm_f32x4 accG[kB16NB], accT[kB16NB];
#pragma unroll
  for (int u = 0; u < kB16NB; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) { accG[u][r] = 0.f; accT[u][r] = 0.f; }
  float ga_acc = 0.f, ruf_acc = 0.f, gkd_acc = 0.f, gth_acc = 0.f;      // per thread (m, h) / wave 0: sums over the tiles

  for (int tile = tile0; tile < tile1; ++tile) {
    const int n0 = tile * 64;
    // (the thread's indices are made opaque per iteration: as loop invariants the compiler hoists the hundreds of LDS / global
    //  addresses derived from them out of the loop and spills them -- 1.7 KB of scratch per lane)
    int tid = tid_k;
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(M), "+s"(B), "+s"(LD), "+s"(NR), "+s"(C));      // (likewise the uniform shape parameters)
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, q = lane >> 4;
    const int n = 16 * wave + l16;                     // this lane's column of the M x 64 results
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
    b16_tri(wave, sP, sW, l16, q, accG);
    bm_f32x4 accP[kB16NB];
    {
      const float gmn = sgm[n], gvn = sgv[n];
#pragma unroll
      for (int i = 0; i < kB16NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * q + r, mc = min(m, kBmKP - 1);
          accP[i][r] = m < kBmKP ? fmaf(sa[mc], gmn, -2.f * sP[mc * kBmST + n] * gvn) : 0.f;
        }
    }
    b16_prod<true>(sG, sW, wave, l16, q, accP);
    BMM_STAMP(4);
    // (last tile: the G block is complete -- its round of atomics drains under phases 3 and 4)
    if (tile + 1 == tile1) b16_flush(wave, accG, gQP + b * MLD + 4, LD, M, l16, q);
    bmm_lds_barrier();                                  // everybody is done with P and gW
    BMM_STAMP(5);
    // ---- phase 3: gP into P's place, the K_uf tile into gW's place; then the NEXT tile's loads (they land under phase 4) -----------
#pragma unroll
    for (int i = 0; i < kB16NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * q + r;
        if (m < kBmKP) sP[m * kBmST + n] = m < M ? accP[i][r] : 0.f;
      }
    bm_store_tile(sW, rk, M, n0, B, tid);
    if (tile + 1 < tile1) load_tiles(n0 + 64, tid);
    bmm_lds_barrier();
    BMM_STAMP(6);
    // ---- phase 4: accT += tril(gP K_uf^T),  gK_uf = T^T gP,  W_uf = gK_uf o K_uf ---------------------------------------------------
    {
      b16_tri(wave, sP, sW, l16, q, accT);
      bm_f32x4 acc[kB16NB];
#pragma unroll
      for (int i = 0; i < kB16NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
      b16_prod<false>(sT, sP, wave, l16, q, acc);
      BMM_STAMP(7);
      if (tile + 1 == tile1) b16_flush(wave, accT, gTT + b * MM, M, M, l16, q);      // drains under the epilogue
      float wv[kB16NB][4];
#pragma unroll
      for (int i = 0; i < kB16NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * q + r;
          wv[i][r] = m < M ? acc[i][r] * sW[min(m, kBmKP - 1) * kBmST + n] : 0.f;      // K_uf is zero in the columns past B
        }
      BMM_STAMP(16);
      bmm_lds_barrier();                                // everybody is done with gP as an operand
      BMM_STAMP(17);
#pragma unroll
      for (int i = 0; i < kB16NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * i + 4 * q + r;
          if (m < kBmKP) sP[m * kBmST + n] = wv[i][r];
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
