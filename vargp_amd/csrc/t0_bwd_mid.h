// The middle of the first-task backward as ONE LDS-resident MFMA kernel (M <= 104, M % 4 == 0, B % 4 == 0): for every
// (s, c) and 64-column tile of the minibatch, with P = T K_uf, W = G^T P of the forward (gp_utils.py:178-186) and the softmax
// gradients gmu, gvar of this tile:
//     gW  = 2 W gvar                                   gP = a gmu^T - 2 P gvar + G gW
//     ga += P gmu            gkd += sum gvar           gG += tril(P gW^T)                (float atomics: 8 tiles per (s, c))
//     gT += tril(gP K_uf^T)                            gK_uf = T^T gP
//     W_uf = gK_uf o K_uf  -> the K_uf block of gRK,   r_uf += row sums,  c_uf += column sums,  gtheta[s, D] += 2 sum W_uf
// i.e. the predictive-moment backward, the products gG || gP, the K_uf share of gT = tril(gQP RK^T) and of gRK = T^T gQP,
// and the K_uf role of the W = gK o K pass (rbf.hip) -- four launches and the HBM round trips of gP, gW, gK_uf -- with
// G, T, and the P / W / K_uf tiles resident in LDS.  What is left for GEMM launches afterwards is the small-column part
// (K = NR): gT += tril(gQP[:, :NR] RK[:, :NR]^T) and gRK[:, :NR] = T^T gQP[:, :NR].
// 256 threads = 4 waves.  f32 MFMA 32x32x2 with the k-pairing of gemm.hip (half-wave h supplies k = 8 g + 4 h + j, j < 4).
//   products with a full M x M result (gG, gT: K = 64 columns): the 10 blocks of the lower triangle, 3 / 3 / 2 / 2 per wave;
//   products with an M x 64 result (gP, gK_uf): wave w owns the 32-column half (w & 1) and the row blocks {0, 3} or {1, 2}
//   (triangular K ranges: both pairs carry the same work), exactly as the forward kernel (t0_fwd_fused_kernel).
#pragma once
#include "t0_bwd_common.h"

namespace vargp {

// Work of wave WV (compile-time: every loop below is fully unrolled and register indices are static).
//   lower-triangle blocks (row block, column block) of the M x M results: 3 / 3 / 2 / 2 per wave;
//   M x 64 results: column half WV & 1, row blocks {0, 3} (waves 0, 1) or {1, 2} (waves 2, 3).
template <int WV> struct BmWave {
  static constexpr int NB = WV < 2 ? 3 : 2;
  static constexpr int R0 = WV < 2 ? 0 : 1, R1 = WV < 2 ? 3 : 2, CBH = WV & 1;
  __host__ __device__ static constexpr int rb(int u) { return WV == 0 ? (u == 0 ? 0 : 1) : (WV == 1 ? 2 : 3); }
  __host__ __device__ static constexpr int cb(int u) { return WV == 0 ? (u == 2 ? 1 : 0) : (WV == 1 ? u : (WV == 2 ? u : 2 + u)); }
};

// how many of a wave's 16 NB block registers go out under the NEXT product's k-groups (the rest under the one after)
__host__ __device__ constexpr int kBmAtomSplit(int nb) { return 8 * nb; }

// acc[u] = X[rows of block rb(u)] Y[rows of block cb(u)]^T over the 64 columns of two M x 64 LDS tiles ([row][col], stride kBmST)
template <int WV>
__device__ __forceinline__ void bm_tri_mfma(const float* __restrict__ sX, const float* __restrict__ sY, int li, int lh,
                                            bm_f32x16 (&acc)[3]) {
  using W = BmWave<WV>;
  const float* xr[3];
  const float* yr[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    xr[u] = sX + min(32 * W::rb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
    yr[u] = sY + min(32 * W::cb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
  }
#pragma unroll
  for (int k = 0; k < 64; k += 8) {
#pragma unroll
    for (int u = 0; u < W::NB; ++u) bm_mfma4(acc[u], bm_frag_kc(xr[u], k), bm_frag_kc(yr[u], k));
  }
}
// register A (flat index u * 16 + r) of the blocks above, added to dst[row * ldd + col] (col <= row < M)
template <int WV, int A>
__device__ __forceinline__ void bm_tri_atomic(const bm_f32x16 (&acc)[3], float* __restrict__ dst, int ldd, int M, int li, int lh) {
  using W = BmWave<WV>;
  constexpr int u = A / 16, r = A % 16, rb = W::rb(u), cb = W::cb(u);
  constexpr int rl = (r & 3) + 8 * (r >> 2);
  if (32 * rb + rl < M) {                              // (uniform) the register holds at least one row < M
    const int row = 32 * rb + rl + 4 * lh, col = 32 * cb + li;
    const bool ok = row < M && (rb != cb || col <= row);
#ifndef BM_EXP_NOATOM
#ifdef BM_EXP_PLAIN_STORES      // tuning builds (wrong results): what the float atomics of the two triangular products cost
    if (ok) dst[(int64_t)row * ldd + col] = acc[u][r];
#else
    if (ok) atomicAdd(&dst[(int64_t)row * ldd + col], acc[u][r]);
#endif
#else
    if (ok && acc[u][r] == 12345.f) dst[(int64_t)row * ldd + col] = 0.f;
#endif
  }
}

// phase 2 of wave WV:  gG += tril(P gW^T)  (atomics into dst),  accP = a gmu^T - 2 P gvar + G gW.  The atomics of the first
// product are issued a few per k-group of the second: a wave can keep ~16 atomics in flight, so 48 of them in a row stall for
// several memory round trips (measured: 5 us per product); spread out they drain under the MFMAs.
template <int WV>
__device__ __forceinline__ void bm_phase2(const float* __restrict__ sA, const float* __restrict__ sP, const float* __restrict__ sW,
                                          const float* __restrict__ sa, const float* __restrict__ sgm,
                                          const float* __restrict__ sgv, float* __restrict__ dst, int ldd, int M, int li, int lh,
                                          bm_f32x16 (&accP)[2], bm_f32x16 (&acc)[3]) {
  using W = BmWave<WV>;
  bm_tri_mfma<WV>(sP, sW, li, lh, acc);
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
  // the chip adds ~1.3 TB/s of atomic bytes: all workgroups issuing a product's 23 KB at once is a 5 us burst.  Half of them
  // go out under this product's MFMAs, the other half under the next M x M product's (bm_phase4).
  constexpr int NAT = kBmAtomSplit(W::NB), PER = (NAT + G1 - 1) / G1;
  const float* arow0 = sA + min(32 * W::R0 + li, kBmKP - 1) * kBmSA + 4 * lh;
  const float* arow1 = sA + min(32 * W::R1 + li, kBmKP - 1) * kBmSA + 4 * lh;
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
    bm_mfma4(accP[1], a1, bb);
    if constexpr (g < G0) bm_mfma4(accP[0], a0, bb);
    bm_for<g * PER, bm_min((g + 1) * PER, NAT)>([&](auto ai) { bm_tri_atomic<WV, decltype(ai)::value>(acc, dst, ldd, M, li, lh); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// phase 4 of wave WV:  gT += tril(gP K_uf^T)  (atomics into dst),  accK = T^T gP  ([0]: row block R0, [1]: row block R1)
template <int WV>
__device__ __forceinline__ void bm_phase4(const float* __restrict__ sA, const float* __restrict__ sP, const float* __restrict__ sW,
                                          float* __restrict__ dst, int ldd, int M, int li, int lh, bm_f32x16 (&accK)[2],
                                          const bm_f32x16 (&prev)[3], float* __restrict__ dprev, int ldprev, bm_f32x16 (&acc)[3]) {
  using W = BmWave<WV>;
  {   // M x M product with the second half of the previous one's atomics between its k-groups
    const float* xr[3];
    const float* yr[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
      xr[u] = sP + min(32 * W::rb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
      yr[u] = sW + min(32 * W::cb(u < W::NB ? u : 0) + li, kBmKP - 1) * kBmST + 4 * lh;
    }
    constexpr int A0 = kBmAtomSplit(W::NB), A1 = 16 * W::NB, PERP = (A1 - A0 + 7) / 8;
    bm_for<0, 8>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
#pragma unroll
      for (int u = 0; u < W::NB; ++u) bm_mfma4(acc[u], bm_frag_kc(xr[u], 8 * g), bm_frag_kc(yr[u], 8 * g));
      bm_for<A0 + g * PERP, bm_min(A0 + (g + 1) * PERP, A1)>(
          [&](auto ai) { bm_tri_atomic<WV, decltype(ai)::value>(prev, dprev, ldprev, M, li, lh); });
    });
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) accK[u][r] = 0.f;
  constexpr int GS0 = 4 * W::R0, GS1 = 4 * W::R1, GE = kBmKP / 8;       // T lower: (T^T gP)[m] sums k >= m
  constexpr int NIT = GE - GS0, NAT = kBmAtomSplit(W::NB), PER = (NAT + NIT - 1) / NIT;
  const int n = 32 * W::CBH + li;
  const float* acol0 = sA + (4 * lh) * kBmSA + min(32 * W::R0 + li, kBmKP - 1);
  const float* acol1 = sA + (4 * lh) * kBmSA + min(32 * W::R1 + li, kBmKP - 1);
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
    bm_mfma4(accK[0], a0, bb);
    if constexpr (g >= GS1) bm_mfma4(accK[1], a1, bb);
    constexpr int it = g - GS0;
    bm_for<it * PER, bm_min((it + 1) * PER, NAT)>([&](auto ai) { bm_tri_atomic<WV, decltype(ai)::value>(acc, dst, ldd, M, li, lh); });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// the second half of a wave's M x M block registers (the first half went out between the k-groups of the product after it)
template <int WV>
__device__ __forceinline__ void bm_tail_atomics(const bm_f32x16 (&acc)[3], float* __restrict__ dst, int ldd, int M, int li, int lh) {
  using W = BmWave<WV>;
  bm_for<kBmAtomSplit(W::NB), 16 * W::NB>([&](auto ai) { bm_tri_atomic<WV, decltype(ai)::value>(acc, dst, ldd, M, li, lh); });
}

#ifdef BM_STAMPS   // per-phase cycle accounting (workgroup (0, 0), thread 0), tuning builds only: tests/native/bm_stamps.py
__device__ unsigned long long g_bm_stamps[16];
extern "C" void vargp_debug_bm_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bm_stamps), 128); }
#ifndef BM_STAMP_BLOCK
#define BM_STAMP_BLOCK 0u      // which workgroup stamps (-DBM_STAMP_BLOCK=...: a late one, in its CU's second or third round)
#endif
#define BM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == (BM_STAMP_BLOCK)) g_bm_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BM_STAMP(i) do { } while (0)
#endif
#ifdef BM_CUT      // tuning builds only: leave the kernel after phase BM_CUT (results are wrong, durations tell the phase costs)
#define BM_CUT_AT(i) do { if (BM_CUT == (i)) return; } while (0)
#else
#define BM_CUT_AT(i) do { } while (0)
#endif

constexpr int kBmNT = (kBmKP * 16 + 255) / 256;     // float4 per thread of one M x 64 tile (7)

// one M x 64 tile ([m][n0 + n], row stride ld in global memory, B % 4 == 0 columns) -> registers.  Branch-free: a float4 lies
// wholly inside or wholly outside the matrix, so the column is clamped and the padding selected in when it is stored (a
// branch around a load makes the compiler drain vmcnt before the next one: one memory round trip per load)
__device__ __forceinline__ void bm_load_tile(const float* __restrict__ base, int ld, int M, int n0, int B, int tid,
                                             float4 (&dst)[kBmNT]) {
  const char* bp = reinterpret_cast<const char*>(base);      // (32-bit byte offsets: see bm_load_mat)
#pragma unroll
  for (int u = 0; u < kBmNT; ++u) {
    const int e = min(tid + 256 * u, kBmKP * 16 - 1);
    const int m = e >> 4, n = (e & 15) * 4;
    const unsigned off = 4u * (__umul24((unsigned)min(m, M - 1), (unsigned)ld) + (unsigned)min(n0 + n, B - 4));
    dst[u] = *reinterpret_cast<const float4*>(bp + off);
  }
}
// registers -> LDS tile [m][n] (stride kBmST), rows m >= M and columns n0 + n >= B zero
__device__ __forceinline__ void bm_store_tile(float* __restrict__ dst, const float4 (&src)[kBmNT], int M, int n0, int B, int tid) {
#pragma unroll
  for (int u = 0; u < kBmNT; ++u) {
    const int e = tid + 256 * u;
    const int m = e >> 4, n = (e & 15) * 4;
    const float4 v = (m < M && n0 + n < B) ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < kBmKP * 16) *reinterpret_cast<float4*>(&dst[m * kBmST + n]) = v;
  }
}

// registers -> LDS [i][j] (stride kBmSA), rows / columns >= M zero
template <bool TRI = false>
__device__ __forceinline__ void bm_store_mat(float* __restrict__ dst, const float4 (&src)[kBmNA], int M, int tid) {
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) {
    const int e = tid + 256 * u;
    const int i = e / kBmNQ, j = (e - i * kBmNQ) * 4;
    const float4 v = (i < M && j < M && (!TRI || j <= i)) ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < kBmKP * kBmNQ) *reinterpret_cast<float4*>(&dst[i * kBmSA + j]) = v;
  }
}

__global__ __launch_bounds__(256) void t0_bwd_mid_kernel(const float* __restrict__ TT, const float* __restrict__ QP,
                                                         const float* __restrict__ Wf, const float* __restrict__ RK,
                                                         const float* __restrict__ gmu, const float* __restrict__ gvar,
                                                         const float* __restrict__ gscale, float* __restrict__ gQP,
                                                         float* __restrict__ gTT, float* __restrict__ gRK,
                                                         float* __restrict__ gkd, float* __restrict__ r_uf,
                                                         float* __restrict__ c_uf, float* __restrict__ gtheta, int S, int C,
                                                         int M, int B, int D, int NR, int LD, int ntile,
                                                         float* __restrict__ zero_out, int zero_n, const BmSoftmax sm) {
  extern __shared__ __attribute__((aligned(16))) float bm_lds[];
  STEP_SPAN(t0, 4);
  // an output of the backward that the NEXT launch accumulates into (g_u_mean: sums over s by atomics): cleared here, by the
  // first workgroup, because the forward -- which clears the workspace's accumulators -- does not know the caller's buffer
  if (blockIdx.x == 0 && zero_out)
    for (int i = threadIdx.x; i < zero_n; i += 256) zero_out[i] = 0.f;
  float* sA = bm_lds;                               // [KP][SA]   G[m][k] (row-major), later T[k][m] (row-major)
  float* sP = sA + kBmKP * kBmSA;                   // [KP][ST]   P tile [m][n], later gP
  float* sW = sP + kBmKP * kBmST;                   // [KP][ST]   W tile -> gW, later the K_uf tile
  float* sa = sW + kBmKP * kBmST;                   // [128]      a = Lz^-1 m
  float* sgm = sa + 128;                            // [64]       seed * gmu of the tile
  float* sgv = sgm + 64;                            // [64]       seed * gvar
  float* scs = sgv + 64;                            // [64]       column sums of W_uf
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  // 1-D grid, XCD-aware: workgroups go to the 8 XCDs round-robin by linear id, and the tiles of one (s, c) share G and T -- with
  // (tile, b) = (id % ntile, id / ntile) the 8 tiles of a matrix sat on 8 different L2s.  Here XCD x works through the
  // matrices b = x, x + 8, ..., all tiles of one before the next (grid = 8 ceil(SC / 8) ntile; the surplus exits).
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int64_t b = (int64_t)(idx / ntile) * 8 + xcd;
  if (b >= (int64_t)S * C) return;
  const int n0 = (idx % ntile) * 64;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  const float* Qb = QP + b * MLD;
  const float* Tb = TT + b * MM;
  const float* Wb = Wf + b * (int64_t)M * B;
  const float* Kb = RK + b * MLD + NR;
  const float gs = gscale ? gscale[0] : 1.f;
  constexpr int NA_ = kBmNA, NT_ = kBmNT;
  BM_STAMP(0);
  // ---- every global load of the kernel up front (clamped indices, padding selected in when the value is stored).
  // sm.eps != NULL (the forward ran with defer_softmax): the Monte-Carlo softmax likelihood of this tile's columns is evaluated
  // HERE -- every (s, c, tile) workgroup redoes the softmax over all classes of its 64 columns (C-fold redundant, F C 64 exps)
  // and keeps the gradient of its own class; the class-0 workgroups also add the tile's share of nll.  One launch less on the
  // critical path of the step.  Its loads go FIRST (vmcnt retires in order: the evaluation then runs under the tile loads) and
  // wide: thread (column quad, likelihood sample) = (tid & 15, tid >> 4) takes one float4 of eps per class -- 18 loads per
  // thread (the first version, one column and F / 4 samples per thread, issued 96 scalar loads in front of the tile loads:
  // 11 us of the kernel's 39 went by before the first LDS store); mu and var are loaded once per workgroup (thread = (class,
  // quad)) and shared through LDS (in P's place, which is written after the barrier that ends the evaluation).
  const int q4 = tid & 15, fs = tid >> 4;
  const int nq = min(n0 + 4 * q4, B - 4);            // B % 4 == 0: a column quad lies wholly inside or outside
  float4 sev[kBmSmC], mu4 = make_float4(0.f, 0.f, 0.f, 0.f), var4 = mu4;
  int64_t yq[4] = {0, 0, 0, 0};
  if (sm.eps) {
    const int s = (int)(b / C);
    const int64_t rc = ((int64_t)s * C + min(fs, C - 1)) * B + nq;
    mu4 = *reinterpret_cast<const float4*>(sm.mu + rc);
    var4 = *reinterpret_cast<const float4*>(sm.var + rc);
    const float* ep = sm.eps + ((int64_t)s * sm.F + min(fs, sm.F - 1)) * C * B + nq;
#pragma unroll
    for (int c = 0; c < kBmSmC; ++c) sev[c] = *reinterpret_cast<const float4*>(ep + (int64_t)min(c, C - 1) * B);
#pragma unroll
    for (int j = 0; j < 4; ++j) yq[j] = sm.y[nq + j];
  }
  //      G, the P and W tiles now; T and the K_uf tile stay in registers until the first two products are done
  float4 rg[NA_], rp[NT_], rw[NT_], rt[NA_], rk[NT_];
  bm_load_mat<true>(Qb + 4, LD, M, tid, rg);          // G = T L_S: lower triangular
  bm_load_tile(Qb + NR, LD, M, n0, B, tid, rp);
  bm_load_tile(Wb, B, M, n0, B, tid, rw);
  const float av = tid < 128 ? Qb[(int64_t)min(tid, M - 1) * LD] : 0.f;
  const int ncl = min(n0 + (tid & 63), B - 1);
  float gmv = 0.f, gvv = 0.f;
  if (!sm.eps) { gmv = gmu[b * B + ncl]; gvv = gvar[b * B + ncl]; }
  bm_load_mat<true>(Tb, M, M, tid, rt);
  bm_load_tile(Kb, LD, M, n0, B, tid, rk);
  BM_STAMP(13);
  if (sm.eps) {
    const int cme = (int)(b % C);
    const float sc1 = 1.f / (float)(S * sm.F);
    float* smu = sP;                                  // [kBmSmC][64] mu, then [kBmSmC][64] sd = sqrt(var)
    float* ssd = sP + kBmSmC * 64;
    *reinterpret_cast<float4*>(&smu[fs * 64 + 4 * q4]) = mu4;
    *reinterpret_cast<float4*>(&ssd[fs * 64 + 4 * q4]) = make_float4(sqrtf(var4.x), sqrtf(var4.y), sqrtf(var4.z), sqrtf(var4.w));
    __syncthreads();
    float pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, ct[4] = {0.f, 0.f, 0.f, 0.f};
    if (4 * wave < sm.F) {                            // uniform: a wave holds four likelihood samples
      const bool live = n0 + 4 * q4 < B && fs < sm.F;
      const int yb[4] = {(int)yq[0], (int)yq[1], (int)yq[2], (int)yq[3]};
      float mx[4], fy[4] = {0.f, 0.f, 0.f, 0.f}, sdme[4] = {1.f, 1.f, 1.f, 1.f};
      float eme[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) mx[j] = -INFINITY;
      // v = mu + sd eps in place over eps (classes in groups of four, skipped by uniform branches beyond C; no loads inside)
#pragma unroll
      for (int cg = 0; cg < kBmSmC / 4; ++cg) {
        if (4 * cg >= C) continue;
#pragma unroll
        for (int c = 4 * cg; c < 4 * cg + 4; ++c) {
          const float4 m4 = *reinterpret_cast<const float4*>(&smu[c * 64 + 4 * q4]);
          const float4 s4 = *reinterpret_cast<const float4*>(&ssd[c * 64 + 4 * q4]);
          const float e[4] = {sev[c].x, sev[c].y, sev[c].z, sev[c].w};
          const float m_[4] = {m4.x, m4.y, m4.z, m4.w}, s_[4] = {s4.x, s4.y, s4.z, s4.w};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = c < C ? fmaf(s_[j], e[j], m_[j]) : -INFINITY;
            mx[j] = fmaxf(mx[j], v[j]);
            if (c == yb[j]) fy[j] = v[j];
            if (c == cme) { eme[j] = e[j]; sdme[j] = s_[j]; }
          }
          sev[c] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      float se[4] = {0.f, 0.f, 0.f, 0.f}, vme[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cg = 0; cg < kBmSmC / 4; ++cg) {
        if (4 * cg >= C) continue;
#pragma unroll
        for (int c = 4 * cg; c < 4 * cg + 4; ++c) {
          const float v[4] = {sev[c].x, sev[c].y, sev[c].z, sev[c].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = c < C ? expf(v[j] - mx[j]) : 0.f;
            se[j] += e;
            if (c == cme) vme[j] = e;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pc = vme[j] * (sc1 / se[j]) - (cme == yb[j] ? sc1 : 0.f);
        pm[j] = live ? pc : 0.f;
        pv[j] = live ? pc * eme[j] * (0.5f / sdme[j]) : 0.f;
        ct[j] = live ? -(fy[j] - (mx[j] + logf(se[j]))) * sc1 : 0.f;
      }
    }
    // the four samples of a wave meet by shuffles (lane = 16 (f % 4) + quad), the four waves in LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pm[j] += __shfl_xor(pm[j], 16, 64); pm[j] += __shfl_xor(pm[j], 32, 64);
      pv[j] += __shfl_xor(pv[j], 16, 64); pv[j] += __shfl_xor(pv[j], 32, 64);
      ct[j] += __shfl_xor(ct[j], 16, 64); ct[j] += __shfl_xor(ct[j], 32, 64);
    }
    BM_STAMP(14);
    float* sred = scs + 64 + 8;                       // [3][4][64]
    if (lane < 16) {
      *reinterpret_cast<float4*>(&sred[(0 * 4 + wave) * 64 + 4 * q4]) = make_float4(pm[0], pm[1], pm[2], pm[3]);
      *reinterpret_cast<float4*>(&sred[(1 * 4 + wave) * 64 + 4 * q4]) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(&sred[(2 * 4 + wave) * 64 + 4 * q4]) = make_float4(ct[0], ct[1], ct[2], ct[3]);
    }
    __syncthreads();
    BM_STAMP(15);
    if (tid < 64) {
      gmv = sred[tid] + sred[64 + tid] + sred[128 + tid] + sred[192 + tid];
      gvv = sred[256 + tid] + sred[320 + tid] + sred[384 + tid] + sred[448 + tid];
      if (cme == 0) {
        float t = sred[512 + tid] + sred[576 + tid] + sred[640 + tid] + sred[704 + tid];
        t = wave_sum(t);
        if (tid == 0) atomicAdd(sm.nll, t);
      }
    }
  }

  BM_STAMP(1);
  // ---- phase 0: G, P, W, a, gmu, gvar into LDS ------------------------------------------------------------------------------
  bm_store_mat<true>(sA, rg, M, tid);
  bm_store_tile(sP, rp, M, n0, B, tid);
  bm_store_tile(sW, rw, M, n0, B, tid);
  if (tid < 128) sa[tid] = tid < M ? av : 0.f;
  if (tid < 64) {
    const bool ok = n0 + tid < B;
    sgm[tid] = ok ? gs * gmv : 0.f;
    sgv[tid] = ok ? gs * gvv : 0.f;
    scs[tid] = 0.f;
  }
  __syncthreads();
  BM_STAMP(2);
  BM_CUT_AT(0);
  // ---- phase 1: gW = 2 W gvar in place; ga += P gmu; gkd += sum gvar ----------------------------------------------------------
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
    // thread (m, h): row m = tid / 2, columns [32 h, 32 h + 32)
    const int m = tid >> 1, h = tid & 1;
    const float* pr = sP + min(m, kBmKP - 1) * kBmST + 32 * h;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 p = *reinterpret_cast<const float4*>(pr + 4 * i);
      const float4 g = *reinterpret_cast<const float4*>(&sgm[32 * h + 4 * i]);
      acc0 = fmaf(p.x, g.x, acc0); acc1 = fmaf(p.y, g.y, acc1); acc0 = fmaf(p.z, g.z, acc0); acc1 = fmaf(p.w, g.w, acc1);
    }
    float t = acc0 + acc1;
    t += __shfl_xor(t, 1, 64);
    if (h == 0 && m < M) atomicAdd(&gQP[b * MLD + (int64_t)m * LD], t);
    if (wave == 0) {
      const float tv = wave_sum(sgv[lane]);
      if (lane == 0) atomicAdd(&gkd[b], tv);
    }
  }
  __syncthreads();
  BM_STAMP(3);
  BM_CUT_AT(1);
  // ---- phase 2: gG += tril(P gW^T)  (atomics),  gP = a gmu^T - 2 P gvar + G gW ------------------------------------------------
  const int cbh = wave & 1;
  const int rbs[2] = {(wave >> 1) ? 1 : 0, (wave >> 1) ? 2 : 3};
  bm_f32x16 accP[2], accG[3];
  float* dG = gQP + b * MLD + 4;
  if (wave == 0) bm_phase2<0>(sA, sP, sW, sa, sgm, sgv, dG, LD, M, li, lh, accP, accG);
  else if (wave == 1) bm_phase2<1>(sA, sP, sW, sa, sgm, sgv, dG, LD, M, li, lh, accP, accG);
  else if (wave == 2) bm_phase2<2>(sA, sP, sW, sa, sgm, sgv, dG, LD, M, li, lh, accP, accG);
  else bm_phase2<3>(sA, sP, sW, sa, sgm, sgv, dG, LD, M, li, lh, accP, accG);
  BM_STAMP(4);
  BM_CUT_AT(2);
  BM_STAMP(5);
  bmm_lds_barrier();                                // everybody is done with P, gW and G (LDS only: the atomics of phase 2 stay in flight)
  BM_STAMP(6);
  BM_CUT_AT(3);
  // ---- phase 3: gP into P's place, T into G's place, the K_uf tile into gW's place ---------------------------------------------
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < kBmKP) sP[m * kBmST + 32 * cbh + li] = m < M ? accP[u][r] : 0.f;
    }
  bm_store_mat<true>(sA, rt, M, tid);
  bm_store_tile(sW, rk, M, n0, B, tid);
  bmm_lds_barrier();
  BM_STAMP(7);
  BM_CUT_AT(4);
  // ---- phase 4: gT += tril(gP K_uf^T)  (atomics),  gK_uf = T^T gP,  W_uf = gK_uf o K_uf --------------------------------------
  {
    bm_f32x16 acc[2];      // [0]: row block rlo, [1]: row block rhi
    bm_f32x16 accT[3];
    float* dT = gTT + b * MM;
    if (wave == 0) bm_phase4<0>(sA, sP, sW, dT, M, M, li, lh, acc, accG, dG, LD, accT);
    else if (wave == 1) bm_phase4<1>(sA, sP, sW, dT, M, M, li, lh, acc, accG, dG, LD, accT);
    else if (wave == 2) bm_phase4<2>(sA, sP, sW, dT, M, M, li, lh, acc, accG, dG, LD, accT);
    else bm_phase4<3>(sA, sP, sW, dT, M, M, li, lh, acc, accG, dG, LD, accT);
    BM_STAMP(8);
    BM_CUT_AT(5);
    const int rlo = rbs[0], rhi = rbs[1];
    const int n = 32 * cbh + li;
    BM_STAMP(9);
    if (wave == 0) bm_tail_atomics<0>(accT, dT, M, M, li, lh);
    else if (wave == 1) bm_tail_atomics<1>(accT, dT, M, M, li, lh);
    else if (wave == 2) bm_tail_atomics<2>(accT, dT, M, M, li, lh);
    else bm_tail_atomics<3>(accT, dT, M, M, li, lh);
    // W_uf = gK_uf o K_uf: through LDS (gP's place, once every wave is done with gP, T and the K_uf tile as operands), so
    // that rows go out as coalesced float4 and the row / column sums are plain loops -- 160 dependent cross-lane shuffles
    // per wave for the row sums of the accumulator layout cost more than the rest of the kernel's epilogue together
    float wv[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rb = u ? rhi : rlo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
        wv[u][r] = m < M ? acc[u][r] * sW[min(m, kBmKP - 1) * kBmST + n] : 0.f;      // K_uf is zero in the columns past B
      }
    }
    // (LDS-only barriers from here on: __syncthreads() would drain the tail atomics issued just above -- they are left to
    //  finish under the epilogue; nothing below is handed over through global memory)
    bmm_lds_barrier();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rb = u ? rhi : rlo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < kBmKP) sP[m * kBmST + n] = wv[u][r];
      }
    }
  }
  BM_STAMP(10);
  bmm_lds_barrier();
  BM_STAMP(11);
  {
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
    // row sums: thread (m, h) = (tid / 2, tid % 2) sums columns [32 h, 32 h + 32) of row m
    const int m = tid >> 1, h = tid & 1;
    const float* pr = sP + min(m, kBmKP - 1) * kBmST + 32 * h;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(pr + 4 * i);
      a0 += v.x + v.z; a1 += v.y + v.w;
    }
    float t = a0 + a1;
    t += __shfl_xor(t, 1, 64);
    if (h == 0 && m < M) atomicAdd(&r_uf[b * M + m], t);          // b * M + m == s * C * M + c * M + m
  }
  bmm_lds_barrier();
  {
    const int s = (int)(b / C);
    float cv = 0.f;
    if (tid < 64) {
      cv = scs[tid];
      if (n0 + tid < B) atomicAdd(&c_uf[(int64_t)s * B + n0 + tid], cv);
    }
    if (wave == 0) {
      const float tot = wave_sum(cv);
      if (lane == 0) atomicAdd(&gtheta[(int64_t)s * (D + 1) + D], 2.f * tot);
    }
  }
  BM_STAMP(12);
}

}  // namespace vargp
