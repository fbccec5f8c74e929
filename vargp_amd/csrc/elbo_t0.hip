// The first-task ELBO as ONE native program: vargp_elbo_t0_fwd / vargp_elbo_t0_bwd sequence every kernel of
// VARGP.loss for a model without previous tasks (reference: var_gp/vargp.py:156-194, gp_utils.py:150-191,
// kernels.py:24-77, likelihoods.py:13-45) and of its gradient on one stream, over one caller-owned workspace.
//
// Why a program and not a composition of the per-op entry points: at the reference's sizes (M = 100 inducing points
// per class, minibatch 512) every kernel is short, so a step costs about (number of launches) x 5 us plus the GEMM and
// Cholesky time.  The program therefore
//   * concatenates everything that is multiplied by T = Lz^-1 into one operand
//         RK[s,c] = [ m | 0 0 0 | L_S | Lu | pad | K_uf[s,c] ]      (M x LD, LD = NR + B rounded up to 4, NR = 4 + 2M -> 4)
//     so that   QP = T RK = [ a | . | G | G2 | . | P ],   gT = tril(gQP RK^T)   and   gRK = T^T gQP
//     are ONE GEMM each (a = Lz^-1 m, G = Lz^-1 L_S, G2 = Lz^-1 Lu (KL), P = Lz^-1 K_uf);
//   * builds K_uu and K_uf with one two-problem GEMM launch (shared 1/sigma^2, gamma^2 and row norms), and their
//     backward products W.Y likewise;
//   * factorises K_uu (S*C matrices) and S_u = Lu Lu^T (C matrices) in one batch;
//   * folds the glue (hyper-parameter sampling and its KL, vec2tril, S_u, zero-fills, softmax likelihood with its
//     gradient, KL reductions, gradient unpacking) into a handful of multi-role kernels.
// About 30 launches per step instead of about 85; numerics are those of the per-op path (same kernels and formulas).
#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"
STEP_SPAN_TABLE(t0)
#include "elbo_shared.h"
#include "t0_bwd_mid.h"
#include "t0_bwd_mid_multi.h"
#include "chol_gram.h"
#include "t0_prologue.h"
#include "t0_bwd_tail.h"

namespace vargp {

constexpr int kKuuSplit = 2;   // K-splits of the K_uu distance GEMM (few workgroups, long K loop: split to use the chip;
                               // 2 / 3 / 4 splits at the BASELINE shape: 4064 / 3989 / 4003 steps/s)
#ifndef VARGP_KL_ROWS
#define VARGP_KL_ROWS 8
#endif
constexpr int kT0TileUnitsMax = 16384;   // (S C <= 2048 at the reference's batch of 512; mirrored by vargp_amd/vargp.py)
constexpr int kTailLdsDefault = 1;
constexpr int kKlRows = VARGP_KL_ROWS;     // rows of one (s, c) block per KL workgroup


struct T0Ws {
  // forward results kept for the backward
  float* queue;                    // 8 tile counters of the backward's merged launch (cleared by the forward)
  float *theta, *eps_theta, *eps_f, *w, *g2, *kd, *na, *nb, *xs, *Lu, *KS, *LL, *TT, *RK, *QP, *W, *mu, *var;
  float *gmu, *gvar;               // accumulators zeroed by the forward prologue (softmax gradient, unscaled)
  float *r_uf, *c_uf, *gtheta;     // accumulators zeroed by the first backward kernel
  float *r_uu, *gW, *gkd, *gQP, *gLL, *gTT, *gRK, *gKS, *Wuu, *Puu, *Puf;
  float* kpart;                    // split-K partial inner products of K_uu (forward only: aliases gRK)
  void* chol;
  size_t chol_bytes;
  int NR, LD;
  int64_t Dp;
  size_t bytes;
};

static T0Ws carve_t0(void* ws, int S, int C, int M, int D, int B, int F) {
  T0Ws o{};
  o.NR = (int)round_up(4 + 2 * M, 4);
  o.LD = (int)round_up(o.NR + B, 4);
  o.Dp = round_up(D, 4);
  const int64_t SC = (int64_t)S * C, MM = (int64_t)M * M, D1 = D + 1;
  float* p = reinterpret_cast<float*>(ws);
  auto take = [&](int64_t n) { float* q = p; p += round_up(n, 64); return q; };
  // theta, eps_theta, eps_f first, in this order (vargp_amd/fused.py exposes them as views)
  o.theta = take(S * D1); o.eps_theta = take(S * D1); o.eps_f = take((int64_t)S * F * C * B);
  o.w = take(S * o.Dp); o.g2 = take(S); o.kd = take(SC); o.queue = take(8);
  o.na = take(SC * M); o.nb = take((int64_t)S * B);
  o.xs = take((int64_t)S * B * D);          // x o 1/sigma_s^2 (written by the norm role of the front launch)
  o.Lu = take(C * MM); o.KS = take((SC + C) * MM); o.LL = take((SC + C) * MM); o.TT = take((SC + C) * MM);
  o.RK = take(SC * M * o.LD); o.QP = take(SC * M * o.LD); o.W = take(SC * M * B);
  o.mu = take(SC * B); o.var = take(SC * B);
  o.gmu = take(SC * B); o.gvar = take(SC * B);                                    // adjacent: one zero range
  o.r_uf = take(SC * M); o.c_uf = take((int64_t)S * B); o.gtheta = take(S * D1);  // adjacent: one zero range
  o.r_uu = take(SC * M); o.gW = take(SC * M * B); o.gkd = take(SC);
  o.gQP = take(SC * M * o.LD); o.gLL = take((SC + C) * MM); o.gTT = take((SC + C) * MM);
  o.gRK = take(SC * M * o.LD); o.gKS = take((SC + C) * MM); o.Wuu = take(SC * MM);
  o.Puu = take(SC * M * D); o.Puf = take(SC * M * D);
  o.kpart = o.gRK;                 // kKuuSplit * SC * M * M <= SC * M * LD floats is checked where it is used
  const size_t cb = vargp_chol_workspace_bytes((int)(SC + C), M, 0), cbb = vargp_chol_workspace_bytes((int)(SC + C), M, 1);
  o.chol_bytes = cb > cbb ? cb : cbb;
  o.chol = p;
  p += round_up((int64_t)(o.chol_bytes + 3) / 4, 64);
  o.bytes = (size_t)((char*)p - (char*)ws);
  return o;
}

// ---------------------------------------------------------------------------------------------------------------
// forward kernels
// ---------------------------------------------------------------------------------------------------------------
// the prologue roles alone (t0_prologue.h; shapes whose K_uu product does not share the launch)
__global__ __launch_bounds__(256) void t0_prologue_kernel(const ProArgs a) {
  __shared__ float red[4];
  t0_prologue_body(a, blockIdx.x, red);
}

// weighted squared row norms of the inducing points (na) and of the minibatch (nb), one wave per row; grid (rows/4, S)
__global__ __launch_bounds__(256) void t0_norm_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                      const float* __restrict__ w, float* __restrict__ na,
                                                      float* __restrict__ nb, int64_t zrows, int64_t xrows, int D,
                                                      int64_t Dp) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int s = blockIdx.y, lane = threadIdx.x & 63;
  if (row >= zrows + xrows) return;
  const bool isz = row < zrows;
  const float* xr = isz ? z + row * D : x + (row - zrows) * D;
  const float* ws = w + s * Dp;
  const float acc = row_norm_wave(xr, ws, D, lane);
  if (lane == 0) {
    if (isz) na[(int64_t)s * zrows + row] = acc; else nb[(int64_t)s * xrows + (row - zrows)] = acc;
  }
}

// Epilogue of the K-split K_uu product and the row norms in one launch (they do not depend on each other: the norms
// the K_uu distances need are the diagonal of the weighted Gram matrix G = sum of the partial products itself).
//   blocks < ncomb: K_uu[s,c,i,j] = g2 exp(-(G_ii + G_jj - 2 G_ij) / 2), exactly g2 on the diagonal; one thread per entry
//   rest          : na, nb of t0_norm_kernel (for the K_uf epilogue), one wave per row and hyper-sample
__global__ __launch_bounds__(256) void t0_combine_norm_kernel(const float* __restrict__ part, int nsplit, int64_t sSplit,
                                                              const float* __restrict__ g2, float* __restrict__ Kuu, int C,
                                                              int M, int64_t total, int ncomb, const float* __restrict__ z,
                                                              const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ na, float* __restrict__ nb,
                                                              int64_t zrows, int64_t xrows, int D, int64_t Dp, int nrow4) {
  if ((int)blockIdx.x < ncomb) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = e % M, i = (e / M) % M;
    const int64_t b = e / ((int64_t)M * M);        // s * C + c
    const int64_t base = b * M * M;
    float gij = 0.f, gii = 0.f, gjj = 0.f;
    for (int k0 = 0; k0 < nsplit; k0 += 4) {       // four splits' loads in flight together
      float vij[4], vii[4], vjj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* pk = part + min(k0 + q, nsplit - 1) * sSplit + base;
        vij[q] = pk[(int64_t)i * M + j]; vii[q] = pk[(int64_t)i * M + i]; vjj[q] = pk[(int64_t)j * M + j];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k0 + q < nsplit) { gij += vij[q]; gii += vii[q]; gjj += vjj[q]; }
      }
    }
    const float gam = g2[b / C];
    Kuu[e] = i == j ? gam : gam * expf(-0.5f * (gii + gjj - 2.f * gij));
    return;
  }
  const int id = (int)blockIdx.x - ncomb;
  const int64_t row = (int64_t)(id % nrow4) * 4 + (threadIdx.x >> 6);
  const int s = id / nrow4, lane = threadIdx.x & 63;
  if (row >= zrows + xrows) return;
  const bool isz = row < zrows;
  const float* xr = isz ? z + row * D : x + (row - zrows) * D;
  const float* ws = w + s * Dp;
  const float acc = row_norm_wave(xr, ws, D, lane);
  if (lane == 0) {
    if (isz) na[(int64_t)s * zrows + row] = acc; else nb[(int64_t)s * xrows + (row - zrows)] = acc;
  }
}

// RK[s,c,i, 0:NR] = [ m_ci | 0 0 0 | LS_c[i,:] | Lu_c[i,:] | 0.. ]  for every s (the K_uf block is written by the GEMM)
__global__ void t0_pack_kernel(const float* __restrict__ m, const float* __restrict__ LS, const float* __restrict__ Lu,
                               float* __restrict__ RK, int C, int M, int NR, int LD, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int col = e % NR;
  const int64_t row = e / NR;            // (s * C + c) * M + i
  const int i = row % M;
  const int64_t c = (row / M) % C;
  float v = 0.f;
  if (col == 0) v = m[c * M + i];
  else if (col >= 4 && col < 4 + M) v = LS[(c * M + i) * M + (col - 4)];
  else if (col >= 4 + M && col < 4 + 2 * M) v = Lu[(c * M + i) * M + (col - 4 - M)];
  RK[row * LD + col] = v;
}

// Two roles.  Blocks < npd: predictive mean / variance (gp_utils.py:178-186), 64 minibatch columns x 4 row lanes:
//   mu = sum_m P a,  var = kd - sum_m P^2 + sum_m W^2.   Blocks >= npd: MVN-KL of q(u) = N(m, Lu Lu^T) against
//   p(u) = N(0, Lz Lz^T) (vargp.py:182-190) from the a and G2 columns of QP, kKlRows rows of one (s, c) per block:
//   kl[s,c] = sum log diag Lz - sum log diag Lu + 0.5 (|G2|_F^2 + |a|^2 - M),  kl_u = (1/S) sum kl[s,c]  (atomic)
__global__ __launch_bounds__(256) void t0_pdiag_kl_fwd_kernel(const float* __restrict__ QP, const float* __restrict__ W,
                                                              const float* __restrict__ kd, const float* __restrict__ Lz,
                                                              const float* __restrict__ Lu, float* __restrict__ mu,
                                                              float* __restrict__ var, float* __restrict__ kl_u, int S,
                                                              int C, int M, int B, int NR, int LD, int nbx, int npd,
                                                              int nkx, uint32_t* rng_counter) {
  __shared__ float red[3][4][64];
  // the noise of this step has been drawn (by the prologue, two launches back): advance the generator's step count
  if (rng_counter && blockIdx.x == 0 && threadIdx.x == 0) rng_counter[0] += 1u;
  if ((int)blockIdx.x < npd) {
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = ((int)blockIdx.x % nbx) * 64 + cx;
    const int64_t b = blockIdx.x / nbx;
    float m0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (col < B) {
      const float* q = QP + b * M * LD;
      const float* p = q + NR + col;
      const float* w = W + b * M * B + col;
#pragma unroll 4
      for (int m = ry; m < M; m += 4) {
        const float pv = p[(int64_t)m * LD], wv = w[(int64_t)m * B];
        m0 = fmaf(pv, q[(int64_t)m * LD], m0);
        d1 = fmaf(pv, pv, d1);
        d2 = fmaf(wv, wv, d2);
      }
    }
    red[0][ry][cx] = m0; red[1][ry][cx] = d1; red[2][ry][cx] = d2;
    __syncthreads();
    if (ry == 0 && col < B) {
      m0 = red[0][0][cx] + red[0][1][cx] + red[0][2][cx] + red[0][3][cx];
      d1 = red[1][0][cx] + red[1][1][cx] + red[1][2][cx] + red[1][3][cx];
      d2 = red[2][0][cx] + red[2][1][cx] + red[2][2][cx] + red[2][3][cx];
      mu[b * B + col] = m0;
      var[b * B + col] = kd[b] - d1 + d2;
    }
    return;
  }
  const int id = (int)blockIdx.x - npd;
  const int64_t b = id / nkx;          // s * C + c
  const int c = b % C;
  const int i0 = (id % nkx) * kKlRows, i1 = min(M, i0 + kKlRows);
  const float* q = QP + b * M * LD;
  float acc = 0.f;
  for (int e = threadIdx.x; e < (i1 - i0) * M; e += 256) {
    const int i = i0 + e / M, j = e % M;
    if (j <= i) { const float v = q[(int64_t)i * LD + 4 + M + j]; acc = fmaf(v, v, acc); }
  }
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    const float a = q[(int64_t)i * LD];
    acc = fmaf(a, a, acc);
    acc += 2.f * (logf(Lz[(b * M + i) * M + i]) - logf(Lu[((int64_t)c * M + i) * M + i])) - 1.f;
  }
  const float t = block_sum<256>(acc, &red[0][0][0]);
  if (threadIdx.x == 0) atomicAdd(kl_u, 0.5f * t / (float)S);
}

// ---------------------------------------------------------------------------------------------------------------
// The middle of the forward as ONE LDS-resident MFMA kernel (M <= 104): for every (s, c) and 64-column tile of the minibatch
//     P = T K_uf[:, tile]   (T = Lz^-1, lower)        W = G^T P   (G = Lz^-1 L_S, lower)
//     mu = sum_m P a,  var = kd - sum_m P^2 + sum_m W^2                                   (gp_utils.py:178-186)
// T, G (from the small-column product QP[:, 0:NR] = T RK[:, 0:NR], computed just before) and the K_uf tile sit in LDS for
// the whole workgroup (55 + 55 + 28 KB of the CU's 160 KB), P goes back into the tile's place as the second product's
// operand, the column reductions run on the accumulators: three launches (QP GEMM, W GEMM, moments) and two HBM round trips
// of P and W become one launch.  256 threads = 4 waves; wave w owns the 32-column half (w & 1) and the row blocks {0, 3}
// or {1, 2} -- with the triangular K ranges ([0, 32 rb + 32) for T, [32 rb, M) for G^T) both pairs carry the same work.
// f32 MFMA 32x32x2 with the k-pairing of gemm.hip: half-wave h supplies k = 8 g + 4 h + j, j < 4, one b128 fragment read
// per lane and group for the K-contiguous operand (T), four b32 reads for the row-major ones.
// The tile workgroups of an (s, c) share out the rows of the KL of q(u) against p(u) (vargp.py:182-190).
constexpr int kFusedTS = 108;    // row stride of T in LDS: 108 mod 64 = 44 -> the 16 rows of a b128 read group hit 16 bank quads
constexpr int kFusedGS = 132;    // row stride of G (k-major)
constexpr int kFusedKS = 68;     // row stride of the K_uf / P tile
constexpr int kFusedK = 104;     // padded inner dimension (M <= 104, multiple of 8)
constexpr size_t kFusedLdsBytes = sizeof(float) * (128 * kFusedTS + kFusedK * kFusedGS + kFusedK * kFusedKS + 128 + 3 * 64);
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#ifdef FF_STAMPS   // tuning builds only: s_memtime of workgroup 0 (lane 0 of each wave) after each phase (tests/native/bm_stamps.py ff)
__device__ unsigned long long g_ff_stamps[4][16];
extern "C" void vargp_debug_ff_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ff_stamps), sizeof(g_ff_stamps)); }
#ifndef FF_STAMP_BLOCK
#define FF_STAMP_BLOCK 0u      // which workgroup stamps (-DFF_STAMP_BLOCK=...: a late one finds the kernel's code in the instruction cache)
#endif
#define FF_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == (FF_STAMP_BLOCK)) g_ff_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

// The two products of t0_fwd_fused_kernel for the wave pair that owns row blocks R0 < R1 ({0, 3} or {1, 2}): every k-group
// range is a compile-time constant, the loops are fully unrolled and the next group's fragments are requested before the current
// group's MFMAs (one wave per SIMD: nothing else hides the LDS latency -- as run-time loops the products took 6.1-7.2k cycles
// for 3.6-4.4k of MFMAs).  The two blocks' MFMAs alternate while both are active (two independent accumulators).
__device__ __forceinline__ void ff_mfma4x2(f32x16_t& x, f32x16_t& y, const float4 a, const float4 c, const float4 b) {
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, x, 0, 0, 0);
  y = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, b.x, y, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, x, 0, 0, 0);
  y = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, b.y, y, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, x, 0, 0, 0);
  y = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, b.z, y, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, x, 0, 0, 0);
  y = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, b.w, y, 0, 0, 0);
}
__device__ __forceinline__ void ff_mfma4(f32x16_t& x, const float4 a, const float4 b) {
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, x, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, x, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, x, 0, 0, 0);
  x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, x, 0, 0, 0);
}
// P = T K: row block rb needs k < 32 rb + 32 (T lower triangular).  acc[0]: block R0, acc[1]: block R1
template <int R0, int R1>
__device__ __forceinline__ void ff_product_p(f32x16_t (&acc)[2], const float* __restrict__ sT, const float* __restrict__ sK, int cb,
                                             int li, int lh) {
  constexpr int GE = kFusedK / 8, G0 = bm_min(GE, 4 * R0 + 4), G1 = bm_min(GE, 4 * R1 + 4);      // G0 <= G1
  const float* arow0 = sT + (32 * R0 + li) * kFusedTS + 4 * lh;
  const float* arow1 = sT + (32 * R1 + li) * kFusedTS + 4 * lh;
  const float* bcol = sK + (4 * lh) * kFusedKS + 32 * cb + li;
  float4 a0 = bm_frag_kc(arow0, 0), a1 = bm_frag_kc(arow1, 0), bb = bm_frag_km(bcol, 0, kFusedKS);
  bm_for<0, G1>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 c0 = a0, c1 = a1, cbv = bb;
    if constexpr (g + 1 < G1) {
      if constexpr (g + 1 < G0) a0 = bm_frag_kc(arow0, 8 * (g + 1));
      a1 = bm_frag_kc(arow1, 8 * (g + 1));
      bb = bm_frag_km(bcol, 8 * (g + 1), kFusedKS);
    }
    if constexpr (g < G0) ff_mfma4x2(acc[0], acc[1], c0, c1, cbv);
    else ff_mfma4(acc[1], c1, cbv);
  });
}
// W = G^T P: row block rb needs k >= 32 rb (G lower triangular).  acc[0]: block R0 (starts alone), acc[1]: block R1
template <int R0, int R1>
__device__ __forceinline__ void ff_product_w(f32x16_t (&acc)[2], const float* __restrict__ sG, const float* __restrict__ sK, int cb,
                                             int li, int lh) {
  constexpr int GE = kFusedK / 8, S0 = 4 * R0, S1 = 4 * R1;                                         // S0 < S1 < GE
  const float* acol0 = sG + (4 * lh) * kFusedGS + 32 * R0 + li;
  const float* acol1 = sG + (4 * lh) * kFusedGS + 32 * R1 + li;
  const float* bcol = sK + (4 * lh) * kFusedKS + 32 * cb + li;
  float4 a0 = bm_frag_km(acol0, 8 * S0, kFusedGS), a1 = a0, bb = bm_frag_km(bcol, 8 * S0, kFusedKS);
  bm_for<S0, GE>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 c0 = a0, c1 = a1, cbv = bb;
    if constexpr (g + 1 < GE) {
      a0 = bm_frag_km(acol0, 8 * (g + 1), kFusedGS);
      if constexpr (g + 1 >= S1) a1 = bm_frag_km(acol1, 8 * (g + 1), kFusedGS);
      bb = bm_frag_km(bcol, 8 * (g + 1), kFusedKS);
    }
    if constexpr (g >= S1) ff_mfma4x2(acc[0], acc[1], c0, c1, cbv);
    else ff_mfma4(acc[0], c0, cbv);
  });
}

// MULTI (throughput-bound shapes: more (s, c, tile) units than the chip has CUs): a workgroup takes `ntile / nparts` consecutive
// tiles of its (s, c) -- T, G and a are staged ONCE per workgroup instead of once per tile (80 of the 106 KB a tile workgroup
// pulls), and the next tile's K_uf loads are in flight under the current tile's products.  nparts = workgroups per (s, c)
// (the single-tile form is nparts == ntile).
// barrier of the tile loop: the multi-tile form hands nothing over through global memory inside it, so it waits for LDS
// operations only and leaves the next tile's loads and this tile's P / W stores in flight (t0_bwd_mid_multi.h)
template <bool MULTI>
__device__ __forceinline__ void ff_barrier() {
  if constexpr (MULTI) bmm_lds_barrier(); else __syncthreads();
}
template <bool VEC4, bool MULTI = false>
__global__ __launch_bounds__(256) void t0_fwd_fused_kernel(const float* __restrict__ TT, float* __restrict__ QP,
                                                           const float* __restrict__ RK, float* __restrict__ W,
                                                           const float* __restrict__ kd, const float* __restrict__ Lz,
                                                           const float* __restrict__ Lu, float* __restrict__ mu,
                                                           float* __restrict__ var, float* __restrict__ kl_u, int S, int C,
                                                           int M, int B, int NR, int LD, int ntile, uint32_t* rng_counter,
                                                           int nparts) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  STEP_SPAN(t0, 3);
  float* sT = lds_f;                              // [128][TS]   T[i][k]
  float* sG = sT + 128 * kFusedTS;                // [K][GS]     G[k][i]
  float* sK = sG + kFusedK * kFusedGS;            // [K][KS]     K_uf tile [k][n], then P[m][n]
  float* sa = sK + kFusedK * kFusedKS;            // [128]       a = Lz^-1 m
  float* red = sa + 128;                          // [3][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  // 1-D grid, XCD-aware (as t0_bwd_mid_kernel): XCD x works through the matrices b = x, x + 8, ..., all tiles of one before the
  // next, so that the tiles of an (s, c) find its T and G in ONE L2 (grid = 8 ceil(SC / 8) ntile; the surplus exits)
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int64_t b = (int64_t)(idx / nparts) * 8 + xcd;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  if (rng_counter && blockIdx.x == 0 && tid == 0) rng_counter[0] += 1u;   // this step's noise has been drawn
  if (b >= (int64_t)S * C) return;
  const int part = idx % nparts;
  // this workgroup's tiles [tile_x, tile_end) of the (s, c) (single-tile form: one)
  int tile_x = MULTI ? (part * ntile) / nparts : part;
  const int tile_end = MULTI ? ((part + 1) * ntile) / nparts : part + 1;
  int n0 = tile_x * 64;
  // ---- stage T, G, the K_uf tile and a (zero-padded: rows / inner indices >= M, columns >= B) ----------------------------
  const float* Tb = TT + b * MM;
  const float* Qb = QP + b * MLD;
  const float* Kb = RK + b * MLD + NR;          // (+ n0: per tile)
  FF_STAMP(0);
  FF_STAMP(1);      // (the KL's loads are part of the staging phase now)
  // ---- every global load first: T, G, the K_uf tile, a, and the first round of the KL's operands; then the LDS stores; then
  //      the KL arithmetic.  (One round trip instead of two, and no faster: 15.5k cycles against 4.3k + 10.7k -- the front of this
  //      kernel is bound by the 24 MB the 240 workgroups pull from memory, T and G eight times each, not by round trips.)
  // (branch-free: a branch around a load makes the compiler wait for every load before issuing the next; indices are
  //  clamped into the operand and the padding is selected in afterwards, so that each loop issues its loads back to back)
  constexpr int NT_ = 128 * (kFusedK / 4) / 256;          // 13 float4 per thread
  constexpr int NG_ = kFusedK * 32 / 256;                 // 13
  constexpr int NK_ = (kFusedK * 16 + 255) / 256;         // 7 (the last one partly out of range)
  float4 rt[NT_], rg[NG_], rk[NK_];
  // (32-bit BYTE offsets from uniform bases, as t0_bwd_common.h: bm_load_mat -- with 64-bit element indices every load carried a
  // v_mad_i64 chain, and with one wave per SIMD the 45 loads of this front took 10.8k cycles just to ISSUE)
  const char* Tbb = reinterpret_cast<const char*>(Tb);
  const char* Qbb = reinterpret_cast<const char*>(Qb);
  const char* Kbb = reinterpret_cast<const char*>(Kb);
#pragma unroll
  for (int u = 0; u < NT_; ++u) {
    const int e = tid + 256 * u;
    const int i = e / (kFusedK / 4), k = (e - i * (kFusedK / 4)) * 4;
    const int ic = min(i, M - 1);      // (T is lower triangular: a float4 wholly above the diagonal is not fetched, bm_load_mat<true>)
    rt[u] = *reinterpret_cast<const float4*>(Tbb + 4u * (__umul24((unsigned)ic, (unsigned)M) + (unsigned)min(k, min(M - 4, ic & ~3))));
  }
#pragma unroll
  for (int u = 0; u < NG_; ++u) {
    const int e = tid + 256 * u;
    const int k = e >> 5, i = (e & 31) * 4;
    const int kc = min(k, M - 1);      // (G = T L_S likewise: G[k][i] = 0 for i > k)
    rg[u] = *reinterpret_cast<const float4*>(Qbb + 4u * (__umul24((unsigned)kc, (unsigned)LD) + 4u + (unsigned)min(i, min(M - 4, kc & ~3))));
  }
  // K_uf tile.  B % 4 == 0 (VEC4; every BASELINE shape): a float4 lies wholly inside or outside the matrix, so the column is
  // clamped and the padding selected in when the value is stored -- NO branch: the previous form, `if (full) float4 else
  // per-element`, left a branch around every one of the seven loads, and the compiler put an s_waitcnt vmcnt(0) at each join:
  // seven memory round trips in a row, each also waiting for the 26 loads of T and G in front of it (10.5k of this kernel's
  // 36k cycles went by before the first LDS store).  Other B: the per-element form, in its own instantiation.
  auto load_ktile = [&](const int n0_) {
    const char* Kt = Kbb + 4 * n0_;
#pragma unroll
    for (int u = 0; u < NK_; ++u) {
      const int e = min(tid + 256 * u, kFusedK * 16 - 1);
      const int k = e >> 4, n = (e & 15) * 4;
      const unsigned ro = 4u * __umul24((unsigned)min(k, M - 1), (unsigned)LD);
      if constexpr (VEC4) {
        rk[u] = *reinterpret_cast<const float4*>(Kt + ro + 4u * (unsigned)min(n, B - 4 - n0_));
      } else {
        const float* src = reinterpret_cast<const float*>(Kt + ro);
        rk[u].x = n0_ + n < B ? src[n] : 0.f;         rk[u].y = n0_ + n + 1 < B ? src[n + 1] : 0.f;
        rk[u].z = n0_ + n + 2 < B ? src[n + 2] : 0.f; rk[u].w = n0_ + n + 3 < B ? src[n + 3] : 0.f;
      }
    }
  };
  load_ktile(n0);
  const float av = tid < 128 ? *reinterpret_cast<const float*>(Qbb + 4u * __umul24((unsigned)min(tid, M - 1), (unsigned)LD)) : 0.f;
  // ---- KL of q(u) against p(u) for this (s, c) (vargp.py:182-190), its rows shared out over the tile workgroups:
  //      kl[s,c] = sum log diag Lz - sum log diag Lu + (|G2|_F^2 + |a|^2 - M) / 2,   kl_u = (1/S) sum kl[s,c]
  const int c_kl = b % C;
  const int per = (M + nparts - 1) / nparts, i0 = part * per, i1 = min(M, i0 + per);
  const int nkl = max((i1 - i0) * M, 0);
  const unsigned mdiv = ((1u << 20) + (unsigned)M - 1u) / (unsigned)M;      // (uniform: one scalar division instead of eight per thread)
  float kv[8];                                            // first round of G2 entries (clamped; eight loads in flight per round)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = min(tid + 256 * u, max(nkl - 1, 0));
    const int ei = (int)(((unsigned)e * mdiv) >> 20);     // e / M (exact: e < 2^11, M <= 104, mdiv = ceil(2^20 / M))
    kv[u] = *reinterpret_cast<const float*>(Qbb + 4u * (__umul24((unsigned)min(i0 + ei, M - 1), (unsigned)LD) + 4u + (unsigned)M + (unsigned)(e - ei * M)));
  }
  // the diagonal terms: thread t < i1 - i0 takes row i0 + t (rows beyond: clamped loads, masked)
  const int idg = min(i0 + tid, M - 1);
  const float dga = *reinterpret_cast<const float*>(Qbb + 4u * __umul24((unsigned)idg, (unsigned)LD));
  const float dlz = Lz[(b * M + idg) * M + idg], dlu = Lu[((int64_t)c_kl * M + idg) * M + idg];
  FF_STAMP(10);
  // ---- LDS stores (zero-padded: rows / inner indices >= M, columns >= B)
#pragma unroll
  for (int u = 0; u < NT_; ++u) {
    const int e = tid + 256 * u;
    const int i = e / (kFusedK / 4), k = (e - i * (kFusedK / 4)) * 4;
    const bool ok = i < M && k < M && k <= i;
    *reinterpret_cast<float4*>(&sT[i * kFusedTS + k]) = ok ? rt[u] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  FF_STAMP(11);
#pragma unroll
  for (int u = 0; u < NG_; ++u) {
    const int e = tid + 256 * u;
    const int k = e >> 5, i = (e & 31) * 4;
    *reinterpret_cast<float4*>(&sG[k * kFusedGS + i]) = (k < M && i < M && i <= k) ? rg[u] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid < 128) sa[tid] = tid < M ? av : 0.f;
  if (tid < 192) red[tid] = 0.f;
  // ---- KL arithmetic (further rounds of loads only when a workgroup's share exceeds 2048 entries)
  float kl_acc = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + 256 * u;
    const int ei = (int)(((unsigned)e * mdiv) >> 20);
    const int i = i0 + ei, j = e - ei * M;
    kl_acc = fmaf((e < nkl && j <= i) ? kv[u] : 0.f, kv[u], kl_acc);      // upper entries are stored zeros
  }
  for (int e0 = 8 * 256; e0 < nkl; e0 += 8 * 256) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = min(e0 + tid + 256 * u, nkl - 1);
      kv[u] = Qb[(int64_t)(i0 + e / M) * LD + 4 + M + e % M];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + tid + 256 * u;
      const int i = i0 + e / M, j = e % M;
      kl_acc = fmaf((e < nkl && j <= i) ? kv[u] : 0.f, kv[u], kl_acc);
    }
  }
  {
    const bool dok = i0 + tid < i1;
    float t = fmaf(dga, dga, 2.f * (logf(dlz) - logf(dlu)) - 1.f);
    kl_acc += dok ? t : 0.f;
    for (int i = i0 + tid + 256; i < i1; i += 256) {     // (more than 256 rows per workgroup: never at M <= 104)
      const float a = Qb[(int64_t)i * LD];
      kl_acc = fmaf(a, a, kl_acc);
      kl_acc += 2.f * (logf(Lz[(b * M + i) * M + i]) - logf(Lu[((int64_t)c_kl * M + i) * M + i])) - 1.f;
    }
  }
  // ---- the tile loop (single-tile form: one pass, straight-line code)
  do {
#pragma unroll
  for (int u = 0; u < NK_; ++u) {
    const int e = tid + 256 * u;
    if (e < kFusedK * 16) {
      const int k = e >> 4, n = (e & 15) * 4;
      *reinterpret_cast<float4*>(&sK[k * kFusedKS + n]) = (k < M && (!VEC4 || n0 + n < B)) ? rk[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  FF_STAMP(12);
  // (MULTI) the next tile's K_uf loads: in flight under this tile's products
  if constexpr (MULTI) { if (tile_x + 1 < tile_end) load_ktile(n0 + 64); }
  FF_STAMP(13);
  ff_barrier<MULTI>();
  FF_STAMP(2);
  if constexpr (MULTI) {
    // ---- throughput-bound shapes: f32 MFMA 16x16x4 blocks (t0_bwd_mid_multi.h: M = 100 is seven 16-row blocks, 1.25x padded work
    // instead of 1.64x), 16 columns of the tile per wave, all seven row blocks: both products carry the same work on every wave
    const int l16 = lane & 15, q = lane >> 4, n = 16 * wave + l16, col = n0 + n;
    bm_f32x4 accP[kB16NB];
#pragma unroll
    for (int i = 0; i < kB16NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) accP[i][r] = 0.f;
    b16_prod<true, kFusedTS>(sT, sK, wave, l16, q, accP);        // P = T K: row block i takes the k-groups g <= i
    ff_barrier<true>();                               // everybody is done with the K_uf tile
    float s_mu = 0.f, s_p2 = 0.f, s_w2 = 0.f;
    char* Pout_b = reinterpret_cast<char*>(QP + b * MLD + NR);
#pragma unroll
    for (int i = 0; i < kB16NB; ++i) {
      float sav[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sav[r] = sa[16 * i + 4 * q + r];      // (128 entries, zero beyond M)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * q + r;
        const float v = m < M ? accP[i][r] : 0.f;       // (rows 104 .. 111 hold clamped reads of row 103)
        if (m < kFusedK) sK[m * kFusedKS + n] = v;
        if (m < M && col < B) *reinterpret_cast<float*>(Pout_b + 4u * (__umul24((unsigned)m, (unsigned)LD) + (unsigned)col)) = v;
        s_mu = fmaf(v, sav[r], s_mu);
        s_p2 = fmaf(v, v, s_p2);
      }
    }
    ff_barrier<true>();
    char* Wb_b = reinterpret_cast<char*>(W + b * (int64_t)M * B);
    bm_f32x4 accW[kB16NB];
#pragma unroll
    for (int i = 0; i < kB16NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) accW[i][r] = 0.f;
    b16_prod<false, kFusedGS>(sG, sK, wave, l16, q, accW);       // W = G^T P: row block i takes the k-groups g >= i
#pragma unroll
    for (int i = 0; i < kB16NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + 4 * q + r;
        if (m < M) {
          const float v = accW[i][r];
          if (col < B) *reinterpret_cast<float*>(Wb_b + 4u * (__umul24((unsigned)m, (unsigned)B) + (unsigned)col)) = v;
          s_w2 = fmaf(v, v, s_w2);
        }
      }
    // column reductions: the four lane groups hold different rows of the same column; a column belongs to ONE wave
    s_mu += __shfl_xor(s_mu, 16, 64); s_p2 += __shfl_xor(s_p2, 16, 64); s_w2 += __shfl_xor(s_w2, 16, 64);
    s_mu += __shfl_xor(s_mu, 32, 64); s_p2 += __shfl_xor(s_p2, 32, 64); s_w2 += __shfl_xor(s_w2, 32, 64);
    if (q == 0) { red[n] = s_mu; red[64 + n] = s_p2; red[128 + n] = s_w2; }
    ff_barrier<true>();
    if (tid < 64 && n0 + tid < B) {
      mu[b * B + n0 + tid] = red[tid];
      var[b * B + n0 + tid] = kd[b] - red[64 + tid] + red[128 + tid];
    }
    tile_x += 1; n0 += 64;
    // (the next pass writes red only after two more barriers; sK after the one above)
  } else {
  const int cb = wave & 1;
  const int rbs[2] = {(wave >> 1) ? 1 : 0, (wave >> 1) ? 2 : 3};
  // ---- P = T K: row block rb needs k < 32 rb + 32.  The wave's two blocks advance together (two independent accumulators:
  // back-to-back MFMAs on ONE accumulator wait for each other), the longer one finishes alone.
  f32x16_t accP[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[u][r] = 0.f;
  if ((wave >> 1) == 0) ff_product_p<0, 3>(accP, sT, sK, cb, li, lh);
  else ff_product_p<1, 2>(accP, sT, sK, cb, li, lh);
  FF_STAMP(3);
  ff_barrier<MULTI>();                                // everybody is done with the K_uf tile
  FF_STAMP(4);
  // P into the tile's place (second product's operand) and out to QP; column sums of P a and P^2 on the way
  const int col = n0 + 32 * cb + li;
  float s_mu = 0.f, s_p2 = 0.f, s_w2 = 0.f;
  char* Pout_b = reinterpret_cast<char*>(QP + b * MLD + NR);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    // a[m] for the block's 16 rows first, unconditionally (sa holds 128 entries, zero beyond M; rows of P beyond M are zero
    // as well: T is zero-padded) -- an LDS read inside `if (m < M)` is waited for element by element
    float sav[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sav[r] = sa[32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * rbs[u] + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = accP[u][r];
      if (m < kFusedK) sK[m * kFusedKS + 32 * cb + li] = v;
      if (m < M && col < B) *reinterpret_cast<float*>(Pout_b + 4u * (__umul24((unsigned)m, (unsigned)LD) + (unsigned)col)) = v;   // (32-bit byte offset)
      s_mu = fmaf(v, sav[r], s_mu);
      s_p2 = fmaf(v, v, s_p2);
    }
  }
  ff_barrier<MULTI>();
  FF_STAMP(5);
  // ---- W = G^T P: row block rb needs k >= 32 rb (G is lower triangular); rbs[0] < ... the block with the smaller rb starts
  // alone, then both advance together
  char* Wb_b = reinterpret_cast<char*>(W + b * (int64_t)M * B);
  {
    const int rlo = min(rbs[0], rbs[1]), rhi = max(rbs[0], rbs[1]);
    f32x16_t accW[2];        // [0]: row block rlo, [1]: row block rhi
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[u][r] = 0.f;
    if ((wave >> 1) == 0) ff_product_w<0, 3>(accW, sG, sK, cb, li, lh);
    else ff_product_w<1, 2>(accW, sG, sK, cb, li, lh);
    FF_STAMP(6);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rb = u ? rhi : rlo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          const float v = accW[u][r];
          if (col < B) *reinterpret_cast<float*>(Wb_b + 4u * (__umul24((unsigned)m, (unsigned)B) + (unsigned)col)) = v;
          s_w2 = fmaf(v, v, s_w2);
        }
      }
    }
  }
  FF_STAMP(7);
  // ---- column reductions: the two half-waves hold different rows of the same column, the waves different row blocks ----------
  s_mu += __shfl_xor(s_mu, 32, 64); s_p2 += __shfl_xor(s_p2, 32, 64); s_w2 += __shfl_xor(s_w2, 32, 64);
  if (lh == 0) {
    atomicAdd(&red[32 * cb + li], s_mu); atomicAdd(&red[64 + 32 * cb + li], s_p2); atomicAdd(&red[128 + 32 * cb + li], s_w2);
  }
  ff_barrier<MULTI>();
  if (tid < 64) {
    if (n0 + tid < B) {
      mu[b * B + n0 + tid] = red[tid];
      var[b * B + n0 + tid] = kd[b] - red[64 + tid] + red[128 + tid];
    }
  }
  FF_STAMP(8);
  }
  } while (MULTI && tile_x < tile_end);
  // ---- KL (partial sum from the top of the kernel)
  __syncthreads();
  const float tkl = block_sum<256>(kl_acc, red);
  if (tid == 0) atomicAdd(kl_u, 0.5f * tkl / (float)S);
  FF_STAMP(9);
}

// ---------------------------------------------------------------------------------------------------------------
// backward kernels
// ---------------------------------------------------------------------------------------------------------------
// First backward launch, three roles by block index.
//   blocks < npd (one per row (b, m) of P): predictive-moment backward (gp_utils.py:178-186): gP into the P block of
//       gQP, gW, the row reduction ga = sum_col P gmu and with it gQP[:, 0] = ga + g a; row 0 of each b also reduces
//       gkd.  gscale (nullable) = seed multiplying the stored unscaled softmax gradients.
//   next nkl blocks: KL backward into the remaining small columns of gQP and the diagonal of gLz:
//       gQP[:, 1..3] = 0 ; gQP[:, 4+M .. 4+2M) = g tril(G2) ; pad columns = 0 ; gLz = diag(g / Lz_ii) ;
//       gT of the S_u factors = 0 ; the G block of gQP and gT of the K_uu factors = 0 (accumulated by K-split GEMMs)
//                                                                                           (g = seed_kl / S)
//   rest: zero-fill of the r / c / gtheta accumulators of the kernel-matrix backward.
// (the G block of gQP is written by a GEMM afterwards)
__global__ __launch_bounds__(256) void t0_bwd_head_kernel(const float* __restrict__ QP, const float* __restrict__ W,
                                                          const float* __restrict__ gmu, const float* __restrict__ gvar,
                                                          const float* __restrict__ gscale, const float* __restrict__ Lz,
                                                          const float* __restrict__ seeds, float* __restrict__ gQP,
                                                          float* __restrict__ gW, float* __restrict__ gkd,
                                                          float* __restrict__ gLz, float* __restrict__ gThead,
                                                          float* __restrict__ gTtail, float* __restrict__ zero_begin, int64_t zero_count, int S, int C,
                                                          int M, int B, int NR, int LD, int npd, int nkx, int nkl, int fused,
                                                          float* __restrict__ g_u_mean, float* __restrict__ gLu_acc) {
  __shared__ float red[4];
  const float g = seeds[1] / (float)S;
  if ((int)blockIdx.x < npd) {
    const int m = (int)blockIdx.x % M;
    const int64_t b = blockIdx.x / M;
    const int64_t offp = (b * M + m) * LD + NR, offw = (b * M + m) * B;
    const float am = QP[(b * M + m) * LD];
    const float gs = gscale ? gscale[0] : 1.f;
    float acc = 0.f, accv = 0.f;
    // four 256-column chunks per batch: their 16 loads first (clamped index), then the stores -- a load behind a store
    // waits for the store as well (vmcnt counts both in order)
    for (int c0 = 0; c0 < B; c0 += 1024) {
      float gm[4], gv[4], pv[4], wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = min(c0 + 256 * q + (int)threadIdx.x, B - 1);
        gm[q] = gmu[b * B + col]; gv[q] = gvar[b * B + col];
        pv[q] = QP[offp + col]; wv[q] = W[offw + col];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = c0 + 256 * q + (int)threadIdx.x;
        if (col < B) {
          const float m_ = gs * gm[q], v_ = gs * gv[q];
          gQP[offp + col] = am * m_ - 2.f * pv[q] * v_;
          gW[offw + col] = 2.f * wv[q] * v_;
          acc = fmaf(pv[q], m_, acc);
          accv += v_;
        }
      }
    }
    const float t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) gQP[(b * M + m) * LD] = t + g * am;
    if (m == 0) {
      const float tv = block_sum<256>(accv, red);
      if (threadIdx.x == 0) gkd[b] = tv;
    }
    return;
  }
  const int id = (int)blockIdx.x - npd;
  if (id >= nkl) {
    const int nz = gridDim.x - npd - nkl;
    for (int64_t i = (int64_t)(id - nkl) * 256 + threadIdx.x; i < zero_count; i += (int64_t)nz * 256) zero_begin[i] = 0.f;
    return;
  }
  const int64_t b = id / nkx;
  const int s = b / C, c = b % C;
  const int i0 = (id % nkx) * kKlRows, i1 = min(M, i0 + kKlRows);
  const float* q = QP + b * M * LD;
  float* gq = gQP + b * M * LD;
  const int nel = (i1 - i0) * M;
  for (int e00 = 0; e00 < nel; e00 += 1024) {        // four elements per thread and batch: loads first, then stores
    float qv[4], lv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e0 = min(e00 + 256 * u + (int)threadIdx.x, nel - 1);
      const int i = i0 + e0 / M, j = e0 % M;
      qv[u] = q[(int64_t)i * LD + 4 + M + j];
      lv[u] = Lz[b * M * M + i * M + j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e0 = e00 + 256 * u + (int)threadIdx.x;
      if (e0 < nel) {
        const int i = i0 + e0 / M, j = e0 % M;
        const int e = i * M + j;
        gq[(int64_t)i * LD + 4 + M + j] = (j <= i) ? g * qv[u] : 0.f;
        gq[(int64_t)i * LD + 4 + j] = 0.f;             // G block and gT: accumulated by K-split GEMMs (float atomics)
        gThead[b * M * M + e] = 0.f;
        gLz[b * M * M + e] = (i == j) ? g / lv[u] : 0.f;
        if (s == 0) {
          gTtail[(int64_t)c * M * M + e] = 0.f;
          if (gLu_acc) gLu_acc[(int64_t)c * M * M + e] = 0.f;      // per-class sums accumulated by t0_bwd_mat.h's atomics
        }
      }
    }
  }
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    // fused: the moment role above did not run (npd = 0); ga and gkd are accumulated by t0_bwd_mid_kernel on top of these
    if (fused) gq[(int64_t)i * LD] = g * q[(int64_t)i * LD];
    if (g_u_mean && s == 0) g_u_mean[(int64_t)c * M + i] = 0.f;
    gq[(int64_t)i * LD + 1] = 0.f; gq[(int64_t)i * LD + 2] = 0.f; gq[(int64_t)i * LD + 3] = 0.f;
    for (int col = 4 + 2 * M; col < NR; ++col) gq[(int64_t)i * LD + col] = 0.f;
  }
  if (fused && i0 == 0 && threadIdx.x == 0) gkd[b] = 0.f;
}

// sum over s of the m and L_S columns of gRK:  g_u_mean[c,i] and gL of the S_u factors (input of the Cholesky backward)
__global__ void t0_unpack_kernel(const float* __restrict__ gRK, float* __restrict__ g_u_mean, float* __restrict__ gLS,
                                 int S, int C, int M, int LD, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int jj = e % (M + 1);
  const int64_t ci = e / (M + 1);        // c * M + i
  const int col = jj == 0 ? 0 : 3 + jj;
  float acc = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {            // eight samples' loads in flight together
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = gRK[((int64_t)min(s0 + u, S - 1) * C * M + ci) * LD + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (s0 + u < S) ? t[u] : 0.f;
  }
  if (jj == 0) g_u_mean[ci] = acc; else gLS[ci * M + (jj - 1)] = acc;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE: one flag per device ordinal and kernel (a process that drives
// several GPUs sets it on each)
static int ensure_dynamic_lds(const void* fn, size_t bytes, std::atomic<unsigned> (&mask)[2], const char* who) {
  int dev = 0;
  VARGP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "%s: hipGetDevice failed", who);
  if (!((mask[dev >> 5].load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    VARGP_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess,
                  "%s: cannot reserve %zu bytes of LDS", who, bytes);
    mask[dev >> 5].fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  return VARGP_OK;
}

static int check_desc(const vargp_elbo_t0_desc* d, const char* who) {
  VARGP_REQUIRE(d, "%s: null descriptor", who);
  VARGP_REQUIRE(d->S > 0 && d->C > 0 && d->M > 0 && d->D > 0 && d->B > 0 && d->F > 0, "%s: bad dims", who);
  VARGP_REQUIRE(d->log_mean && d->z && d->u_mean && d->u_tril_vec && d->x && (d->y || d->ext_lik) && d->scalars && d->info && d->ws,
                "%s: null pointer", who);
  const bool native = d->eps_f == nullptr && !d->ext_lik;    // the program draws its own noise
  VARGP_REQUIRE(!native || (d->rng_counter && d->eps_theta == nullptr && d->rng_sample_offset >= 0),
                "%s: native noise needs rng_counter, eps_theta == eps_f == NULL and a sample offset >= 0", who);
  VARGP_REQUIRE(d->map_est ? d->S == 1
                           : (d->log_logvar && d->prior_log_mean && d->prior_log_logvar && (native || d->eps_theta)),
                "%s: hyper-parameter arguments inconsistent with map_est", who);
  VARGP_REQUIRE(d->ws_bytes >= vargp_elbo_t0_workspace_bytes(d->S, d->C, d->M, d->D, d->B, d->F),
                "%s: workspace too small", who);
  return VARGP_OK;
}

// Workgroups per (s, c) of the LDS-resident tile kernels (t0_fwd_fused_kernel, t0_bwd_mid_kernel): one workgroup per CU (their
// LDS), so with SC * ntile tile units on `cus` CUs the launch takes  rounds x (set-up + tiles per workgroup x tile time).
// nparts == ntile is the single-tile form (latency-bound shapes: every unit its own CU); fewer, longer workgroups stage T and
// G once for several tiles and keep the M x M accumulators in registers across them (`setup` = that set-up in tile times).
// VARGP_T0_MULTI (tuning aid): 1 = the multi-tile kernels also where every tile has a workgroup of its own (nparts == ntile)
// most (sample, class, 64-column) tile units the LDS-resident middles take (beyond: the round-2 sequences / the block program;
// vargp_amd/vargp.py: first_task_as_block mirrors it).  VARGP_T0_UNITS: tuning aid
static int64_t t0_tile_units_max() {
  static const int64_t env = [] { const char* e = getenv("VARGP_T0_UNITS"); return e ? atoll(e) : (int64_t)kT0TileUnitsMax; }();
  return env;
}
// Role-merged launches (pivot / adjoint chains of the S C + C matrices beside the big product of the same phase) pay while the
// chains are few: a latency chain on 40 CUs hides under a product on the other 216.  With many hyper-samples the chains are a
// throughput problem of their own and the product does better as a launch of its own -- two workgroups per CU (the merged
// kernels hold it to one: the chain role's registers / LDS), 128-row tiles -- with the chains on a side stream (SideFork) so that
// the two launches share the chip.  Measured (MI355X, Split-MNIST first task, steps/s merged -> apart in line -> apart on two
// streams): S = 8 (90 chains) 2472 -> 2226 -> 2400, S = 16 (170) 1372 -> 1366 -> 1418, S = 32 (330) 697 -> 745 -> 783,
// S = 64 (650) 373 -> 413 -> 421; with the Gram matrices built by the chain workgroups (chol_gram.h, same rule) apart also wins at
// S = 8: 2468 merged, 2495 merged + Gram, 2511 apart + Gram.  Default: apart from a third of the CU count on.
// VARGP_T0_UNMERGE = chain count from which on (tuning aid).
static int t0_unmerge_chains() {
  static const int env = [] { const char* e = getenv("VARGP_T0_UNMERGE"); return e ? atoi(e) : -1; }();
  return env >= 0 ? env : vargp_cu_count() / 3 + 1;
}
// ... and then the chains go to a side stream (SideFork, common.h).  VARGP_T0_SIDE=0: in line (tuning aid)
static bool t0_side_stream() {
  static const int env = [] { const char* e = getenv("VARGP_T0_SIDE"); return e ? atoi(e) : 1; }();
  return env != 0;
}
static bool t0_force_multi() {
  static const int env = [] { const char* e = getenv("VARGP_T0_MULTI"); return e ? atoi(e) : 0; }();
  return env == 1;
}
constexpr float kBwdMidSetup = 0.35f;      // set-up of a t0_bwd_mid_multi_kernel workgroup (G / T staging + its round of atomics) in tile times
static int t0_tile_parts(int64_t SC, int ntile, int cus, float setup) {
  static const int env = [] { const char* e = getenv("VARGP_T0_PARTS"); return e ? atoi(e) : 0; }();   // tuning aid
  if (env > 0) return env < ntile ? env : ntile;
  if (SC * ntile <= cus) return ntile;
  int best = ntile;
  float best_t = 1e30f;
  for (int np = 1; np <= ntile; ++np) {
    const float t = (float)cdiv(SC * np, (int64_t)cus) * (setup + (float)cdiv(ntile, np));
    if (t < best_t - 1e-6f) { best_t = t; best = np; }
  }
  return best;
}

// which backward the shapes get (the forward needs to know: it clears the accumulators of the LDS-resident one)
struct T0BwdPaths { bool fused_bwd, mat_bwd; };
static T0BwdPaths t0_bwd_paths(const vargp_elbo_t0_desc* d, const T0Ws& o) {
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, LD = o.LD;
  // LDS-resident backward middle (t0_bwd_mid.h): same shapes as the forward's fused middle, plus B % 4 == 0 (float4 rows of W)
  static const int fused_bwd_env = [] { const char* e = getenv("VARGP_T0_FUSED_BWD"); return e ? atoi(e) : 1; }();   // tuning aid
  const int ntile = cdiv(B, 64);
  T0BwdPaths r;
  r.fused_bwd = fused_bwd_env && M <= kBmKP && M >= 4 && (M % 4) == 0 && (LD % 4) == 0 && (B % 4) == 0 &&
                (int64_t)S * C * ntile <= t0_tile_units_max();
  // ... and everything per matrix after it (small columns of gT / gRK, Cholesky adjoint, W_uu) as one LDS-resident workgroup
  // per matrix inside the launch of the P_uf product (t0_bwd_mat.h)
  static const int mat_bwd_env = [] { const char* e = getenv("VARGP_T0_MAT_BWD"); return e ? atoi(e) : 1; }();   // tuning aid
  // (the S_u matrices redo T_s^T gG_s for every hyper-sample: worth it for the few samples of the reference's configs and for
  // the 8 / 16 per GPU of BASELINE config 4 on 8 / 4 GPUs -- the product the chains hide under grows with S just as they do)
  r.mat_bwd = r.fused_bwd && mat_bwd_env && (D % 4) == 0 && S <= kTailSMax &&
              ((reinterpret_cast<uintptr_t>(d->z) | reinterpret_cast<uintptr_t>(d->x)) & 15) == 0;
  return r;
}

// One backward per forward on the LDS-resident path (the forward clears what the backward adds into).  The C ABI enforces it
// itself: host-side state per workspace, updated when a call is ISSUED (which is also when a hipGraph capture records it, so a
// captured fwd -> bwd sequence is checked once and replays as recorded).  kT0Cleared: a forward has cleared the accumulators
// and no backward has consumed them yet.
enum { kT0NoClear = 0, kT0Cleared = 1, kT0Consumed = 2, kT0SoftmaxDeferred = 16 /* flag: the forward left the likelihood to bwd */ };
static std::mutex g_t0_state_mu;
static std::unordered_map<const void*, int> g_t0_state;
static void t0_state_set(const void* ws, int v) {
  std::lock_guard<std::mutex> lock(g_t0_state_mu);
  if (g_t0_state.size() > 4096) {
    // workspaces come and go; the map only has to know the live ones.  Entries between their forward and their backward
    // (kT0Cleared) are never evicted: dropping one would make that workspace's backward fail with "no forward"
    for (auto it = g_t0_state.begin(); it != g_t0_state.end();)
      it = ((it->second & 15) == kT0Cleared && it->first != ws) ? std::next(it) : g_t0_state.erase(it);
  }
  g_t0_state[ws] = v;
}
static int t0_state_get(const void* ws) {
  std::lock_guard<std::mutex> lock(g_t0_state_mu);
  auto it = g_t0_state.find(ws);
  return it == g_t0_state.end() ? -1 : it->second;
}

}  // namespace vargp

using namespace vargp;

extern "C" size_t vargp_elbo_t0_workspace_bytes(int S, int C, int M, int D, int B, int F) {
  return carve_t0(nullptr, S, C, M, D, B, F).bytes + 256;
}

extern "C" int vargp_elbo_t0_lik_buffers(const vargp_elbo_t0_desc* d, float** mu, float** var, float** gmu, float** gvar) {
  VARGP_REQUIRE(d && d->ws, "elbo_t0_lik_buffers: null pointer");
  const T0Ws o = carve_t0(d->ws, d->S, d->C, d->M, d->D, d->B, d->F);
  if (mu) *mu = o.mu;
  if (var) *var = o.var;
  if (gmu) *gmu = o.gmu;
  if (gvar) *gvar = o.gvar;
  return VARGP_OK;
}

extern "C" int vargp_elbo_t0_fwd(const vargp_elbo_t0_desc* d, vargp_stream_t stream) {
  int rc = check_desc(d, "elbo_t0_fwd");
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, F = d->F, SC = S * C;
  const T0Ws o = carve_t0(d->ws, S, C, M, D, B, F);
  const int NR = o.NR, LD = o.LD;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  const bool fused_softmax = C <= 16;
  const bool native = d->eps_f == nullptr && !d->ext_lik;
  const float* eps_f = native ? o.eps_f : d->eps_f;

  // the merged factorisation + K_uf launch also writes L_S into RK; then the prologue writes RK's other small columns
  const bool merge_chol = D > kRbfDirectD && M > 50 && M <= 100 && (D % 4) == 0 && (LD % 4) == 0 &&
                          ((reinterpret_cast<uintptr_t>(d->z) | reinterpret_cast<uintptr_t>(d->x)) & 15) == 0;
  static const int ksp = [] { const char* e = getenv("VARGP_KUU_SPLIT"); return e ? atoi(e) : kKuuSplit; }();   // tuning aid
  static const int front_env = [] { const char* e = getenv("VARGP_T0_FRONT"); return e ? atoi(e) : 1; }();    // tuning aid
  // merged factorisation launch + K-split K_uu product: the norms ride along
  const bool split_kuu = merge_chol && D >= 256 && (int64_t)ksp * M <= LD && ksp > 1;
  // ... and then the prologue shares the launch of the split product (every workgroup evaluates the 1/sigma^2 it needs itself),
  // and the partial products are summed and exponentiated by the factorising workgroups as they load: two launches fewer
  const bool front = split_kuu && front_env && ksp <= kCholPartMax && D <= kProKuuMaxD;
  // ... with several hyper-samples the Gram matrices are a throughput problem (64 x 64 tiles of a 100-row matrix: 1.64x the work,
  // partial sums written and re-read) while a factorising workgroup has time to spare under the K_uf product of its launch: it
  // builds its Gram matrix itself (chol_gram.h: 16-row MFMA blocks, lower triangle) and the front launch keeps prologue + norms.
  // Not at S = 3: there the chain workgroups ARE the critical path of their launch.
  // Measured (steps/s without -> with): S = 3 5113 -> 4577, S = 8 2476 -> 2500, S = 16 1405 -> 1441, S = 64 420 -> 434; on from the
  // chain count at which the merged launches are taken apart (t0_unmerge_chains).  VARGP_T0_GRAM_IN_CHAIN = S from which on (tuning aid)
  static const int gic_env = [] { const char* e = getenv("VARGP_T0_GRAM_IN_CHAIN"); return e ? atoi(e) : -1; }();
  const bool gram_in_chain = front && (gic_env >= 0 ? S >= gic_env : SC + C >= t0_unmerge_chains()) && M > 64 && (M % 4) == 0 &&
                             D <= kCgMaxD && D >= 32;
  ProArgs a{};
  {
    if (merge_chol) { a.RK = o.RK; a.u_mean = d->u_mean; a.NR = NR; a.LD = LD; }
    a.mean = d->log_mean; a.logvar = d->log_logvar; a.pmean = d->prior_log_mean; a.plogvar = d->prior_log_logvar;
    a.eps_theta = d->eps_theta; a.vec = d->u_tril_vec;
    a.theta = o.theta; a.w = o.w; a.g2 = o.g2; a.kd = o.kd; a.Lu = o.Lu; a.Su = o.KS + SC * MM; a.scalars = d->scalars;
    a.bump = d->bump;
    // S_u = Lu Lu^T by the workgroup that factorises it (CholExtra::su_Lu, chol_small3.h: fp32 chains of 64 < M <= 100 on the matrix
    // core) instead of 100-long dot products in the prologue's Lu role.  VARGP_T0_SU=0: off (tuning aid)
    {
      static const int f32_env = [] { const char* e = getenv("VARGP_CHOL_F32"); return e ? atoi(e) : kCholF32Default; }();
      // VARGP_T0_SU=0: the Lu role's dot products (tuning aid).  Measured at Cfg2 (steps/s): 5323 with them; 5375 with the S_u chains
      // building their matrix and the norm role in front of the (now short) prologue roles -- front launch 22.5 -> 19.0 us, the merged
      // launch 35.2 -> 37.1 (the S_u chains become its longest); a workgroup per class inside the front launch instead: 5019 (its
      // packed-vector loads are slow, and the heavier kernel costs the front launch a workgroup slot per SIMD)
      static const int su_env = [] { const char* e = getenv("VARGP_T0_SU"); return e ? atoi(e) : 1; }();
      a.su_in_chain = (VARGP_CHOL_BLK16 && su_env && f32_env && merge_chol && M > 64 && (M % 4) == 0 && M <= 100) ? 1 : 0;
    }
    a.zero_begin = o.gmu; a.zero_count = o.r_uf - o.gmu; a.info = d->info; a.Dp = o.Dp;
    a.S = S; a.C = C; a.M = M; a.D = D; a.ninfo = SC + C; a.map_est = d->map_est;
    a.nzero_blocks = (int)std::min<int64_t>(64, cdiv(a.zero_count, 1024));
    if (native) {
      const int64_t per_sample_f = (int64_t)F * C * B;
      a.native = 1; a.seed = d->rng_seed; a.rng_counter = d->rng_counter;
      a.g0_theta = (int64_t)d->rng_sample_offset * (D + 1); a.g0_f = (int64_t)d->rng_sample_offset * per_sample_f;
      a.n_f = S * per_sample_f;
      a.eps_theta_out = o.eps_theta; a.eps_f_out = o.eps_f;
      a.nrng_blocks = (int)std::min<int64_t>(512, cdiv(a.n_f + 7, 1024));
    }
  }
  ZeroJobs bwd_zero{};
  const T0BwdPaths bwd_paths = t0_bwd_paths(d, o);
  const bool clear_bwd = bwd_paths.mat_bwd;
  // the likelihood inside the backward's tile kernel (one launch less): only where that kernel runs and its softmax fits
  // (float4 reads of the noise there: a caller's eps_f must sit on a 16-byte boundary, the workspace's own does)
  // (not on throughput-bound shapes -- the multi-tile form of that kernel, t0_bwd_mid_multi.h: the evaluation is C-fold redundant
  //  there, 11k cycles of vector work per tile that such a launch cannot hide; those shapes keep the softmax launch)
  const bool bwd_multi = bwd_paths.fused_bwd && (t0_tile_parts(SC, cdiv(B, 64), vargp_cu_count(), kBwdMidSetup) < cdiv(B, 64) || t0_force_multi());
  const bool defer_softmax = d->defer_softmax && !d->ext_lik && fused_softmax && bwd_paths.fused_bwd && !bwd_multi &&
                             F <= 4 * kBmSmF && C <= kBmSmC && reinterpret_cast<uintptr_t>(eps_f) % 16 == 0;
  t0_state_set(d->ws, (clear_bwd ? kT0Cleared : kT0NoClear) | (defer_softmax ? kT0SoftmaxDeferred : 0));
  if (clear_bwd) {
    // accumulators of the LDS-resident backward (atomics of t0_bwd_mid.h / t0_bwd_mat.h / t0_bwd_tail.h), cleared in the forward,
    // where it costs nothing (spare workgroups under the pivot chains; shapes without that launch: the prologue's zero role):
    // column 0 (ga) and the G block of gQP, gT, gkd, r_uf / c_uf / gtheta
    bwd_zero.j[0] = ZeroJob{o.gQP, (int64_t)SC * M, 4 + M, LD};
    bwd_zero.j[1] = ZeroJob{o.gTT, 1, (SC + C) * MM, 0};
    bwd_zero.j[2] = ZeroJob{o.queue, 1, 8, 0};          // work queue of the P_uf tiles (launch_bwdmat_gemm)
    bwd_zero.j[3] = ZeroJob{o.gkd, 1, SC, 0};
    bwd_zero.j[4] = ZeroJob{o.r_uf, 1, o.r_uu - o.r_uf, 0};
    if (SC + C >= t0_unmerge_chains())                  // per-class sums of the L_S gradient shares (BwdMatArgs::gL_acc; gLL's tail is free on this path)
      bwd_zero.j[5] = ZeroJob{o.gLL + SC * MM, 1, C * MM, 0};
    if (!merge_chol) {
      a.zero = bwd_zero;
      const int64_t zt = (int64_t)SC * M * (4 + M) + (SC + 2 * C) * MM + (o.r_uu - o.r_uf);
      a.nzero_blocks = (int)std::min<int64_t>(128, cdiv(a.zero_count + zt, 2048));
    }
  }
  const int npro = 1 + S + a.nzero_blocks + a.nrng_blocks + cdiv((int64_t)C * MM, 256);
  if (!front) {
    ProfScope prof("t0_prologue", st);
    hipLaunchKernelGGL(t0_prologue_kernel, dim3(npro), dim3(256), 0, st, a);
  }
  // kernel matrices: K_uu -> KS[:SC], K_uf -> the trailing block of RK
  bool merged = false;
  if (D <= kRbfDirectD) {
    rc = rbf_direct_launch(d->z, nullptr, o.w, o.g2, o.KS, M, S, C, M, M, D, o.Dp, 0, st);
    if (rc) return rc;
    rc = rbf_direct_launch(d->z, d->x, o.w, o.g2, o.RK + NR, LD, S, C, M, B, D, o.Dp, 1, st);
    if (rc) return rc;
  } else {
    const int64_t zrows = (int64_t)C * M;
    if (!split_kuu)
      hipLaunchKernelGGL(t0_norm_kernel, dim3(cdiv(zrows + B, 4), S), dim3(256), 0, st, d->z, d->x, o.w, o.na, o.nb, zrows,
                         (int64_t)B, D, o.Dp);
    GemmParams p0{}, p1{};
    p0.A = d->z; p0.B = d->z; p0.C = o.KS;
    p0.M = M; p0.N = M; p0.K = D; p0.lda = D; p0.ldb = D; p0.ldc = M;
    p0.nb1 = C; p0.nb2 = 1;
    p0.sA[1] = (int64_t)M * D; p0.sB[1] = (int64_t)M * D;
    p0.sC[0] = C * MM; p0.sC[1] = MM;
    p0.alpha = 1.f;
    p0.kscale = o.w; p0.ks_ld = o.Dp; p0.g2 = o.g2;
    p0.na = o.na; p0.sNa[0] = zrows; p0.sNa[1] = M;
    p0.nbv = o.na; p0.sNb[0] = zrows; p0.sNb[1] = M;
    p0.same_xy = 1;
    p1.A = d->z; p1.B = d->x; p1.C = o.RK + NR;
    p1.M = C * M; p1.N = B; p1.K = D; p1.lda = D; p1.ldb = D; p1.ldc = LD;
    p1.nb1 = 1; p1.nb2 = 1;
    p1.sC[0] = (int64_t)C * MLD;
    p1.alpha = 1.f;
    p1.kscale = o.w; p1.ks_ld = o.Dp; p1.g2 = o.g2;
    p1.na = o.na; p1.sNa[0] = zrows;
    p1.nbv = o.nb; p1.sNb[0] = B;
    if (merge_chol && chol_rbf_gemm_applicable(M, p1)) {
      // K_uu first, then ONE launch in which SC + C workgroups factorise (K_uu + eps I, S_u + eps I) while the rest of
      // the chip builds K_uf, which nothing needs before the factors are done
      if (split_kuu) {
        // 4 SC workgroups with D/64 slabs each would leave half the chip idle for the length of that K loop: split K,
        // partial inner products to scratch, distance/exp epilogue afterwards
        GemmParams ps = p0;
        ps.splitk = ksp; ps.sSplit = SC * MM; ps.C = o.kpart;
        if (front) {
          // (the norm role also writes x o 1/sigma_s^2: the K_uf product then runs without scale loads / multiplies in its
          // main loop -- step 209.5 -> 207.7 us at S = 3, 510 -> 494 us at S = 8)
          NormArgs nr{d->z, d->x, o.na, o.nb, zrows, (int64_t)B, 16, (int)cdiv(zrows + B, 16), o.xs};
          rc = launch_pro_kuu(a, npro, nr, ps, gram_in_chain ? 0 : SC, st);
          if (rc) return rc;
          p1.B = o.xs; p1.sB[0] = (int64_t)B * D; p1.kscale = nullptr;
        } else {
          rc = launch_gemm(ps, 0, 1, SC, true, st, "rbf_kuu_gemm");
          if (rc) return rc;
          const int64_t total = SC * MM;
          const int ncomb = cdiv(total, 256), nrow4 = cdiv(zrows + B, 4);
          hipLaunchKernelGGL(t0_combine_norm_kernel, dim3(ncomb + nrow4 * S), dim3(256), 0, st, o.kpart, ksp, SC * MM, o.g2,
                             o.KS, C, M, total, ncomb, d->z, d->x, o.w, o.na, o.nb, zrows, (int64_t)B, D, o.Dp, nrow4);
        }
      } else {
        rc = launch_gemm(p0, 0, 1, SC, true, st, "rbf_kuu_gemm");
        if (rc) return rc;
      }
      // L_S[c] -> RK[s, c, :, 4:4+M] for every s; K_uu and S_u arrive with both triangles; of the K_uu factors only
      // the diagonal of L is ever used (log-determinant, and L^T diag(.) in the backward), everything else goes through T
      CholExtra lx{o.RK + 4, SC, LD, MLD, (int64_t)C * MLD, S, 1, 1};
      if (front) {   // the matrices b < SC arrive as K-split partial Gram matrices (and leave as K_uu in KS for the backward)
        lx.part = o.kpart; lx.nsplit = ksp; lx.sSplit = SC * MM; lx.g2 = o.g2; lx.part_C = C; lx.Kout = o.KS;
        if (gram_in_chain) { lx.part = nullptr; lx.gram_z = d->z; lx.gram_w = o.w; lx.gram_D = D; lx.gram_Dp = o.Dp; }
      }
      if (a.su_in_chain) lx.su_Lu = o.Lu;
      // many hyper-samples: the chains fill the chip by themselves and hide nothing -- the product runs as a launch of its own, two
      // workgroups per CU (the merged launch holds it to one by the chain role's registers): t0_unmerge_chains()
      const bool unmerge = SC + C >= t0_unmerge_chains();
      {
        // (apart: the chains on a side stream, so that the product's workgroups fill the CUs the chains' rounds leave idle)
        SideFork fork(st, unmerge && t0_side_stream());
        rc = launch_chol_rbf_gemm(o.KS, d->jitter, o.LL, o.TT, d->info, SC + C, M, p1, unmerge ? 0 : S, fork.side(), &lx,
                                  clear_bwd ? &bwd_zero : nullptr);
        if (rc) return rc;
        if (unmerge) {
          static const int kuf_tile = [] { const char* e = getenv("VARGP_T0_KUF_TILE"); return e ? atoi(e) : 0; }();   // tuning aid
          p1.tile = kuf_tile;
          rc = launch_gemm(p1, 0, 1, S, true, st, "rbf_kuf_gemm");
          if (rc) return rc;
        }
        rc = fork.join();
        if (rc) return rc;
      }
      merged = true;
    } else {
      rc = launch_gemm_pair(p0, SC, p1, S, 0, 1, true, st, "rbf_kuu_gemm", "rbf_kuf_gemm");
      if (rc) return rc;
    }
  }
  // both factorisations (K_uu + eps I for every (s, c); S_u + eps I for every c) in one batch
  if (!merged) {
    rc = chol_inv_fwd_impl(o.KS, d->jitter, o.LL, o.TT, nullptr, d->info, SC + C, M, o.chol, o.chol_bytes, false, st);
    if (rc) return rc;
  }
  if (!merged) {
    VARGP_REQUIRE(!merge_chol, "elbo_t0_fwd: merged launch expected but not applicable");
    const int64_t total = (int64_t)SC * M * NR;
    hipLaunchKernelGGL(t0_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, d->u_mean, o.LL + SC * MM, o.Lu, o.RK, C,
                       M, NR, LD, total);
  }
  // early hand-over of the Cholesky status (include/vargp_hip.h: info_host / info_event): every factorisation of the forward is
  // behind us on the stream
  if (d->info_host && d->info_event) {
    VARGP_REQUIRE(hipMemcpyAsync(d->info_host, d->info, sizeof(int32_t) * (size_t)(SC + C), hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipEventRecord(reinterpret_cast<hipEvent_t>(d->info_event), st) == hipSuccess,
                  "elbo_t0_fwd: copy / event record of the early Cholesky status failed");
  }
  static const int fused_env = [] { const char* e = getenv("VARGP_T0_FUSED"); return e ? atoi(e) : 1; }();   // tuning aid
  const int ntile = cdiv(B, 64);
  const bool fused_mid = fused_env && M <= kFusedK && (M % 4) == 0 && (LD % 4) == 0 && (int64_t)SC * ntile <= t0_tile_units_max();
  if (fused_mid) {
    // small columns first (a = T m, G = T L_S, G2 = T Lu: one M x NR x M product per (s, c)), then the LDS-resident kernel
    GemmParams p = flat_gemm(o.TT, M, MM, o.RK, LD, MLD, o.QP, LD, MLD, M, NR, M);
    p.triA = 1;
    rc = launch_gemm(p, 0, 0, SC, false, st, "t0_qps_gemm");
    if (rc) return rc;
    static std::atomic<unsigned> attr_set_mask[4][2] = {};      // 64 device ordinals per instantiation
    const int nparts = t0_tile_parts(SC, ntile, vargp_cu_count(), 0.5f);
    const bool multi = nparts < ntile || t0_force_multi();
    const dim3 grid(8 * cdiv(SC, 8) * nparts);
    uint32_t* rngc = native ? d->rng_counter : nullptr;
#define VARGP_FF(V4, MT, SLOT)                                                                                                  \
  do {                                                                                                                          \
    rc = ensure_dynamic_lds(reinterpret_cast<const void*>(t0_fwd_fused_kernel<V4, MT>), kFusedLdsBytes, attr_set_mask[SLOT],    \
                            "elbo_t0_fwd");                                                                                     \
    if (rc) return rc;                                                                                                          \
    hipLaunchKernelGGL((t0_fwd_fused_kernel<V4, MT>), grid, dim3(256), kFusedLdsBytes, st, o.TT, o.QP, o.RK, o.W, o.kd, o.LL,   \
                       o.Lu, o.mu, o.var, d->scalars + 1, S, C, M, B, NR, LD, ntile, rngc, nparts);                             \
  } while (0)
    if (B % 4 == 0) { if (multi) VARGP_FF(true, true, 0); else VARGP_FF(true, false, 1); }
    else { if (multi) VARGP_FF(false, true, 2); else VARGP_FF(false, false, 3); }
#undef VARGP_FF
  } else {
    {  // QP = T RK
      GemmParams p = flat_gemm(o.TT, M, MM, o.RK, LD, MLD, o.QP, LD, MLD, M, NR + B, M);
      p.triA = 1;
      rc = launch_gemm(p, 0, 0, SC, false, st, "t0_qp_gemm");
      if (rc) return rc;
    }
    {  // W = G^T P
      GemmParams p = flat_gemm(o.QP + 4, LD, MLD, o.QP + NR, LD, MLD, o.W, B, (int64_t)M * B, M, B, M);
      p.triA = 2;
      rc = launch_gemm(p, 1, 0, SC, false, st, "t0_w_gemm");
      if (rc) return rc;
    }
    const int nbx = cdiv(B, 64), npd = nbx * SC, nkx = cdiv(M, kKlRows);
    hipLaunchKernelGGL(t0_pdiag_kl_fwd_kernel, dim3(npd + nkx * SC), dim3(256), 0, st, o.QP, o.W, o.kd, o.LL, o.Lu, o.mu,
                       o.var, d->scalars + 1, S, C, M, B, NR, LD, nbx, npd, nkx, native ? d->rng_counter : nullptr);
  }
  if (defer_softmax || d->ext_lik) {
    // (nothing: t0_bwd_mid_kernel evaluates the likelihood of its tile -- value into scalars[2], gradient straight into its LDS;
    // ext_lik: the caller evaluates it on the moments of ALL classes and stores the seeded gradients into gmu / gvar)
  } else if (fused_softmax) {
    const int64_t total = (int64_t)S * F * B;
    hipLaunchKernelGGL(t0_softmax_kernel<16>, dim3(cdiv(total, 256)), dim3(256), 0, st, o.mu, o.var, eps_f, d->y,
                       d->scalars + 2, o.gmu, o.gvar, S, F, C, B);
  } else {
    rc = vargp_softmax_nll_fwd(o.mu, o.var, eps_f, d->y, d->scalars + 2, S, F, C, B, stream);
    if (rc) return rc;
  }
  return check_launch("elbo_t0_fwd");
}

extern "C" int vargp_elbo_t0_bwd(const vargp_elbo_t0_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar,
                                 float* g_z, float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream) {
  int rc = check_desc(d, "elbo_t0_bwd");
  if (rc) return rc;
  VARGP_REQUIRE(seeds && g_z && g_u_mean && g_u_tril_vec && (d->defer_hyper || (g_log_mean && g_log_logvar)),
                "elbo_t0_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, F = d->F, SC = S * C;
  const T0Ws o = carve_t0(d->ws, S, C, M, D, B, F);
  const int NR = o.NR, LD = o.LD;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD, MB = (int64_t)M * B;
  const bool fused_softmax = C <= 16 && !d->ext_lik;      // ext_lik: gmu / gvar arrive seeded, as from the generic kernel
  const bool native = d->eps_f == nullptr && !d->ext_lik;
  const float* eps_f = native ? o.eps_f : d->eps_f;
  const float* eps_theta = native ? o.eps_theta : d->eps_theta;

  if (!fused_softmax && !d->ext_lik) {   // C > 16: gradient of the likelihood from the generic kernel (already scaled by its seed)
    rc = vargp_softmax_nll_bwd(o.mu, o.var, eps_f, d->y, seeds + 2, o.gmu, o.gvar, S, F, C, B, stream);
    if (rc) return rc;
  }
  const int ntile = cdiv(B, 64);
  const T0BwdPaths paths = t0_bwd_paths(d, o);
  const bool fused_bwd = paths.fused_bwd, mat_bwd = paths.mat_bwd;
  const int state_all = t0_state_get(d->ws);
  const bool softmax_deferred = state_all >= 0 && (state_all & kT0SoftmaxDeferred) != 0;
  VARGP_REQUIRE(!softmax_deferred || fused_bwd, "elbo_t0_bwd: the forward deferred the likelihood to a backward path this call does not take");
  if (mat_bwd) {
    // the accumulators must have been cleared by a forward on THIS workspace that took the same decision (the decision
    // depends on the alignment of d->z / d->x and on tuning variables) and must not have been consumed by a backward yet
    const int state = state_all < 0 ? state_all : (state_all & ~kT0SoftmaxDeferred);
    VARGP_REQUIRE(state == kT0Cleared,
                  "elbo_t0_bwd: %s -- this path allows ONE vargp_elbo_t0_bwd per vargp_elbo_t0_fwd (the forward clears the "
                  "accumulators the backward adds into); run the forward again",
                  state == kT0Consumed ? "second backward on one forward"
                                       : "no forward on this workspace with the same z / x alignment");
    t0_state_set(d->ws, kT0Consumed);
  } else if (softmax_deferred) {
    t0_state_set(d->ws, kT0NoClear);           // (the likelihood is added into scalars[2] once)
  }
  // mat_bwd: no head launch -- the forward's zero role has cleared the accumulators, the seed-dependent KL columns (g a, g G2) are
  // formed by the chain kernel from QP, g_u_mean is cleared by the tile kernel.  ONE backward per forward on this path.
  if (!mat_bwd)
  {
    const int npd = fused_bwd ? 0 : M * SC, nkx = cdiv(M, kKlRows), nkl = nkx * SC;
    const int64_t zc = o.r_uu - o.r_uf;
    const int nz = (int)std::min<int64_t>(64, cdiv(zc, 1024));
    hipLaunchKernelGGL(t0_bwd_head_kernel, dim3(npd + nkl + nz), dim3(256), 0, st, o.QP, o.W, o.gmu, o.gvar,
                       fused_softmax ? seeds + 2 : nullptr, o.LL, seeds, o.gQP, o.gW, o.gkd, o.gLL, o.gTT, o.gTT + SC * MM, o.r_uf,
                       zc, S, C, M, B, NR, LD, npd, nkx, nkl, fused_bwd ? 1 : 0, mat_bwd ? g_u_mean : nullptr,
                       mat_bwd ? o.gLL + SC * MM : nullptr);
  }
  if (fused_bwd) {
    static std::atomic<unsigned> attr_set_mask[2] = {}, attr_set_mask_m[2] = {};
    // throughput-bound shapes (more tile units than CUs): the multi-tile form (t0_bwd_mid_multi.h)
    const int nparts = t0_tile_parts(SC, ntile, vargp_cu_count(), kBwdMidSetup);
    const BmSoftmax smx = softmax_deferred ? BmSoftmax{o.mu, o.var, eps_f, d->y, d->scalars + 2, F} : BmSoftmax{};
    if (nparts < ntile || t0_force_multi()) {
      VARGP_REQUIRE(!softmax_deferred, "elbo_t0_bwd: the multi-tile backward does not evaluate a deferred likelihood");
      rc = ensure_dynamic_lds(reinterpret_cast<const void*>(t0_bwd_mid_multi_kernel), kBwdMidMultiLdsBytes, attr_set_mask_m, "elbo_t0_bwd");
      if (rc) return rc;
      ProfScope prof("t0_bwd_mid", st);
      hipLaunchKernelGGL(t0_bwd_mid_multi_kernel, dim3(8 * cdiv(SC, 8) * nparts), dim3(256), kBwdMidMultiLdsBytes, st, o.TT, o.QP, o.W, o.RK,
                         o.gmu, o.gvar, fused_softmax ? seeds + 2 : nullptr, o.gQP, o.gTT, o.gRK, o.gkd, o.r_uf, o.c_uf, o.gtheta, S, C, M,
                         B, D, NR, LD, ntile, nparts, mat_bwd ? g_u_mean : nullptr, C * M);
    } else {
      rc = ensure_dynamic_lds(reinterpret_cast<const void*>(t0_bwd_mid_kernel), kBwdMidLdsBytes, attr_set_mask, "elbo_t0_bwd");
      if (rc) return rc;
      ProfScope prof("t0_bwd_mid", st);
      hipLaunchKernelGGL(t0_bwd_mid_kernel, dim3(8 * cdiv(SC, 8) * ntile), dim3(256), kBwdMidLdsBytes, st, o.TT, o.QP, o.W, o.RK, o.gmu, o.gvar,
                         fused_softmax ? seeds + 2 : nullptr, o.gQP, o.gTT, o.gRK, o.gkd, o.r_uf, o.c_uf, o.gtheta, S, C, M, B, D,
                         NR, LD, ntile, mat_bwd ? g_u_mean : nullptr, C * M, smx);
    }
    if (!mat_bwd) {
      // what the tiles cannot see: the small columns [a | . | G | G2 | .] of QP = T RK (K = NR):
      //   gT += tril(gQP[:, :NR] RK[:, :NR]^T)   on top of the tiles' atomics;   gRK[:, :NR] = T^T gQP[:, :NR]
      GemmParams p = flat_gemm(o.gQP, LD, MLD, o.RK, LD, MLD, o.gTT, M, MM, M, M, NR);
      p.triC = 1; p.D = o.gTT; p.ldd = M; p.beta = 1.f;
      GemmParams q = flat_gemm(o.TT, M, MM, o.gQP, LD, MLD, o.gRK, LD, MLD, M, NR, M);
      q.triA = 2;
      rc = launch_gemm_pair2(p, 0, 1, SC, q, 1, 0, SC, st, "t0_gt_grk_small");
      if (rc) return rc;
      rc = chol_inv_bwd_first(o.TT, o.gTT, SC + C, M, o.chol, o.chol_bytes, nullptr, 0, 0, 0, st);
      if (rc) return rc;
    }
  } else {
  {  // W = G^T P:  gG = P gW^T (G block of gQP),  gP += G gW   -- independent of each other: one launch
    GemmParams p = flat_gemm(o.QP + NR, LD, MLD, o.gW, B, MB, o.gQP + 4, LD, MLD, M, M, B);
    p.splitk = ksplit(B);     // 4 SC tiles with a B-long K loop: split K so that the chip is busy (atomic accumulation)
    GemmParams q = flat_gemm(o.QP + 4, LD, MLD, o.gW, B, MB, o.gQP + NR, LD, MLD, M, B, M);
    q.triA = 1; q.D = o.gQP + NR; q.beta = 1.f;
    rc = launch_gemm_pair2(p, 0, 1, SC, q, 0, 0, SC, st, "t0_gg_gp_gemm");
    if (rc) return rc;
  }
  {  // QP = T RK:  gT = tril(gQP RK^T)
    GemmParams p = flat_gemm(o.gQP, LD, MLD, o.RK, LD, MLD, o.gTT, M, MM, M, M, NR + B);
    p.triC = 1;
    p.splitk = ksplit(NR + B);
    p.sSplit = 0;
    rc = launch_gemm(p, 0, 1, SC, false, st, "t0_gt_gemm");
    if (rc) return rc;
  }
  // gRK = T^T gQP shares a launch with the first product of the Cholesky backward (w1 = gT T^T), which only needs gT
  GemmParams grk = flat_gemm(o.TT, M, MM, o.gQP, LD, MLD, o.gRK, LD, MLD, M, NR + B, M);
  grk.triA = 2;
  rc = chol_inv_bwd_first(o.TT, o.gTT, SC + C, M, o.chol, o.chol_bytes, &grk, 1, 0, SC, st);
  if (rc) return rc;
  }
  const int64_t zrows = (int64_t)C * M;
  bool fused_tail = false;
  GemmParams p0{}, p1{};       // the W.Y products of the kernel-matrix backward: P_uu = W_uu z per (s, c), P_uf = W_uf x per s
  p0.A = o.Wuu; p0.B = d->z; p0.C = o.Puu;
  p0.M = M; p0.N = D; p0.K = M; p0.lda = M; p0.ldb = D; p0.ldc = D;
  p0.nb1 = C; p0.nb2 = 1;
  p0.sA[0] = C * MM; p0.sA[1] = MM;
  p0.sB[1] = (int64_t)M * D;
  p0.sC[0] = zrows * D; p0.sC[1] = (int64_t)M * D;
  p0.alpha = 1.f;
  p1.A = o.gRK + NR; p1.B = d->x; p1.C = o.Puf;
  p1.M = C * M; p1.N = D; p1.K = B; p1.lda = LD; p1.ldb = D; p1.ldc = D;
  p1.nb1 = 1; p1.nb2 = 1;
  p1.sA[0] = C * MLD;
  p1.sC[0] = zrows * D;
  p1.alpha = 1.f;
  if (mat_bwd) {
    BwdMatArgs ma{};
    ma.QP = o.QP; ma.TT = o.TT; ma.LL = o.LL; ma.gQP = o.gQP; ma.RK = o.RK; ma.KS = o.KS; ma.seeds = seeds; ma.gTT = o.gTT;
    ma.gKS = o.gKS; ma.Wuu = o.Wuu; ma.r_uu = o.r_uu; ma.gtheta = o.gtheta;
    ma.g_u_mean = g_u_mean; ma.gLu_part = o.gLL;                 // (gLL is free on this path: no head launch, no Cholesky-adjoint op)
    ma.S = S; ma.C = C; ma.M = M; ma.D = D; ma.NR = NR; ma.LD = LD;
    // all S C + C matrices next to P_uf = W_uf x (which only needs the tile kernel's W_uf) ...
    // many hyper-samples (as in the forward: vargp_elbo_t0_fwd): the chains fill the chip by themselves, the product runs as a
    // launch of its own (two workgroups per CU), and the S_u roles -- which would walk all S samples, one workgroup per class,
    // with nothing left to hide under -- read per-class sums the K_uu roles accumulate (cleared by the forward's zero role)
    const bool unmerge = SC + C >= t0_unmerge_chains();
    if (unmerge) {
      SideFork fork(st, t0_side_stream());               // (the chains on a side stream, as in the forward)
      ma.gL_acc = o.gLL + SC * MM;
      rc = launch_bwdmat_gemm(ma, 0, SC, p1, 0, fork.side(), "t0_bwdmat_kuu", nullptr);
      if (rc) return rc;
      rc = launch_bwdmat_gemm(ma, SC, C, p1, 0, fork.side(), "t0_bwdmat_su", nullptr);
      if (rc) return rc;
      static const int puf_tile = [] { const char* e = getenv("VARGP_T0_PUF_TILE"); return e ? atoi(e) : 0; }();   // tuning aid
      p1.tile = puf_tile;
      rc = launch_gemm(p1, 0, 0, S, false, st, "rbf_kuf_bwd_product");
      if (rc) return rc;
      rc = fork.join();
    } else {
      rc = launch_bwdmat_gemm(ma, 0, SC + C, p1, S, st, "rbf_kuf_bwd_gemm", reinterpret_cast<int*>(o.queue));
    }
    if (rc) return rc;
    // ... then P_uu = W_uu z; with few samples inside the launch that consumes it (t0_bwd_tail.h)
    static const int tail_env = [] { const char* e = getenv("VARGP_T0_TAIL"); return e ? atoi(e) : 1; }();   // tuning aid
    fused_tail = tail_env && S <= kTailSMax;
    if (!fused_tail) {
      static const int puu_tile = [] { const char* e = getenv("VARGP_T0_PUU_TILE"); return e ? atoi(e) : 0; }();   // tuning aid
      p0.tile = puu_tile;
      rc = launch_gemm(p0, 0, 0, SC, false, st, "rbf_kuu_bwd_gemm");
      if (rc) return rc;
    }
  } else {
  {
    const int64_t total = (int64_t)C * M * (M + 1);
    hipLaunchKernelGGL(t0_unpack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, o.gRK, g_u_mean, o.gLL + SC * MM, S, C,
                       M, LD, total);
  }
  // gLL is lower-triangular by construction (diagonal for the K_uu factors, the L_S block of gRK for the S_u ones)
  rc = chol_inv_bwd_impl(o.LL, o.TT, o.gLL, o.gTT, o.gKS, SC + C, M, o.chol, o.chol_bytes, true, st, true);
  if (rc) return rc;
  // kernel matrices -> z, theta
  {
    const int gx = cdiv(B, 256), gy = cdiv(zrows, kWRows), nuf = fused_bwd ? 0 : gx * gy * S;   // fused: W_uf is done
    const int nuu = SC * cdiv(M, kUuRows);
    const int ngv = cdiv((int64_t)C * MM, 256);      // + the gradient of the packed Cholesky vector of q(u)
    hipLaunchKernelGGL(t0_w_kernel, dim3(nuf + nuu + ngv), dim3(256), 0, st, o.RK, o.gRK, o.KS, o.gKS, o.Wuu, o.r_uu, o.r_uf,
                       o.c_uf, o.gtheta, S, C, M, B, D, NR, LD, gx, gy, nuf, nuu, d->u_tril_vec, o.Lu, seeds, g_u_tril_vec, 1);
  }
  rc = launch_gemm_pair(p0, SC, p1, S, 0, 0, false, st, "rbf_kuu_bwd_gemm", "rbf_kuf_bwd_gemm");
  if (rc) return rc;
  }
  {
    const int nzy = cdiv(zrows, kFinRows), nxy = cdiv(B, kFinRows), gx = cdiv(D, 64);
    GvecArgs gv{};
    int ngy = 0;
    if (mat_bwd) {               // the gradient of the packed Cholesky vector of q(u) from the per-class sums: extra grid rows
      gv.vec = d->u_tril_vec; gv.Lu = o.Lu; gv.gSu = o.gKS + SC * MM; gv.gLu_part = o.gLL; gv.seeds = seeds;
      gv.gvec = g_u_tril_vec; gv.S = S; gv.C = C; gv.M = M; gv.y0 = nzy + nxy;
      ngy = cdiv(cdiv((int64_t)C * MM, 256), gx);
    }
    if (fused_tail) {
      ProfScope prof("t0_puu_final", st);
      TailArgs ta{};
      ta.z = d->z; ta.x = d->x; ta.Wuu = o.Wuu; ta.Puf = o.Puf; ta.r_uu = o.r_uu; ta.r_uf = o.r_uf; ta.c_uf = o.c_uf; ta.w = o.w;
      ta.gz = g_z; ta.gtheta = o.gtheta; ta.S = S; ta.C = C; ta.M = M; ta.D = D; ta.B = B; ta.Dp = o.Dp;
      const int rem = M % 32;
      ta.nrb = (rem > 0 && rem <= 8) ? M / 32 : cdiv(M, 32);       // a short remainder goes to the vector units
      const int ngr = cdiv(M, 32);
      ta.ncb = cdiv(D, 32); ta.nz = cdiv(C * ta.nrb * ta.ncb + C * (ngr * (ngr + 1) / 2), 4);      // z + packed-vector wave-blocks
      ta.nx = gx * cdiv(B, kTailXRows); ta.gx = gx;
      ta.nrem = (rem > 0 && rem <= 8) ? C * gx * cdiv(rem, 4) : 0;
      const dim3 grid(ta.nz + ta.nrem + ta.nx);
      // more than four samples: the per-sample operands of the z role staged through LDS (t0_puu_final_lds_kernel)
      static const int tail_lds_env = [] { const char* e = getenv("VARGP_T0_TAIL_LDS"); return e ? atoi(e) : kTailLdsDefault; }();   // tuning aid
      if ((S > 4 && tail_lds_env) || tail_lds_env == 2) {      // (2: also with few samples -- tuning aid)
        static std::atomic<unsigned> attr_set_mask_t[2] = {};
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(t0_puu_final_lds_kernel), kTailLdsBytes, attr_set_mask_t, "elbo_t0_bwd");
        if (rc) return rc;
        int nzg = C * ta.nrb * cdiv(ta.ncb, 4), ngvw = cdiv(C * (ngr * (ngr + 1) / 2), 4);
        // VARGP_EXP_TAIL (timing only, wrong results): 1 = the z groups alone, 2 = the other roles alone
        static const int exp_tail = [] { const char* e = getenv("VARGP_EXP_TAIL"); return e ? atoi(e) : 0; }();
        if (exp_tail == 1) { ngvw = 0; ta.nrem = 0; ta.nx = 0; }
        if (exp_tail == 2) nzg = 0;
        hipLaunchKernelGGL(t0_puu_final_lds_kernel, dim3(nzg + ngvw + ta.nrem + ta.nx), dim3(512), kTailLdsBytes, st, ta, gv, nzg, ngvw);
      } else
      switch (S) {
        case 1: hipLaunchKernelGGL(t0_puu_final_kernel<1>, grid, dim3(256), 0, st, ta, gv); break;
        case 2: hipLaunchKernelGGL(t0_puu_final_kernel<2>, grid, dim3(256), 0, st, ta, gv); break;
        case 3: hipLaunchKernelGGL(t0_puu_final_kernel<3>, grid, dim3(256), 0, st, ta, gv); break;
        case 4: hipLaunchKernelGGL(t0_puu_final_kernel<4>, grid, dim3(256), 0, st, ta, gv); break;
        default: hipLaunchKernelGGL(t0_puu_final_kernel<0>, grid, dim3(256), 0, st, ta, gv); break;      // 5 .. kTailSMax: a loop
      }
    } else {
      hipLaunchKernelGGL(t0_final_kernel, dim3(gx, nzy + nxy + ngy), dim3(256), 0, st, d->z, d->x, o.r_uu, o.r_uf, o.c_uf,
                         o.Puu, o.Puf, o.w, g_z, o.gtheta, zrows, (int64_t)B, D, o.Dp, S, nzy, gv);
    }
  }
  if (!d->defer_hyper)
    hipLaunchKernelGGL(t0_hyper_bwd_kernel, dim3(cdiv(D + 1, 256)), dim3(256), 0, st, d->log_mean, d->log_logvar,
                       d->prior_log_mean, d->prior_log_logvar, eps_theta, o.gtheta, o.g2, o.gkd, seeds, g_log_mean,
                       g_log_logvar, S, C, D + 1, d->map_est);
  return check_launch("elbo_t0_bwd");
}

extern "C" int vargp_elbo_t0_hyper_desc(const vargp_elbo_t0_desc* d, const float* seeds, vargp_hyper_grad_desc* out) {
  int rc = check_desc(d, "elbo_t0_hyper_desc");
  if (rc) return rc;
  VARGP_REQUIRE(seeds && out, "elbo_t0_hyper_desc: null pointer");
  const T0Ws o = carve_t0(d->ws, d->S, d->C, d->M, d->D, d->B, d->F);
  out->log_mean = d->log_mean; out->log_logvar = d->log_logvar;
  out->prior_log_mean = d->prior_log_mean; out->prior_log_logvar = d->prior_log_logvar;
  out->eps_theta = (d->eps_f == nullptr && !d->ext_lik) ? o.eps_theta : d->eps_theta;
  out->gtheta = o.gtheta; out->g2 = o.g2; out->gkd = o.gkd; out->seeds = seeds;
  out->S = d->S; out->C = d->C; out->D1 = d->D + 1; out->map_est = d->map_est;
  return VARGP_OK;
}
