// Device code and small host helpers shared by the two native ELBO programs (elbo_t0.hip: first task; elbo_tn.hip:
// tasks t > 0 and any M): softplus / sigmoid, the counter-based normal generator, the fused softmax likelihood kernel,
// the hyper-parameter backward, the K-split rule and the flat batched-GEMM descriptor.  `static`: one copy per
// translation unit.
#pragma once
#include "common.h"
#include "elbo_rng.h"
#include <stdlib.h>

namespace vargp {

// Monte-Carlo softmax likelihood (likelihoods.py:13-45) and its gradient in one pass, C <= CMAX: one thread per
// (s, f, b) keeps the class vector in registers, adds -log softmax_y / (S F) to nll and its share of
// d nll / d mu, d nll / d var to the (pre-zeroed) accumulators.  The backward only scales them by the incoming seed.
template <int CMAX>
static __global__ __launch_bounds__(256) void t0_softmax_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                         const float* __restrict__ eps, const int64_t* __restrict__ y,
                                                         float* __restrict__ nll, float* __restrict__ gmu,
                                                         float* __restrict__ gvar, int S, int F, int C, int B) {
  __shared__ float red[4];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (e < (int64_t)S * F * B) {
    const int b = e % B, f = (e / B) % F, s = e / ((int64_t)B * F);
    const int yb = (int)y[b];
    float sd[CMAX], ev[CMAX], v[CMAX], mx = -INFINITY, fy = 0.f;
    // all 3 CMAX loads first, unconditional on a clamped class index (a load behind `c < C` gets a branch and a
    // `s_waitcnt vmcnt(0)` of its own: CMAX memory round trips in a row), values of the padding classes replaced after
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int cc = c < C ? c : C - 1;
      const int64_t i = ((int64_t)s * C + cc) * B + b;
      sd[c] = var[i];
      ev[c] = eps[(((int64_t)s * F + f) * C + cc) * B + b];
      v[c] = mu[i];
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      sd[c] = c < C ? sqrtf(sd[c]) : 1.f;
      ev[c] = c < C ? ev[c] : 0.f;
      v[c] = c < C ? v[c] + sd[c] * ev[c] : -INFINITY;
      mx = fmaxf(mx, v[c]);
      if (c == yb) fy = v[c];
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { v[c] = c < C ? expf(v[c] - mx) : 0.f; se += v[c]; }
    const float sc1 = 1.f / (float)(S * F), sc = sc1 / se;
    contrib = -(fy - (mx + logf(se))) * sc1;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        const int64_t i = ((int64_t)s * C + c) * B + b;
        const float p = v[c] * sc - (c == yb ? sc1 : 0.f);
        atomicAdd(&gmu[i], p);
        atomicAdd(&gvar[i], p * ev[c] * 0.5f / sd[c]);
      }
    }
  }
  const float t = block_sum<256>(contrib, red);
  if (threadIdx.x == 0) atomicAdd(nll, t);
}

// gtheta (+ the gamma^2 of the predictive variance, kernels.py:58-60) -> variational hyper-parameters, plus the
// gradient of kl_hypers scaled by its seed (kernels.py:62-77)
static __global__ void t0_hyper_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                    const float* __restrict__ pmean, const float* __restrict__ plogvar,
                                    const float* __restrict__ eps, const float* __restrict__ gtheta,
                                    const float* __restrict__ g2, const float* __restrict__ gkd,
                                    const float* __restrict__ seeds, float* __restrict__ gmean,
                                    float* __restrict__ glogvar, int S, int C, int D1, int map_est) {
  __shared__ float red[4];
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = d < D1;
  const int dc = live ? d : D1 - 1;
  const float hs = map_est ? 0.f : 0.5f * expf(0.5f * logvar[dc]);
  float gm = 0.f, gv = 0.f;
  // eight samples per batch, loads first (S = 64 made the one-load-at-a-time loop 55 us long)
  for (int s0 = 0; s0 < S; s0 += 8) {
    float gt[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = min(s0 + u, S - 1);
      gt[u] = gtheta[s * D1 + dc];
      ev[u] = map_est ? 0.f : eps[s * D1 + dc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < S) {
        gm += gt[u];
        if (!map_est) gv = fmaf(gt[u] * hs, ev[u], gv);
      }
    }
  }
  // gamma^2 of the predictive variance (theta_D):  g_s += 2 g2[s] sum_c gkd[s, c].  The whole last block shares out the
  // S C terms (one thread walking them was 40 of this kernel's 45 us at S = 64) and reduces the two sums they feed.
  if (blockIdx.x == gridDim.x - 1) {          // (uniform) the block that holds d = D1 - 1
    const int D = D1 - 1;
    const float hsD = map_est ? 0.f : 0.5f * expf(0.5f * logvar[D]);
    float am = 0.f, av = 0.f;
    for (int e = threadIdx.x; e < S * C; e += blockDim.x) {
      const int s = e / C;
      const float t = 2.f * g2[s] * gkd[e];
      am += t;
      if (!map_est) av = fmaf(t * hsD, eps[s * D1 + D], av);
    }
    const float tm = block_sum<256>(am, red);
    __syncthreads();
    const float tv = block_sum<256>(av, red);
    if (d == D) { gm += tm; gv += tv; }
  }
  if (!live) return;
  if (!map_est) {
    const float g = seeds[0];
    gm += g * (mean[d] - pmean[d]) * expf(-plogvar[d]);
    gv += g * 0.5f * (expf(logvar[d] - plogvar[d]) - 1.f);
  }
  gmean[d] = gm;
  glogvar[d] = gv;
}

// K-splits for the products with few output tiles and a long K: about 4 slabs of 64 per workgroup (measured best)
static int ksplit(int K) {
  static const int per = [] { const char* e = getenv("VARGP_KSPLIT_PER"); return e ? atoi(e) : 256; }();   // tuning aid
  const int s = K / per;
  return s < 1 ? 1 : (s > 8 ? 8 : s);
}

static GemmParams flat_gemm(const float* A, int lda, int64_t sA, const float* B, int ldb, int64_t sB, float* C, int ldc,
                            int64_t sC, int M, int N, int K) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.D = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldc;
  p.nb1 = 1; p.nb2 = 1;
  p.sA[0] = sA; p.sB[0] = sB; p.sC[0] = sC; p.sD[0] = sC;
  p.alpha = 1.f; p.beta = 0.f;
  return p;
}


#ifndef VARGP_W_ROWS
#define VARGP_W_ROWS 8
#endif
#ifndef VARGP_UU_ROWS
#define VARGP_UU_ROWS 16
#endif
#ifndef VARGP_FIN_ROWS
#define VARGP_FIN_ROWS 32
#endif
constexpr int kWRows = VARGP_W_ROWS;      // rows per workgroup of the W = gK o K pass (K_uf role)
constexpr int kUuRows = VARGP_UU_ROWS;    // rows per workgroup of the same pass, K_uu role
constexpr int kFinRows = VARGP_FIN_ROWS;   // rows per workgroup of the RBF finalisation

// gradient of the packed Cholesky vector of q(u):  gLu = sum_s gRK[.., Lu block] - seed_kl diag(1/Lu_ii) + 2 gS_u Lu,
// through vec2tril (softplus on the diagonal).  One thread per (c, i, k <= i).  (A role of t0_w_kernel.)
// gRK: [S][C][M][LD] with the Lu block at column 4 + M (summed over s here), or -- S = 1, LD = M, off = 0 -- the per-class sum
// array's layout (t0_bwd_mat.h: [S][C][M][M], lower triangles).
__device__ __forceinline__ void t0_gvec_role(int blk, const float* __restrict__ vec, const float* __restrict__ Lu,
                                             const float* __restrict__ gSu, const float* __restrict__ gRK,
                                             const float* __restrict__ seeds, float* __restrict__ gvec, int S, int C,
                                             int M, int LD, int off = -1) {
  if (off < 0) off = 4 + M;
  const int64_t e = (int64_t)blk * 256 + threadIdx.x;
  if (e >= (int64_t)C * M * M) return;
  const int k = e % M, i = (e / M) % M;
  const int64_t c = e / ((int64_t)M * M);
  if (k > i) return;
  const float* gs = gSu + (c * M + i) * M;
  const float* lu = Lu + c * M * M + k;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  // Lu[j][k] = 0 (stored) for j < k: start at the aligned group containing k.  16 products per batch, their loads in
  // flight together (clamped index, masked value): M / 16 memory round trips in a row instead of M / 4.
  for (int j0 = k & ~3; j0 < M; j0 += 16) {
    float gv[16], lv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int j = min(j0 + q, M - 1);
      gv[q] = gs[j]; lv[q] = lu[(int64_t)j * M];
    }
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
      acc0 = fmaf(j0 + q < M ? gv[q] : 0.f, lv[q], acc0);
      acc1 = fmaf(j0 + q + 1 < M ? gv[q + 1] : 0.f, lv[q + 1], acc1);
      acc2 = fmaf(j0 + q + 2 < M ? gv[q + 2] : 0.f, lv[q + 2], acc2);
      acc3 = fmaf(j0 + q + 3 < M ? gv[q + 3] : 0.f, lv[q + 3], acc3);
    }
  }
  float g = 2.f * ((acc0 + acc1) + (acc2 + acc3));
  for (int s = 0; s < S; ++s) g += gRK[(((int64_t)s * C + c) * M + i) * LD + off + k];
  const int64_t idx = c * ((int64_t)M * (M + 1) / 2) + (int64_t)i * (i + 1) / 2 + k;
  if (i == k) {
    g -= seeds[1] / lu[(int64_t)i * M];
    const float x = vec[idx];
    g *= (x > 20.f) ? 1.f : sigmoid_t0(x);
  }
  gvec[idx] = g;
}

// arguments of the packed-Cholesky-vector gradient when it rides in t0_final_kernel's launch (gvec == NULL: off)
struct GvecArgs {
  const float *vec, *Lu, *gSu, *gLu_part, *seeds;      // gLu_part: [S][C][M][M] (t0_bwd_mat.h)
  float* gvec;
  int S, C, M, y0;
};

// W = gK o K for both kernel matrices (see rbf.hip for the algebra).
//   blocks < nuf : K_uf, in place on the K_uf block of gRK (row stride LD); row sums r_uf, column sums c_uf (atomics),
//                  2 sum W into gtheta[s, D]
//   next nuu     : K_uu, one wave per row: Wuu = W + W^T, r_uu = its row sums, sum Wuu (= 2 sum W) into gtheta[s, D]
//   rest         : the packed-Cholesky-vector gradient (t0_gvec_role), which only shares the launch
static __global__ __launch_bounds__(256) void t0_w_kernel(const float* __restrict__ RK, float* __restrict__ gRK,
                                                   const float* __restrict__ Kuu, const float* __restrict__ gKuu,
                                                   float* __restrict__ Wuu, float* __restrict__ r_uu,
                                                   float* __restrict__ r_uf, float* __restrict__ c_uf,
                                                   float* __restrict__ gtheta, int S, int C, int M, int B, int D, int NR,
                                                   int LD, int gx, int gy, int nuf, int nuu, const float* __restrict__ vec,
                                                   const float* __restrict__ Lu, const float* __restrict__ seeds,
                                                   float* __restrict__ gvec, int sym_guu) {
  __shared__ float red[4];
  // block order: the packed-vector gradient first (long per-thread loops: started early they finish under the rest),
  // then K_uu, then K_uf
  const int ngv = (int)gridDim.x - nuf - nuu;
  if ((int)blockIdx.x < ngv) {
    t0_gvec_role((int)blockIdx.x, vec, Lu, gKuu + (int64_t)S * C * M * M, gRK, seeds, gvec, S, C, M, LD);
    return;
  }
  const int bid = (int)blockIdx.x - ngv;
  const int lane = threadIdx.x & 63;
  if (bid >= nuu) {
    const int id = bid - nuu;
    const int col = (id % gx) * 256 + threadIdx.x;
    const int CM = C * M;
    const int row0 = ((id / gx) % gy) * kWRows;
    const int64_t s = id / (gx * gy);
    const bool cok = col < B;
    const int cc = cok ? col : B - 1;
    const int rend = min(kWRows, CM - row0);
    if ((B & 3) == 0 && (kWRows & 3) == 0) {
      // 16 bytes per lane: the block's 256 columns are one float4 per lane, wave w takes the rows w, w + 4, ... of the block.
      // Per thread kWRows / 4 pairs of loads and one store each instead of kWRows pairs of scalar ones, and one wave
      // reduction per ROW (the scalar version: one per row and 64 columns) -- the pass ran at 1.9 TB/s (Permuted-MNIST t = 1:
      // 246 MB in 124 us).
      __shared__ float cred[4][256];
      const int wv = threadIdx.x >> 6;
      const int c4 = (id % gx) * 256 + 4 * lane;               // first of the lane's four columns
      const bool c4ok = c4 < B;                                // (B % 4 == 0: all four or none)
      const int c4c = c4ok ? c4 : B - 4;
      constexpr int RW = kWRows / 4;
      float4 kv4[RW], gv4[RW];
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        const int64_t off = (s * CM + row0 + min(wv + 4 * q, rend - 1)) * LD + NR + c4c;
        kv4[q] = *reinterpret_cast<const float4*>(RK + off);
        gv4[q] = *reinterpret_cast<const float4*>(gRK + off);
      }
      float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        const int rr = wv + 4 * q;
        const bool ok = c4ok && rr < rend;
        float4 v = make_float4(kv4[q].x * gv4[q].x, kv4[q].y * gv4[q].y, kv4[q].z * gv4[q].z, kv4[q].w * gv4[q].w);
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) *reinterpret_cast<float4*>(gRK + (s * CM + row0 + rr) * LD + NR + c4) = v;
        cs4.x += v.x; cs4.y += v.y; cs4.z += v.z; cs4.w += v.w;
        const float rs = wave_sum((v.x + v.y) + (v.z + v.w));
        if (lane == 0 && rs != 0.f) atomicAdd(&r_uf[s * CM + row0 + rr], rs);
      }
      *reinterpret_cast<float4*>(&cred[wv][4 * lane]) = cs4;
      __syncthreads();
      const float csum = (cred[0][threadIdx.x] + cred[1][threadIdx.x]) + (cred[2][threadIdx.x] + cred[3][threadIdx.x]);
      if (col < B) atomicAdd(&c_uf[s * B + col], csum);
      const float tot = block_sum<256>(col < B ? csum : 0.f, red);
      if (threadIdx.x == 0) atomicAdd(&gtheta[s * (D + 1) + D], 2.f * tot);
      return;
    }
    float csum = 0.f;
    // every load of the block first (clamped indices, masked values), then stores and reductions: a load issued after a
    // store, or inside a branch, waits for everything before it (vmcnt counts loads and stores in order)
    float kv[kWRows], gv[kWRows];
#pragma unroll
    for (int rr = 0; rr < kWRows; ++rr) {
      const int64_t off = (s * CM + row0 + min(rr, rend - 1)) * LD + NR + cc;
      kv[rr] = RK[off]; gv[rr] = gRK[off];
    }
#pragma unroll
    for (int rr = 0; rr < kWRows; ++rr) {
      const bool ok = cok && rr < rend;
      const float v = ok ? kv[rr] * gv[rr] : 0.f;
      if (ok) gRK[(s * CM + row0 + rr) * LD + NR + col] = v;
      csum += v;
      const float rs = wave_sum(v);
      if (lane == 0 && rs != 0.f) atomicAdd(&r_uf[s * CM + row0 + rr], rs);
    }
    if (cok) atomicAdd(&c_uf[s * B + col], csum);
    const float tot = block_sum<256>(csum, red);
    if (threadIdx.x == 0) atomicAdd(&gtheta[s * (D + 1) + D], 2.f * tot);
    return;
  }
  // K_uu role: kUuRows consecutive rows of one (s, c) matrix per block, a wave takes every 4th of them
  const int nchunk = (M + kUuRows - 1) / kUuRows;
  const int id = bid;
  const int64_t b = id / nchunk;
  const int i0 = (id % nchunk) * kUuRows, i1 = min(M, i0 + kUuRows);
  const float* K = Kuu + b * M * M;
  const float* gK = gKuu + b * M * M;
  float tot = 0.f;
  // a wave's rows (every 4th of the block's) advance together, two 64-column chunks at a time, all loads of a batch
  // before its stores
  constexpr int RW = kUuRows / 4;
  const int wv = threadIdx.x >> 6;
  float acc[RW], dsum = 0.f;
#pragma unroll
  for (int r = 0; r < RW; ++r) acc[r] = 0.f;
  for (int j0 = 0; j0 < M; j0 += 128) {
    float kv[RW][2], gv[RW][2], kt[RW][2], gt[RW][2];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = min(i0 + wv + 4 * r, M - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = min(j0 + 64 * h + lane, M - 1);
        kv[r][h] = K[(int64_t)i * M + j]; gv[r][h] = gK[(int64_t)i * M + j];
        kt[r][h] = 0.f; gt[r][h] = 0.f;
        if (!sym_guu) { kt[r][h] = K[(int64_t)j * M + i]; gt[r][h] = gK[(int64_t)j * M + i]; }
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = i0 + wv + 4 * r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = j0 + 64 * h + lane;
        if (i < i1 && j < M) {
          // sym_guu: gK_uu is symmetric (it comes out of the Cholesky backward), so W + W^T = 2 W: no transposed reads
          const float v = sym_guu ? 2.f * kv[r][h] * gv[r][h] : kv[r][h] * gv[r][h] + kt[r][h] * gt[r][h];
          // the diagonal counts for gamma only (K_ii = gamma^2: see rbf_w_self_kernel, rbf.hip)
          const bool dg = i == j;
          Wuu[b * M * M + (int64_t)i * M + j] = dg ? 0.f : v;
          acc[r] += dg ? 0.f : v;
          dsum += dg ? v : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = i0 + wv + 4 * r;
    const float a = wave_sum(acc[r]);
    if (i < i1) {
      if (lane == 0) r_uu[b * M + i] = a;
      tot += a;
    }
  }
  tot += wave_sum(dsum);
  // every lane of a wave holds the wave's total: count it once
  const float t = block_sum<256>(lane == 0 ? tot : 0.f, red);
  if (threadIdx.x == 0) atomicAdd(&gtheta[(b / C) * (D + 1) + D], t);
}

// minibatch side of the lengthscale gradient: gtheta[s, d] += w_sd sum_b x_bd^2 c_uf[s, b]; block (bx, by) = 64 columns of D x
// ROWS rows of x; red: [2][4][64] floats of LDS
template <int ROWS>
__device__ __forceinline__ void t0_final_x_body(const float* __restrict__ x, const float* __restrict__ c_uf,
                                                const float* __restrict__ w, float* __restrict__ gtheta, int64_t rows, int D,
                                                int64_t Dp, int S, int bx, int by, float (*red)[4][64]) {
  constexpr int RJ = ROWS / 4;
  const int dx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int d = bx * 64 + dx;
  const bool dok = d < D;
  const int dc = dok ? d : D - 1;
  const int64_t row0 = (int64_t)by * ROWS;
  float xa[RJ];
  int64_t rcl[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int64_t row = row0 + ry + 4 * j;
    rcl[j] = row < rows ? row : rows - 1;
    xa[j] = x[rcl[j] * D + dc] * ((dok && row < rows) ? 1.f : 0.f);
  }
  for (int s = 0; s < S; ++s) {
    const float wv = dok ? w[s * Dp + dc] : 0.f;
    float cu[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) cu[j] = c_uf[(int64_t)s * rows + rcl[j]];
    float th = 0.f;
#pragma unroll
    for (int j = 0; j < RJ; ++j) th += xa[j] * (cu[j] * xa[j]);
    red[s & 1][ry][dx] = th;
    __syncthreads();
    if (ry == 0 && dok) {
      const float t = red[s & 1][0][dx] + red[s & 1][1][dx] + red[s & 1][2][dx] + red[s & 1][3][dx];
      atomicAdd(&gtheta[(int64_t)s * (D + 1) + d], wv * t);
    }
  }
}

// RBF finalisation (rbf.hip), both kernel matrices at once.  grid (ceil(D/64), nzy + nxy), 64 d-columns x 4 row lanes.
//   y-blocks < nzy  (inducing points):  gz[row,d] = -sum_s w_sd ((r_uu + r_uf) z - (P_uu + P_uf))
//                                       gtheta[s,d] += w_sd sum_row z ((r_uu z - P_uu) + (r_uf z - 2 P_uf))
//   y-blocks >= nzy (minibatch side):   gtheta[s,d] += w_sd sum_n c_uf x^2
static __global__ __launch_bounds__(256) void t0_final_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                       const float* __restrict__ r_uu, const float* __restrict__ r_uf,
                                                       const float* __restrict__ c_uf, const float* __restrict__ Puu,
                                                       const float* __restrict__ Puf, const float* __restrict__ w,
                                                       float* __restrict__ gz, float* __restrict__ gtheta,
                                                       int64_t zrows, int64_t xrows, int D, int64_t Dp, int S, int nzy,
                                                       const GvecArgs gv = GvecArgs{}) {
  __shared__ float red[2][4][64];      // double-buffered by sample parity: one barrier per sample
  if (gv.gvec && (int)blockIdx.y >= gv.y0) {      // extra rows of the grid: the packed-Cholesky-vector gradient (only shares the launch)
    t0_gvec_role(((int)blockIdx.y - gv.y0) * (int)gridDim.x + (int)blockIdx.x, gv.vec, gv.Lu, gv.gSu, gv.gLu_part, gv.seeds, gv.gvec,
                 gv.S, gv.C, gv.M, gv.M, 0);
    return;
  }
  if ((int)blockIdx.y >= nzy) {
    t0_final_x_body<kFinRows>(x, c_uf, w, gtheta, xrows, D, Dp, S, (int)blockIdx.x, (int)blockIdx.y - nzy, red);
    return;
  }
  constexpr int RJ = kFinRows / 4;
  const int dx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dx;
  const bool dok = d < D;
  const int dc = dok ? d : D - 1;
  const int64_t rows = zrows;
  const int64_t row0 = (int64_t)blockIdx.y * kFinRows;
  const float* src = z;
  // Every load is unconditional, on clamped indices, and its value masked afterwards: a load inside a per-row branch is
  // followed by its own `s_waitcnt vmcnt(0)`, i.e. one memory round trip per row and sample (8 x S of them in a row).
  float xa[RJ], ga[RJ], msk[RJ];
  int64_t rcl[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int64_t row = row0 + ry + 4 * j;
    msk[j] = (dok && row < rows) ? 1.f : 0.f;
    rcl[j] = row < rows ? row : rows - 1;
  }
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    xa[j] = src[rcl[j] * D + dc] * msk[j];
    ga[j] = 0.f;
  }
  {
    for (int s = 0; s < S; ++s) {
      const float wv = dok ? w[s * Dp + dc] : 0.f;
      float p1[RJ], p2[RJ], r1[RJ], r2[RJ];
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        const int64_t sr = (int64_t)s * rows + rcl[j];
        p1[j] = Puu[sr * D + dc]; p2[j] = Puf[sr * D + dc];
        r1[j] = r_uu[sr]; r2[j] = r_uf[sr];
      }
      float th = 0.f;
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        const float rx1 = r1[j] * xa[j], rx2 = r2[j] * xa[j];
        const float q1 = p1[j] * msk[j], q2 = p2[j] * msk[j];
        ga[j] -= wv * ((rx1 - q1) + (rx2 - q2));
        th += xa[j] * ((rx1 - q1) + (rx2 - 2.f * q2));
      }
      red[s & 1][ry][dx] = th;
      __syncthreads();
      if (ry == 0 && dok) {
        const float t = red[s & 1][0][dx] + red[s & 1][1][dx] + red[s & 1][2][dx] + red[s & 1][3][dx];
        atomicAdd(&gtheta[(int64_t)s * (D + 1) + d], wv * t);
      }
    }
    if (dok) {
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        const int64_t row = row0 + ry + 4 * j;
        if (row < rows) gz[row * D + d] = ga[j];
      }
    }
  }
}


}  // namespace vargp
