// Cholesky + inverse factor of one matrix per workgroup (n <= 112) with the trailing updates on the fp64 matrix core.
// Device code only.  EXPERIMENT of round 2, NOT part of the library build: kept as the starting point for a faster pivot
// chain.  Correct (the chol tests of tests/test_hip_ops.py pass with it in place of chol3_body) but not yet faster:
// n = 100, batch 30 on MI355X: L only 55 us, L + T 79 us against 44 / 46 us of chol3_body.  Where its time goes (phase
// knock-out builds): the per-round tile updates 24 us (each round re-reads and re-writes every trailing 16 x 16 tile in
// LDS: 306 tile round trips per matrix -- a left-looking order that keeps a block column's accumulators in registers
// across the K loop would cut that to ~110), the T phase 24 us (its block rows are serialised by barriers and every MFMA
// waits for its own operand reads; block COLUMNS are independent and could run barrier-free, one per wave), output 7 us
// (unbatched), the 4 x 4 factorisations and panel rows only ~2 us.  To try it: include it from chol.hip and launch
// cholb_body with kBlkLdsBytes of dynamic LDS for 50 < n <= 112.
//
// chol3_body eliminates one pivot per barrier (rank-1 updates from registers): n dependent rounds of
// barrier -> LDS round trip -> reciprocal -> publish, ~0.47 us each.  Here the matrix lives in LDS (fp64, 112 x 113) and
// the elimination advances FOUR pivots per round:
//   (1) every thread factorises the 4 x 4 diagonal block redundantly in registers (10 broadcast LDS reads, 4 rsqrt) and
//       inverts it -- no barrier, nobody waits for a publishing wave;
//   (2) thread r turns row r of the 4-wide panel into L (X T44^T: 10 FMAs);                          -- barrier --
//   (3) the trailing matrix takes the rank-4 update as ONE v_mfma_f64_16x16x4_f64 per 16 x 16 lower tile (K = 4 is the
//       instruction's native depth), tiles shared out over the four waves.                            -- barrier --
// n / 4 rounds of two barriers instead of n rounds of one, and the O(n^3) arithmetic runs at the fp64 matrix rate.
// T = L^-1 afterwards: the diagonal 16 x 16 blocks by forward substitution (one thread per column, all blocks at once),
// then block row by block row T[k, j] = -T_kk sum_i L[k, i] T[i, j] on the matrix core (the accumulator tile is the next
// product's B operand as it stands: C/D row = (lane >> 4) + 4 reg is exactly the k index of k-chunk `reg`).
// T is stored transposed in the upper triangle of the same LDS array (the algorithm only reads the lower one), its
// diagonal (1 / L_ii) in a side array.
#pragma once
#include "common.h"
#include <math.h>

namespace vargp {

typedef double f64x4_t __attribute__((ext_vector_type(4)));
constexpr int kBlkNP = 112;                 // padded size (7 tiles of 16)
constexpr int kBlkLd = 113;                 // odd row stride (doubles)
constexpr size_t kBlkLdsBytes = sizeof(double) * ((size_t)kBlkNP * kBlkLd + kBlkNP + 8);

__device__ __forceinline__ double rsqrt_f64(double d) {
  // v_rsq_f64 is good to ~2^-26; ONE Newton step gives ~2^-50, far below the fp32 rounding of the results (the
  // reciprocal square roots sit on the critical path of every round, four in a row)
  double x = __builtin_amdgcn_rsq(d);
  x = x * fma(-0.5 * d * x, x, 1.5);
  return x;
}

// One matrix (batch entry b).  lds: kBlkLdsBytes of dynamic LDS, 8-byte aligned.
__device__ __forceinline__ void cholb_body(const int64_t b, const float* __restrict__ A, int lda, int64_t strideA, float eps,
                                           float* __restrict__ L, int ldl, int64_t strideL, float* __restrict__ T, int ldt,
                                           int64_t strideT, int32_t* __restrict__ info, int info_base, int n,
                                           double* __restrict__ lds) {
  double* Am = lds;                                   // [NP][LD]
  double* dT = lds + kBlkNP * kBlkLd;                 // [NP]  1 / L_ii
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;
  const int NP = (n + 15) & ~15, nblk = NP >> 4;

  // ---- load: lower triangle (+ jitter), identity on the padding, zeros elsewhere
  // (eight global loads per thread in flight, on clamped indices: a load inside the bounds branch would be a memory round
  //  trip of its own, 49 of them in a row)
  for (int e0 = tid; e0 < NP * NP; e0 += 256 * 8) {
    float av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = min(e0 + 256 * u, NP * NP - 1);
      const int i = e / NP, j = e - i * NP;
      av[u] = A[(int64_t)min(i, n - 1) * lda + min(j, n - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < NP * NP) {
        const int i = e / NP, j = e - i * NP;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < n && j <= i) v = (double)av[u] + (i == j ? (double)eps : 0.0);
        Am[i * kBlkLd + j] = v;
      }
    }
  }
  if (tid < kBlkNP) dT[tid] = 1.0;                       // (padding rows keep this)
  __syncthreads();

  int fail = 0;
  const int nstep = (n + 3) >> 2;
  for (int s = 0; s < nstep; ++s) {
    const int k0 = 4 * s;
    // (1) 4 x 4 diagonal block, redundantly in every thread (same-address LDS reads)
    const double* Dk = Am + k0 * kBlkLd + k0;
    const double a00 = Dk[0];
    const double a10 = Dk[kBlkLd], a11 = Dk[kBlkLd + 1];
    const double a20 = Dk[2 * kBlkLd], a21 = Dk[2 * kBlkLd + 1], a22 = Dk[2 * kBlkLd + 2];
    const double a30 = Dk[3 * kBlkLd], a31 = Dk[3 * kBlkLd + 1], a32 = Dk[3 * kBlkLd + 2], a33 = Dk[3 * kBlkLd + 3];
    const double i0 = rsqrt_f64(a00);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = fma(-l10, l10, a11);
    const double i1 = rsqrt_f64(d1);
    const double l21 = fma(-l20, l10, a21) * i1, l31 = fma(-l30, l10, a31) * i1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, a22));
    const double i2 = rsqrt_f64(d2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * i2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
    const double i3 = rsqrt_f64(d3);
    if (!(a00 > 0.0)) fail = k0 + 1;
    else if (!(d1 > 0.0)) fail = k0 + 2;
    else if (!(d2 > 0.0)) fail = k0 + 3;
    else if (!(d3 > 0.0)) fail = k0 + 4;
    if (fail) break;                                   // uniform: every thread read the same values
    // T44 = L44^-1 (lower)
    const double t10 = -i1 * (l10 * i0);
    const double t21 = -i2 * (l21 * i1);
    const double t20 = -i2 * fma(l21, t10, l20 * i0);
    const double t32 = -i3 * (l32 * i2);
    const double t31 = -i3 * fma(l32, t21, l31 * i1);
    const double t30 = -i3 * fma(l32, t20, fma(l31, t10, l30 * i0));
    // (2) panel: row r of L[:, k0:k0+4] = X T44^T
    if (tid < NP && tid >= k0 + 4) {
      double* X = Am + tid * kBlkLd + k0;
      const double x0 = X[0], x1 = X[1], x2 = X[2], x3 = X[3];
      X[0] = x0 * i0;
      X[1] = fma(x1, i1, x0 * t10);
      X[2] = fma(x2, i2, fma(x1, t21, x0 * t20));
      X[3] = fma(x3, i3, fma(x2, t32, fma(x1, t31, x0 * t30)));
    }
    __syncthreads();
    // (3) trailing update: A[r][c] -= sum_k L[r][k0+k] L[c][k0+k] on the lower tiles that contain rows / columns >= k0 + 4
    const int kt = k0 + 4;                            // first trailing row / column
    if (kt < NP) {
      const int cb0 = kt >> 4, m = nblk - cb0, ntile = m * (m + 1) / 2;
      // a wave's tiles (<= 7 of the 28) advance together: every LDS read first, then the MFMAs, then the stores -- tile by
      // tile each one would expose its own LDS round trip.  The tile count of the wave is uniform: scalar branches.
      constexpr int MT = 7;
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      const int cnt = __builtin_amdgcn_readfirstlane((ntile - wv + 3) >> 2);
      double av[MT], bv[MT];
      f64x4_t cv[MT];
      int off[MT];
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        if (u < cnt) {
          const int t = wv + 4 * u;
          int rb = 0, rem = t;                        // tile t -> (rb, cb) in the lower triangle of the m x m block grid
          while (rem > rb) { rem -= rb + 1; ++rb; }
          const int r0 = 16 * (cb0 + rb), c0 = 16 * (cb0 + rem);
          const int ar = r0 + l15, bc = c0 + l15;
          const double a = Am[ar * kBlkLd + k0 + l4], bb = Am[bc * kBlkLd + k0 + l4];
          av[u] = ar >= kt ? -a : 0.0;                 // finished rows / columns of the first tile stay as they are
          bv[u] = bc >= kt ? bb : 0.0;
          off[u] = (r0 + l4) * kBlkLd + c0 + l15;
          const double* Cp = Am + off[u];
          cv[u][0] = Cp[0]; cv[u][1] = Cp[4 * kBlkLd]; cv[u][2] = Cp[8 * kBlkLd]; cv[u][3] = Cp[12 * kBlkLd];
        }
      }
#pragma unroll
      for (int u = 0; u < MT; ++u)
        if (u < cnt) cv[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], cv[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        if (u < cnt) {
          double* Cp = Am + off[u];
          Cp[0] = cv[u][0]; Cp[4 * kBlkLd] = cv[u][1]; Cp[8 * kBlkLd] = cv[u][2]; Cp[12 * kBlkLd] = cv[u][3];
        }
      }
    }
    __syncthreads();
    // the diagonal block itself and 1 / L_ii.  Written only now: the first trailing tile of this round re-writes the
    // finished entries it contains (adding zero), and the next round touches none of these before its own barrier.
    if (tid == 0) {
      double* Dw = Am + k0 * kBlkLd + k0;
      Dw[0] = a00 * i0;
      Dw[kBlkLd] = l10; Dw[kBlkLd + 1] = d1 * i1;
      Dw[2 * kBlkLd] = l20; Dw[2 * kBlkLd + 1] = l21; Dw[2 * kBlkLd + 2] = d2 * i2;
      Dw[3 * kBlkLd] = l30; Dw[3 * kBlkLd + 1] = l31; Dw[3 * kBlkLd + 2] = l32; Dw[3 * kBlkLd + 3] = d3 * i3;
      dT[k0] = i0; dT[k0 + 1] = i1; dT[k0 + 2] = i2; dT[k0 + 3] = i3;
    }
  }
  __syncthreads();

  if (fail) {
    if (tid == 0 && info) { if (info[b] == 0) info[b] = info_base + fail; }
    const float qnan = __builtin_nanf("");
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      L[(int64_t)i * ldl + j] = qnan;
      if (T) T[(int64_t)i * ldt + j] = qnan;
    }
    return;
  }

  if (T) {
    // ---- T, diagonal 16 x 16 blocks: thread = one column of one block, forward substitution
    if (tid < NP) {
      const int kb = tid >> 4, jj = tid & 15, base = 16 * kb;
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i >= jj) {
          double acc = (i == jj) ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (k < i && k >= jj) acc = fma(-Am[(base + i) * kBlkLd + base + k], x[k], acc);
          x[i] = acc * dT[base + i];
        }
      }
      // strictly lower entries, transposed into the upper part of the diagonal tile
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i > jj) Am[(base + jj) * kBlkLd + base + i] = x[i];
    }
    __syncthreads();
    // ---- T, off-diagonal blocks, block row by block row: T[k, j] = -T_kk sum_{i = j}^{k-1} L[k, i] T[i, j]
    auto tblk = [&](int i, int j, int kk, int cc) -> double {      // T[16 i + kk][16 j + cc]
      const int gi = 16 * i + kk, gj = 16 * j + cc;
      if (gi > gj) return Am[gj * kBlkLd + gi];
      return gi == gj ? dT[gi] : 0.0;
    };
    for (int k = 1; k < nblk; ++k) {
      for (int j = wave; j < k; j += 4) {
        f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int i = j; i < k; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double a = Am[(16 * k + l15) * kBlkLd + 16 * i + 4 * q + l4];       // L[16k + r][16i + 4q + kk]
            const double bv = tblk(i, j, 4 * q + l4, l15);                            // T[16i + 4q + kk][16j + c]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
          }
        }
        f64x4_t out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double a = -tblk(k, k, l15, 4 * q + l4);                              // -T_kk[r][4q + kk]
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[q], out, 0, 0, 0);        // B[kk][c] = acc row 4q + kk
        }
        // T[16k + r][16j + c], r = l4 + 4 g, c = l15  ->  transposed position
#pragma unroll
        for (int g = 0; g < 4; ++g) Am[(16 * j + l15) * kBlkLd + 16 * k + l4 + 4 * g] = out[g];
      }
      __syncthreads();
    }
  }

  // ---- output (fp32): L lower with zeros above, T likewise
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    L[(int64_t)i * ldl + j] = j <= i ? (float)Am[i * kBlkLd + j] : 0.f;
    if (T) T[(int64_t)i * ldt + j] = j < i ? (float)Am[j * kBlkLd + i] : (j == i ? (float)dT[i] : 0.f);
  }
}

}  // namespace vargp
