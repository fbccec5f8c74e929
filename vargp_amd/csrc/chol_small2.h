// Register-resident Cholesky + inverse factor of one small matrix per workgroup, two rows per thread (256 threads,
// n <= 100).  Device code only, shared by chol.hip (stand-alone launch) and gemm.hip (the launch that runs the
// factorisations of K_uu next to the K_uf kernel-matrix GEMM).  See chol.hip for the algorithm.
#pragma once
#include "common.h"
#include <math.h>
#ifndef VARGP_CHOL_ABL
#define VARGP_CHOL_ABL 0   // timing ablations (wrong results): 1 no FMAs, 2 no pivot-row reads, 3 no pivot-row publish, 4 all
#endif
#ifndef STAMP
#define STAMP(i) do { } while (0)
#endif

namespace vargp {

constexpr int kCholP = 5;       // threads per matrix row

// 1/d for a pivot in the normal range: hardware estimate + two Newton steps (full fp64 accuracy; the IEEE division
// sequence with its scaling / fix-up steps is three times longer and sits on the critical path of every pivot)
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(x, fma(-d, x, 1.0), x);
  x = fma(x, fma(-d, x, 1.0), x);
  return x;
}
// Packed lower-triangular index.
__device__ __forceinline__ int pk(int i, int j) { return i * (i + 1) / 2 + j; }

// One pivot step of chol_inv_small2_kernel, instantiated for every pivot index J (compile-time recursion): the slot
// holding the pivot column, jb = J / P, is then a compile-time register index and a step is about 100 instructions of
// straight-line code.  (Indexing the register file with a run-time group number costs a 2K-long select chain per
// group; that was a quarter of the kernel's run time.)
struct Chol2Ctx {
  double* pbuf; int pstride; double* dpiv; double* sd; float* Lp;
  int n, R, tid, q, part, ra, rb;
  bool minea, mineb;
};
template <int K, int J>
__device__ __forceinline__ void chol2_steps(const Chol2Ctx& cx, double (&va)[K], double (&vb)[K], int& fail
#ifdef VARGP_CHOL_STAMPS
                                            , unsigned long long (&acc_)[8], unsigned long long& last_
#endif
) {
  using F = double;
  constexpr int P = kCholP;
  if constexpr (J < K * P) {
    constexpr int jb = J / P, ps = J % P, j = J;
    if (j >= cx.n || fail) return;                         // uniform
    const int part = cx.part, q = cx.q, ra = cx.ra, rb = cx.rb;
    F* p = cx.pbuf + (j & 1) * cx.pstride;
    STAMP(0);                                              // loop overhead
    const bool second = j >= cx.R;                         // uniform: the pivot row lives in the b set
    const int qj = second ? j - cx.R : j;
    if (__builtin_amdgcn_ballot_w64(q == qj) != 0) {       // only the wave that holds row j
      if (q == qj) {
        const bool piv = part == ps;                       // this lane's slot jb is the pivot d_j: publish 1 in its place
        if (second) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (VARGP_CHOL_ABL == 3 || VARGP_CHOL_ABL == 4) { if (k != jb) continue; }
            p[k * P + part] = (k == jb && piv) ? F(1) : vb[k];
          }
          if (piv) cx.dpiv[j & 1] = vb[jb];
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if (VARGP_CHOL_ABL == 3 || VARGP_CHOL_ABL == 4) { if (k != jb) continue; }
            p[k * P + part] = (k == jb && piv) ? F(1) : va[k];
          }
          if (piv) cx.dpiv[j & 1] = va[jb];
        }
      }
    }
    STAMP(1);                                              // publish (only the owning wave does work)
    __syncthreads();
    STAMP(2);                                              // barrier wait
    // every LDS read of the step is issued up front (one round trip: with one wave per SIMD nothing else hides the
    // latency), then the FMAs consume them in order
    const F d = cx.dpiv[j & 1];
    const F pra = p[cx.minea ? ra : 0], prb = p[cx.mineb ? rb : 0];
    const F* pp = p + part;
    F pv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) pv[k] = (VARGP_CHOL_ABL == 2 || VARGP_CHOL_ABL == 4) ? d : pp[k * P];
    __builtin_amdgcn_sched_group_barrier(0x100, K + 3, 0);   // DS reads first
#ifdef VARGP_CHOL_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(3);                                              // LDS reads landed
#endif
    if (!(d > F(0))) { fail = j + 1; return; }              // uniform
    if (cx.tid == 0) cx.sd[j] = d;
    const F di = fast_rcp(d);                              // every lane for itself: off the publishing wave's path
    const bool belowa = cx.minea && ra > j, belowb = cx.mineb && rb > j;
    if (part == 0) {
      if (belowa) cx.Lp[pk(ra, j)] = (float)pra;            // park column j of A (unscaled L column)
      if (belowb) cx.Lp[pk(rb, j)] = (float)prb;
    }
    const F ma = belowa ? -pra * di : F(0), mb = belowb ? -prb * di : F(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (VARGP_CHOL_ABL == 1 || VARGP_CHOL_ABL == 4) { asm volatile("" ::"v"(pv[k])); if (k != jb) continue; }
      // slot (i, j) itself restarts as an inverse entry: 0 + m * 1
      va[k] = (k == jb && part == ps) ? ma : fma(ma, pv[k], va[k]);
      vb[k] = (k == jb && part == ps) ? mb : fma(mb, pv[k], vb[k]);
    }
#ifdef VARGP_CHOL_STAMPS
    asm volatile("" ::"v"(va[0]), "v"(vb[K - 1]));
    STAMP(4);                                              // multipliers + FMAs
    chol2_steps<K, J + 1>(cx, va, vb, fail, acc_, last_);
#else
    chol2_steps<K, J + 1>(cx, va, vb, fail);
#endif
  }
}

// Same elimination with TWO rows per thread (rows q and q + R, R = ceil(n/2)): a pivot-row value read from LDS feeds
// two FMAs, so the LDS traffic per step halves and a 100 x 100 matrix needs only 256 threads (4 waves, one per SIMD).
// The reciprocal of the pivot is published with the row (one fp64 division per step instead of one per thread).
template <int K>
__device__ __forceinline__ void chol2_body(const int64_t b, const float* __restrict__ A, int lda, int64_t strideA, float eps,
                                           float* __restrict__ L, int ldl, int64_t strideL, float* __restrict__ T, int ldt,
                                           int64_t strideT, float* __restrict__ logdet, int32_t* __restrict__ info,
                                           int info_base, int n, int logdet_accumulate) {
  using F = double;
  constexpr int P = kCholP;
  __shared__ F pbuf[2][K * P + 8];
  __shared__ F dpiv[2];
  __shared__ F sd[K * P + 8];
  __shared__ float Lp[(K * P) * (K * P + 1) / 2];
  __shared__ float red[4];

  const int tid = threadIdx.x;
  const int q = tid / P, part = tid % P;
  const int R = (n + 1) / 2;
  const int ra = q, rb = q + R;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;
  const bool minea = q < R, mineb = q < R && rb < n;

  F va[K], vb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * P + part;
    va[k] = F(0); vb[k] = F(0);
    if (e < n) {   // only the lower triangle of the input is trusted: mirror it
      if (minea) {
        const int hi = ra > e ? ra : e, lo = ra > e ? e : ra;
        va[k] = (F)A[(int64_t)hi * lda + lo] + (e == ra ? (F)eps : F(0));
      }
      if (mineb) {
        const int hi = rb > e ? rb : e, lo = rb > e ? e : rb;
        vb[k] = (F)A[(int64_t)hi * lda + lo] + (e == rb ? (F)eps : F(0));
      }
    }
  }

#ifdef VARGP_CHOL_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  int fail = 0;
  const Chol2Ctx cx{&pbuf[0][0], K * P + 8, dpiv, sd, Lp, n, R, tid, q, part, ra, rb, minea, mineb};
#ifdef VARGP_CHOL_STAMPS
  chol2_steps<K, 0>(cx, va, vb, fail, acc_, last_);
#else
  chol2_steps<K, 0>(cx, va, vb, fail);
#endif
  __syncthreads();
#ifdef VARGP_CHOL_STAMPS
  if (tid == 0 && b == 0) for (int i = 0; i < 8; ++i) g_chol_stamps[i] = acc_[i];
#endif
  if (fail) {
    if (tid == 0 && info) { if (info[b] == 0) info[b] = info_base + fail; }
    const float qnan = __builtin_nanf("");
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      L[(int64_t)i * ldl + j] = qnan;
      if (T) T[(int64_t)i * ldt + j] = qnan;
    }
    if (logdet && tid == 0) logdet[b] = qnan;
    return;
  }
  float ld_acc = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = half ? rb : ra;
    if (half ? mineb : minea) {
      const F si = sqrt(sd[r]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e = k * P + part;
        const F vk = half ? vb[k] : va[k];
        if (e < r) {
          L[(int64_t)r * ldl + e] = (float)((F)Lp[pk(r, e)] / sqrt(sd[e]));
          if (T) T[(int64_t)r * ldt + e] = (float)(vk / si);
        } else if (e == r) {
          L[(int64_t)r * ldl + e] = (float)si;
          if (T) T[(int64_t)r * ldt + e] = (float)(F(1) / si);
          ld_acc += (float)log(si);
        } else if (e < n) {
          L[(int64_t)r * ldl + e] = 0.f;
          if (T) T[(int64_t)r * ldt + e] = 0.f;
        }
      }
    }
  }
  if (logdet) {
    ld_acc = wave_sum(ld_acc);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = ld_acc;
    __syncthreads();
    if (tid == 0) {
      const float tot = red[0] + red[1] + red[2] + red[3];
      if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot;
    }
  }
}

}  // namespace vargp
