// Register-resident Cholesky + inverse factor of one small matrix per workgroup (256 threads = 4 waves, n <= 100).
// Device code only, shared by chol.hip (stand-alone launch) and gemm.hip (the launch that runs the factorisations of
// K_uu next to the K_uf kernel-matrix GEMM).
//
// Algorithm: forward elimination on [A | I] held in place in registers (fp64), which yields L = chol(A) and
// T = L^-1 from one sweep: slot (i, e) holds the Schur-complement entry A_ie until column e has been eliminated (step
// e), afterwards entry (i, e) of the unit-lower inverse.  The Schur complement stays symmetric, so at step j ROW j of the
// register file is the whole pivot vector (inverse row j for e < j, 1 in place of the pivot d_j, column j of A for
// e > j) and a step is   v_i <- v_i + m_i * p   for every row i > j,   m_i = -A_ij / d_j.
//
// Mapping (what makes a step cheap): COLUMNS across waves, ROWS across lanes.
//   wave w owns the columns e = 4k + w (k < KC); lane l holds row l (set a) and row l + 64 (set b).
//   * the pivot-row entries a wave needs are then the same for all its lanes and sit in the registers of ONE of its
//     own lanes: v_readlane hands them to the FMAs as scalar operands -- no LDS traffic for the pivot row at all (a
//     rows-across-waves mapping moves 20-25 fp64 values per lane per step through LDS and is bound by that);
//   * the multipliers m_i need column j, which lives in one wave (j % 4): that wave computes them (pivot by v_readlane,
//     reciprocal by rcp + Newton) and hands them to the others through LDS, two values per lane.  It does so one step
//     AHEAD: in step j the next pivot column is updated first and its wave prepares step j + 1 while the others are
//     still in the bulk of the update, so the one barrier per pivot rarely waits.
//   All pivot steps are instantiated at compile time (template recursion): every register index, the set holding the
//   pivot row and its lane are static, and the finished rows of set a drop out of the update once j >= 64.
// At the end  L_ie = A_ie(at step e) / sqrt(d_e)  (parked in fp32 registers when column e is eliminated),
// T_ie = v_ie / sqrt(d_i),  T_ii = 1 / sqrt(d_i).
#pragma once
#include "common.h"
#include <math.h>
#ifndef STAMP
#define STAMP(i) do { } while (0)
#endif
#ifdef VARGP_CHOL_STAMPS
#define VARGP_STAMP_PARAMS , unsigned long long (&acc_)[8], unsigned long long& last_
#define VARGP_STAMP_ARGS , acc_, last_
#else
#define VARGP_STAMP_PARAMS
#define VARGP_STAMP_ARGS
#endif

namespace vargp {

// 1/d for a pivot in the normal range: hardware estimate + two Newton steps (full fp64 accuracy; the IEEE division
// sequence with its scaling / fix-up steps is three times longer and sits on the critical path of every pivot)
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(x, fma(-d, x, 1.0), x);
  x = fma(x, fma(-d, x, 1.0), x);
  return x;
}

// value of `v` in lane `lane` (wave-uniform index) broadcast to the whole wave
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

struct Chol3Ctx {
  double* mcol;   // [2][128]  multipliers of set a (0..63) and set b (64..127), double-buffered by step parity
  double* dpiv;   // [2]       pivot d_j (<= 0 or NaN: not positive-definite)
  double* sd;     // [n]       all pivots, for the final scaling
  int n, lane, w;
};

// What the wave holding column J does before step J can start: pivot d_J (from the lane holding row J), its
// reciprocal, the multipliers m_i = -A_iJ / d_J of all rows below (to LDS), and it parks column J of A (the unscaled
// column of L) before the slot is reused for the inverse.
template <int KC, int SETS, int J>
__device__ __forceinline__ void chol3_prepare(const Chol3Ctx& cx, double (&va)[KC], double (&vb)[KC], float (&la)[KC],
                                              float (&lb)[KC]) {
  constexpr int kj = J / 4;
  constexpr bool second = J >= 64;                          // row J lives in set b (rows 64..)
  constexpr int jl = second ? J - 64 : J;
  double* mcol = cx.mcol + (J & 1) * 128;
  const double d = lane_bcast(second ? vb[kj] : va[kj], jl);
  if (cx.lane == 0) { cx.dpiv[J & 1] = d; cx.sd[J] = d; }
  const double di = fast_rcp(d);
  if constexpr (!second) {                                  // rows of set a below the pivot exist only while J < 63
    la[kj] = (float)va[kj];
    mcol[cx.lane] = (cx.lane > J && cx.lane < cx.n) ? -va[kj] * di : 0.0;
  }
  if constexpr (SETS == 2) {
    lb[kj] = (float)vb[kj];
    const int rb = cx.lane + 64;
    mcol[64 + cx.lane] = (rb > J && rb < cx.n) ? -vb[kj] * di : 0.0;
  }
}

template <int KC, int SETS, int J>
__device__ __forceinline__ void chol3_steps(const Chol3Ctx& cx, double (&va)[KC], double (&vb)[KC], float (&la)[KC],
                                            float (&lb)[KC], int& fail VARGP_STAMP_PARAMS) {
  if constexpr (J < 4 * KC && J < 64 * SETS) {
    constexpr int kj = J / 4, wj = J % 4;
    constexpr int k1 = (J + 1) / 4, w1 = (J + 1) % 4;        // slot / wave of the next pivot column
    constexpr bool second = J >= 64;                         // the pivot row lives in set b
    constexpr int jl = second ? J - 64 : J;                  // its lane
    if (J >= cx.n || fail) return;                           // uniform
    const int lane = cx.lane, w = cx.w;
    const bool owner = w == wj;                              // uniform: this wave holds column J
    STAMP(0);
    __syncthreads();                                         // multipliers and pivot of step J are in LDS
    STAMP(2);
    const double* mcol = cx.mcol + (J & 1) * 128;
    const double d = cx.dpiv[J & 1];
    double ma = 0.0, mb = 0.0;
    if constexpr (!second) ma = mcol[lane];                  // J >= 64: every row of set a is above the pivot
    if constexpr (SETS == 2) mb = mcol[64 + lane];
    // this wave's columns of the pivot row come straight out of the registers of the lane that holds row J
    // (v_readlane -> scalar operands of the FMAs); the pivot itself counts as 1
    double pv[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) pv[k] = lane_bcast(second ? vb[k] : va[k], jl);
    if (owner) pv[kj] = 1.0;
    if (!(d > 0.0)) { fail = J + 1; return; }                // uniform
    STAMP(3);
    // slot (i, J) itself restarts as an inverse entry, 0 + m * 1, in the rows below the pivot
#define VARGP_CHOL3_UPDATE(k)                                                                          \
    do {                                                                                                 \
      if constexpr (!second) {                                                                           \
        const double ta = fma(ma, pv[k], va[k]);                                                         \
        va[k] = ((k) == kj && owner && lane > J) ? ma : ta;                                              \
      }                                                                                                  \
      if constexpr (SETS == 2) {                                                                         \
        const double tb = fma(mb, pv[k], vb[k]);                                                         \
        vb[k] = ((k) == kj && owner && lane + 64 > J) ? mb : tb;                                         \
      }                                                                                                  \
    } while (0)
    // look-ahead: the next pivot column first, so that its wave can prepare step J + 1 (pivot, reciprocal, multipliers)
    // while everybody else is still in the bulk of this update; the barrier of step J + 1 then finds them ready
    if constexpr (k1 < KC) VARGP_CHOL3_UPDATE(k1);
    if constexpr (J + 1 < 4 * KC && J + 1 < 64 * SETS) {
      if (w == w1 && J + 1 < cx.n) chol3_prepare<KC, SETS, J + 1>(cx, va, vb, la, lb);
    }
    STAMP(1);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (k != k1) VARGP_CHOL3_UPDATE(k);
    }
#undef VARGP_CHOL3_UPDATE
#ifdef VARGP_CHOL_STAMPS
    asm volatile("" ::"v"(va[0]), "v"(vb[KC - 1]));
    STAMP(4);
#endif
    chol3_steps<KC, SETS, J + 1>(cx, va, vb, la, lb, fail VARGP_STAMP_ARGS);
  }
}

template <int KC> constexpr int chol3_stage_floats() { return 4 * KC * (4 * KC + 1); }

// One matrix (index b of the batch).  n <= 64: SETS = 1 (rows = lanes); n <= 128 and n <= 4 KC: SETS = 2.
template <int KC, int SETS>
__device__ __forceinline__ void chol3_body(const int64_t b, const float* __restrict__ A, int lda, int64_t strideA, float eps,
                                           float* __restrict__ L, int ldl, int64_t strideL, float* __restrict__ T, int ldt,
                                           int64_t strideT, float* __restrict__ logdet, int32_t* __restrict__ info,
                                           int info_base, int n, int logdet_accumulate, float* __restrict__ stage) {
  // `stage` = chol3_stage_floats<KC>() floats of LDS owned by the kernel (the matrix on its way in, L and T on their
  // way out), so that a kernel with other roles can hand over LDS it has anyway
  constexpr int NP = 4 * KC + 4;
  __shared__ double mcol[2][128];
  __shared__ double dpiv[2];
  __shared__ double sd[NP], sq[NP];
  __shared__ float red[4];
  constexpr int LS = 4 * KC + 1;               // odd row stride: lanes (= rows) hit distinct banks

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ra = lane, rb = lane + 64;       // set a: rows 0..63, set b: rows 64..n-1
  const bool minea = ra < n, mineb = SETS == 2 && rb < n;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;

  // the matrix comes in through LDS: coalesced global reads, then every lane picks the entries of its rows (a lane
  // owns whole rows, so reading them straight from global memory would touch one cache line per lane and entry)
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    stage[i * LS + j] = A[(int64_t)i * lda + j];
  }
  __syncthreads();
  double va[KC], vb[KC];
  float la[KC], lb[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int e = 4 * k + w;
    va[k] = 0.0; vb[k] = 0.0; la[k] = 0.f; lb[k] = 0.f;
    if (e < n) {   // only the lower triangle of the input is trusted: mirror it
      if (minea) {
        const int hi = ra > e ? ra : e, lo = ra > e ? e : ra;
        va[k] = (double)stage[hi * LS + lo] + (e == ra ? (double)eps : 0.0);
      }
      if (mineb) {
        const int hi = rb > e ? rb : e, lo = rb > e ? e : rb;
        vb[k] = (double)stage[hi * LS + lo] + (e == rb ? (double)eps : 0.0);
      }
    }
  }

  int fail = 0;
  const Chol3Ctx cx{&mcol[0][0], dpiv, sd, n, lane, w};
  if (w == 0) chol3_prepare<KC, SETS, 0>(cx, va, vb, la, lb);     // step 0 has no predecessor to prepare it
#ifdef VARGP_CHOL_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  chol3_steps<KC, SETS, 0>(cx, va, vb, la, lb, fail VARGP_STAMP_ARGS);
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
  if (tid < n) {   // sq = sqrt(d), sd <- 1 / sqrt(d): one square root and one division per pivot, not per entry
    const double s = sqrt(sd[tid]);
    sq[tid] = s;
    sd[tid] = 1.0 / s;
  }
  __syncthreads();
  // L, then T: entries to LDS by their owners, coalesced rows out
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* out = pass == 0 ? L : T;
    const int ldo = pass == 0 ? ldl : ldt;
    if (out == nullptr) continue;                           // uniform
#pragma unroll
    for (int half = 0; half < SETS; ++half) {
      const int r = half ? rb : ra;
      if (half ? mineb : minea) {
        const double si = sq[r], isi = sd[r];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int e = 4 * k + w;
          if (e < n) {
            float v = 0.f;
            if (e < r) v = pass == 0 ? (float)((double)(half ? lb[k] : la[k]) * sd[e]) : (float)((half ? vb[k] : va[k]) * isi);
            else if (e == r) v = pass == 0 ? (float)si : (float)isi;
            stage[r * LS + e] = v;
          }
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e - i * n;
      out[(int64_t)i * ldo + j] = stage[i * LS + j];
    }
    __syncthreads();
  }
  if (logdet) {   // sum_j log L_jj
    float acc = 0.f;
    for (int j = tid; j < n; j += 256) acc += (float)log(sq[j]);
    const float tot = block_sum<256>(acc, red);
    if (tid == 0) { if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot; }
  }
}

}  // namespace vargp
