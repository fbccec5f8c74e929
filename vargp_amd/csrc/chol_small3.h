// Register-resident Cholesky + inverse factor of one small matrix per workgroup (256 threads = 4 waves, n <= 100).
// Device code only, shared by chol.hip (stand-alone launch) and gemm.hip (the launch that runs the factorisations of
// K_uu next to the K_uf kernel-matrix GEMM).
//
// Algorithm: forward elimination on [A | I] held in place in registers (fp64), which yields L = chol(A) and
// T = L^-1 from one sweep: entry (i, e) holds the Schur-complement entry A_ie until column e has been eliminated (step
// e), afterwards entry (i, e) of the unit-lower inverse.  A step is   v_i <- v_i + m_i * p   for every row i > j, with
// p = row j (the Schur complement stays symmetric, so row j is the pivot vector: inverse row j for e < j, the pivot
// d_j at e = j, column j of A for e > j) and m_i = -A_ij / d_j.  Rows i <= j are finished; row j's frozen tail
// (e > j) IS column j of the unscaled factor (L_ej = A_je / sqrt(d_j)), so L needs no storage of its own.
//
// Mapping (what makes a step cheap): ROWS across waves and register slots, COLUMNS across lanes.
//   wave w holds the rows i = 4k + w in slot k (k < KC); lane l holds column l (set a) and column l + 64 (set b).
//   * a finished row is a register slot, and which slots are finished at step j is known at compile time (all pivot
//     steps are instantiated by template recursion): they drop out of the update -- half the work of a mapping that
//     puts rows on lanes, where finished rows still ride along in every instruction;
//   * the pivot row p is one slot of one wave: that wave writes it to LDS together with the scaled copy q = -p / d_j
//     (one value per lane and set each) and everybody reads q back the same way;
//   * the multiplier of row i is -A_ij / d_j, and A_ij = p_i by symmetry: the update is  v_ie <- v_ie + p_i q_e  with
//     p_i read from the published row as a same-address LDS broadcast (two rows per ds_read2_b64) -- per live row
//     half an LDS instruction and one FMA per column set, nothing else;
//   * the inverse entry (i, j) restarts as 0 + m_i * 1.  Instead of a select in every slot, the pivot is published as
//     1 + d_j: A_ij + p_i q_j = p_i - p_i (1 + d_j) / d_j = m_i, up to eps * d_j relative, in fp64;
//   * look-ahead: the wave holding row j + 1 updates that slot first in step j and publishes it at once (pivot by
//     v_readlane, reciprocal by rcp + Newton), so the one barrier per pivot rarely waits.
// 1/d_j by v_rcp_f64 + two Newton steps; matrix I/O staged through LDS (coalesced global access, transposition for L).
//
// Round 4 (what runs by default, VARGP_CHOL_BLOCK4 = 1): the elimination takes FOUR pivots per barrier (chol4_steps below: the
// block's four rows through LDS once, the 4 x 4 in-block elimination redundantly in every wave, multipliers of the rank-4 update
// by v_readlane from the wave's own copy of the eliminated rows -- no LDS broadcast reads); the arithmetic is a template
// parameter (fp64, or the reference's fp32 inside the merged first-task launch); a row's two column sets are one 2-vector
// (v_pk_fma_f32 in fp32); every load of the I/O phases is unconditional on clamped indices and float4 where the rows allow, L
// and T leave through one staging matrix.  The rank-1 version described above (chol3_publish / chol3_steps) is kept behind
// VARGP_CHOL_BLOCK4 = 0 as the reference point of DESIGN_HISTORY.md's measurements.
#pragma once
#include "common.h"
#include "chol_gram.h"
#include "chol_blk16.h"
#include <math.h>
#include <type_traits>
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

#ifdef VARGP_CHOL_PHASES   // tuning builds: wall-clock (100 MHz) stamps of the phases of chol3_body, matrix 0 and the last one
__device__ unsigned long long g_chol_phase[64];
__device__ int g_chol_phase_last;
#define CHOL_PHASE(i) do { if (threadIdx.x == 0 && (b == 0 || b == phase_last_)) g_chol_phase[(b == 0 ? 0 : 16) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CHOL_PHASE(i) do { } while (0)
#endif

// compile-time loop: f(integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void bm_chol_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    bm_chol_for<B + 1, E>(f);
  }
}

// entry (i, c) of the padding beyond the n x n matrix: an identity block (rows i >= n pivot on 1 and touch nothing)
template <class R> __device__ __forceinline__ R chol_pad(int i, int c, int n) { return (i >= n && i == c) ? R(1) : R(0); }

// a row's two column sets (columns l and l + 64 of lane l) as ONE 2-vector: in fp32 the update of both is a single
// v_pk_fma_f32 (register pair, the multiplier broadcast by op_sel)
template <class R> using chol_v2 = R __attribute__((ext_vector_type(2)));

// 1/d for a pivot in the normal range: hardware estimate + two Newton steps (full fp64 accuracy; the IEEE division
// sequence with its scaling / fix-up steps is three times longer and sits on the critical path of every pivot)
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(x, fma(-d, x, 1.0), x);
  x = fma(x, fma(-d, x, 1.0), x);
  return x;
}
// fp32 chain (R = float, see chol3_body): v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float fast_rcp(float d) {
  float x = __builtin_amdgcn_rcpf(d);
  return fmaf(x, fmaf(-d, x, 1.0f), x);
}

// value of `v` in lane `lane` (compile-time or wave-uniform index) broadcast to the whole wave
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// VARGP_CHOL_FLAGSYNC (tuning experiment): no barrier per pivot -- the publishing wave sets an LDS flag after
// its row, the readers poll it together with the row (LDS serves a wave's instructions in order, so a set flag means the row
// is there).  Four row buffers: row J + 4 is published by the wave that published row J, after it has read row J + 3, whose
// publisher had read row J + 2, ... -- by then every wave has read row J.
#ifndef VARGP_CHOL_FLAGSYNC
#define VARGP_CHOL_FLAGSYNC 0
#endif
// 1: four pivots per barrier (chol4_steps, below); 0: the rank-1 version (one pivot per barrier)
#ifndef VARGP_CHOL_BLOCK4
#define VARGP_CHOL_BLOCK4 1
#endif
// 1: fp32 chains of 64 < n <= 100 run the blocked elimination on the matrix core (chol_blk16.h) wherever the matrix passes through
// the staging area in LDS; 0: the register-resident elimination below everywhere
#ifndef VARGP_CHOL_BLK16
#define VARGP_CHOL_BLK16 1
#endif
constexpr int kCholBufs = VARGP_CHOL_FLAGSYNC ? 4 : 2;
template <class R>
struct Chol3Ctx {
  R* prow;        // [kCholBufs][128]  pivot row p (columns 0..127; the pivot itself stored as 1 + d), buffered by J % kCholBufs
  R* qrow;        // [kCholBufs][128]  q = -p / d, same layout
  R* dpiv;        // [kCholBufs]       the pivot d (<= 0 or NaN: not positive-definite)
  double* sd;     // [n]       all pivots, for the final scaling
  int n, lane, w;
  int* flag;   // [kCholBufs]  FLAGSYNC: J + 1 once row J is in its buffer
};

// Row J (slot J / 4 of wave J % 4) to LDS, final once step J - 1 has updated it: p, the scaled copy q = -p / d_J and
// d_J.  Column J counts as 1 + d_J in both, so that the update below leaves m_i = -p_i / d_J in entry (i, J):
// A_iJ + p_i q_J = p_i - p_i (1 + d) / d = -p_i / d   (A_iJ = p_i by symmetry, to rounding).
template <class R, int KC, int SETS, int J>
__device__ __forceinline__ void chol3_publish(const Chol3Ctx<R>& cx, const chol_v2<R> (&v)[KC]) {
  constexpr bool second = J >= 64;
  constexpr int jl = second ? J - 64 : J;
  R* prow = cx.prow + (J % kCholBufs) * 128;
  R* qrow = cx.qrow + (J % kCholBufs) * 128;
  R pa = v[J / 4].x, pb = v[J / 4].y;
  const R d = lane_bcast(second ? pb : pa, jl);
  const R ndi = -fast_rcp(d);
  if constexpr (second) { if (cx.lane == jl) pb = R(1) + d; } else { if (cx.lane == jl) pa = R(1) + d; }
  prow[cx.lane] = pa;
  qrow[cx.lane] = pa * ndi;
  if constexpr (SETS == 2) {
    prow[64 + cx.lane] = pb;
    qrow[64 + cx.lane] = pb * ndi;
  }
  if (cx.lane == 0) { cx.dpiv[J % kCholBufs] = d; cx.sd[J] = (double)d; }
#if VARGP_CHOL_FLAGSYNC
  asm volatile("" ::: "memory");      // (ordering for the compiler; the LDS itself serves a wave's writes in order)
  if (cx.lane == 0) cx.flag[J % kCholBufs] = J + 1;
  asm volatile("" ::: "memory");
#endif
}

template <class R, int KC, int SETS, int J>
__device__ __forceinline__ void chol3_steps(const Chol3Ctx<R>& cx, chol_v2<R> (&v)[KC], int& fail VARGP_STAMP_PARAMS) {
  if constexpr (J < 4 * KC && J < 64 * SETS) {
    constexpr int kj = J / 4, wj = J % 4;                    // slot / wave of the pivot row
    constexpr int k1 = (J + 1) / 4, w1 = (J + 1) % 4;        // ... of the next one
    constexpr bool has_next = k1 < KC && J + 1 < 64 * SETS;
    if (J >= cx.n || fail) return;                           // uniform
    const int lane = cx.lane, w = cx.w;
    STAMP(0);
#if !VARGP_CHOL_FLAGSYNC
    __syncthreads();                                         // row J is in LDS
#endif
    STAMP(2);
    // every LDS read of the step up front: the pivot, this lane's columns of q, and -- same address in every lane --
    // the entries p_i of the rows i = 4k + w this wave still updates (p_i = A_iJ by symmetry: the multiplier of row i)
    const R* qrow = cx.qrow + (J % kCholBufs) * 128;
    const R* pw = cx.prow + (J % kCholBufs) * 128 + w;
    R d, qa, qb = R(0), pr[KC];
#if VARGP_CHOL_FLAGSYNC
    int seen;
    do {       // the flag first, the row behind it in the same batch of reads; again if the flag was not up yet
      asm volatile("" ::: "memory");        // compiler barriers only: every round re-reads, the flag read stays first;
      seen = cx.flag[J % kCholBufs];        // the LDS serves a wave's reads in order, no wait between them
      asm volatile("" ::: "memory");
      d = cx.dpiv[J % kCholBufs];
      qa = qrow[lane];
      if constexpr (SETS == 2) qb = qrow[64 + lane];
#pragma unroll
      for (int k = kj; k < KC; ++k) pr[k] = pw[4 * k];
      asm volatile("" ::: "memory");
    } while (__builtin_amdgcn_readfirstlane(seen) != J + 1);
#else
    d = cx.dpiv[J % kCholBufs];
    qa = qrow[lane];
    if constexpr (SETS == 2) qb = qrow[64 + lane];
#pragma unroll
    for (int k = kj; k < KC; ++k) pr[k] = pw[4 * k];
    __builtin_amdgcn_sched_group_barrier(0x100, (KC - kj + 1) / 2 + 3, 0);   // DS reads first
#endif
    if (!(d > R(0))) { fail = J + 1; return; }               // uniform
    // rows i = 4k + w > J only: slots below kj are finished for every wave, slot kj for the waves w <= wj
    if (w <= wj) pr[kj] = R(0);
    STAMP(3);
    const chol_v2<R> q2 = {qa, qb};
#define VARGP_CHOL3_UPDATE(k)                                                \
    do {                                                                       \
      if constexpr (SETS == 2) {                                               \
        const chol_v2<R> p2_ = {pr[k], pr[k]};                                 \
        v[k] = __builtin_elementwise_fma(p2_, q2, v[k]);                       \
      } else {                                                                 \
        v[k].x = fma(pr[k], qa, v[k].x);                                       \
      }                                                                        \
    } while (0)
    // look-ahead: the next pivot row first, published at once
    if constexpr (has_next) {
      VARGP_CHOL3_UPDATE(k1);
      if (w == w1 && J + 1 < cx.n) chol3_publish<R, KC, SETS, J + 1>(cx, v);
    }
    STAMP(1);
#pragma unroll
    for (int k = kj; k < KC; ++k) {
      if (!(has_next && k == k1)) VARGP_CHOL3_UPDATE(k);
    }
#undef VARGP_CHOL3_UPDATE
#ifdef VARGP_CHOL_STAMPS
    asm volatile("" ::"v"(v[KC - 1].x), "v"(v[KC - 1].y));
    STAMP(4);
#endif
    chol3_steps<R, KC, SETS, J + 1>(cx, v, fail VARGP_STAMP_ARGS);
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Blocked variant of the elimination (round 4): FOUR pivots per barrier.  The rows 4K .. 4K+3 of block K are slot K of the
// four waves.  Each wave publishes its row of the block to LDS (2 values per lane); after ONE barrier every wave reads all
// four rows at its own columns and runs the 4 x 4 in-block elimination redundantly (the uniform entries it needs -- the
// pivots and the in-block multipliers -- are lanes 4K .. 4K+3 of those rows: v_readlane with compile-time lane numbers).  Every
// wave then owns the four ELIMINATED pivot rows p_a and their scaled copies q_a = -p_a / d_a across its lanes, and the
// rank-4 update of its live rows i = 4k + w (k > K) needs nothing from LDS any more: the multiplier of row i for pivot a is
// p_a[i] (symmetry of the Schur complement, as in the rank-1 version), which is lane i of the wave's own copy of p_a -- one
// v_readlane into an SGPR and one FMA (fp32: one v_pk_fma_f32 for both column sets) per row and pivot.  Against the rank-1
// version per four pivots: 1 barrier instead of 4, 16 LDS reads per wave instead of ~50 contended broadcast reads, one
// LDS round trip instead of four.  Look-ahead as before: the rows of block K + 1 are updated and published first.
// ------------------------------------------------------------------------------------------------------------------------------
template <class R>
struct Chol4Ctx {
  R* rows;        // [2][4][128]  the four rows of block K (as published, before the in-block elimination), buffer K % 2
  int n, lane, w;
};

template <class R, int KC, int SETS, int K>
__device__ __forceinline__ void chol4_publish(const Chol4Ctx<R>& cx, const chol_v2<R> (&v)[KC]) {
  R* dst = cx.rows + ((K & 1) * 4 + cx.w) * 128;
  dst[cx.lane] = v[K].x;
  if constexpr (SETS == 2) dst[64 + cx.lane] = v[K].y;
}

// value of lane `lane` (wave-uniform, in an SGPR) of `v`
__device__ __forceinline__ float lane_pick(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_pick(double v, int lane) { return lane_bcast(v, lane); }

template <class R, int KC, int SETS, int K0>
__device__ __forceinline__ void chol4_steps(const Chol4Ctx<R>& cx, chol_v2<R> (&v)[KC], int& fail) {
  if constexpr (K0 < KC && 4 * K0 < 64 * SETS) {
    constexpr int j0 = 4 * K0;
    if (j0 >= cx.n || fail) return;                          // uniform
    const int lane = cx.lane;
    const int w = __builtin_amdgcn_readfirstlane(cx.w);
    __syncthreads();                                         // the four rows of block K0 are in LDS
    const R* src = cx.rows + (K0 & 1) * 4 * 128;
    chol_v2<R> p[4], q[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      p[a].x = src[a * 128 + lane];
      p[a].y = SETS == 2 ? src[a * 128 + 64 + lane] : R(0);
    }
    // ---- in-block elimination: p[a] becomes the final pivot row j0 + a, q[a] = -p[a] / d_a with column j0 + a counted as
    //      1 + d_a (the inverse's entry (i, j) then restarts as m_i = -A_ij / d_j, see the rank-1 version) ----------------------
    // (rows / columns beyond n carry an identity block -- see the loads in chol3_body -- so every pivot of a block is a real
    // one: d = 1, no multipliers; a non-positive pivot is recorded and the block finished on garbage: one exit per block)
    bm_chol_for<0, 4>([&](auto ai) {
      constexpr int a = decltype(ai)::value;
      constexpr int j = j0 + a;
      constexpr bool second = j >= 64;
      constexpr int jl = second ? j - 64 : j;
      const R d = lane_bcast(second ? p[a].y : p[a].x, jl);
      fail = (fail == 0 && !(d > R(0))) ? j + 1 : fail;      // uniform (every wave computes the same d)
      const R ndi = -fast_rcp(d);
      const R qd = ndi * (R(1) + d);
      q[a].x = p[a].x * ndi;
      q[a].y = p[a].y * ndi;
      if constexpr (second) q[a].y = lane == jl ? qd : q[a].y; else q[a].x = lane == jl ? qd : q[a].x;
      bm_chol_for<a + 1, 4>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        const R c = lane_bcast(second ? p[b].y : p[b].x, jl);         // A_{j0+b, j}: the multiplier of block row b
        if constexpr (SETS == 2) {
          const chol_v2<R> c2 = {c, c};
          p[b] = __builtin_elementwise_fma(c2, q[a], p[b]);
        } else {
          p[b].x = fma(c, q[a].x, p[b].x);
        }
      });
    });
    if (fail) return;                                        // uniform
    // the owner of row j0 + w keeps the eliminated row (finished: inverse entries | d | column of the unscaled factor)
    v[K0] = w == 0 ? p[0] : (w == 1 ? p[1] : (w == 2 ? p[2] : p[3]));
    // ---- rank-4 update of the live rows: slot k > K0 of every wave; multiplier of row i = 4k + w for pivot a: p[a][i] ------
    // (all four multipliers into SGPRs first, then the FMAs: a VALU read of an SGPR right behind the v_readlane that wrote
    // it costs wait states)
#define VARGP_CHOL4_UPDATE(k)                                                                       \
    do {                                                                                              \
      const int li_ = 4 * ((k) & 15) + w;                                                             \
      R c_[4];                                                                                        \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) c_[a] = lane_pick(4 * (k) >= 64 ? p[a].y : p[a].x, li_); \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                 \
        if constexpr (SETS == 2) {                                                                    \
          const chol_v2<R> c2_ = {c_[a], c_[a]};                                                      \
          v[k] = __builtin_elementwise_fma(c2_, q[a], v[k]);                                          \
        } else {                                                                                      \
          v[k].x = fma(c_[a], q[a].x, v[k].x);                                                        \
        }                                                                                             \
      }                                                                                               \
    } while (0)
    constexpr int k1 = K0 + 1;
    constexpr bool has_next = k1 < KC && 4 * k1 < 64 * SETS;
    if constexpr (has_next) {                                // look-ahead: the next block's rows first, published at once
      VARGP_CHOL4_UPDATE(k1);
      if (4 * k1 < cx.n) chol4_publish<R, KC, SETS, k1>(cx, v);
    }
#if !defined(VARGP_CHOL_EXP) || VARGP_CHOL_EXP != 2      // (tuning builds: 2 = no bulk update -- timing only, wrong results)
#pragma unroll
    for (int k = K0 + 2; k < KC; ++k) VARGP_CHOL4_UPDATE(k);
#endif
#undef VARGP_CHOL4_UPDATE
    chol4_steps<R, KC, SETS, K0 + 1>(cx, v, fail);
  }
}

template <int KC> constexpr int chol3_stage_floats() { return 4 * KC * (4 * KC + 1); }

// One matrix (index b of the batch).  n <= 64: SETS = 1 (columns = lanes); n <= 128 and n <= 4 KC: SETS = 2.
// R = the arithmetic of the elimination: double (default; more accurate than the reference's fp32 LAPACK path on
// ill-conditioned matrices) or float (the reference's own arithmetic, gp_utils.py:5-11 is torch.cholesky in fp32: half the LDS
// bytes per broadcast, a one-instruction reciprocal, full-rate FMAs -- selected per launch, see launch_chol_rbf_gemm_impl).
template <int KC, int SETS, class R = double>
__device__ __forceinline__ void chol3_body(const int64_t b, const float* __restrict__ A, int lda, int64_t strideA, float eps,
                                           float* __restrict__ L, int ldl, int64_t strideL, float* __restrict__ T, int ldt,
                                           int64_t strideT, float* __restrict__ logdet, int32_t* __restrict__ info,
                                           int info_base, int n, int logdet_accumulate, float* __restrict__ stage,
                                           const CholExtra* extra = nullptr) {
  // `stage` = chol3_stage_floats<KC>() floats of LDS owned by the kernel (the matrix on its way in, L and T on their
  // way out), so that a kernel with other roles can hand over LDS it has anyway
  constexpr int NP = 4 * KC + 4;
#if VARGP_CHOL_BLOCK4
  __shared__ R rows4[2][4][128];               // the four rows of a pivot block, double-buffered
#else
  __shared__ R prow[kCholBufs][128], qrow[kCholBufs][128];
  __shared__ R dpiv[kCholBufs];
  __shared__ int flag[4];
#endif
  __shared__ double sd[NP], sq[NP];
  __shared__ float red[4];
  constexpr int LS = 4 * KC + 1;               // odd row stride: column-wise access hits distinct banks too

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ca = lane, cb = lane + 64;       // set a: columns 0..63, set b: columns 64..n-1
  const bool minea = ca < n, mineb = SETS == 2 && cb < n;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;

  const int di = 256 / n, dj = 256 - di * n;   // element e -> e + 256 without a division per element
  chol_v2<R> v[KC];
  // fp32, 64 < n: the blocked elimination on the matrix core takes the matrix from the staging area (chol_blk16.h)
  constexpr bool kBlk = VARGP_CHOL_BLK16 && std::is_same<R, float>::value && KC == 25 && SETS == 2;
  bool in_stage = false;                       // (uniform) the input path has left the whole matrix (lower triangle valid) in `stage`
#ifdef VARGP_CHOL_PHASES
  const int64_t phase_last_ = (int64_t)gridDim.x < 0 ? 0 : g_chol_phase_last;
#endif
  CHOL_PHASE(0);
  bool gram_done = false;
  if constexpr (KC == 25 && SETS == 2) {
  if (extra && extra->gram_z && b < extra->first) {
    gram_done = true;
    // The workgroup builds its kernel matrix itself (chol_gram.h): weighted Gram matrix of the class's inducing points on the MFMA
    // (panels inside `stage`), then K_ij = g2 exp(-(G_ii + G_jj - 2 G_ij) / 2), exactly g2 on the diagonal -- the formula of the
    // partial-sum path below -- out to Kout (float4 rows) and into `stage`, from where every thread picks its entries.
    static_assert(chol3_stage_floats<KC>() >= kCgLdsFloats, "staging matrix too small for the Gram panels");
    const int pc = extra->part_C;
    const int sidx = (int)(b / pc), cidx = (int)(b - (int64_t)sidx * pc);
    const float* zc = extra->gram_z + (int64_t)cidx * n * extra->gram_D;
    const float* wv = extra->gram_w + (int64_t)sidx * extra->gram_Dp;
    float* Kout = extra->Kout + b * (int64_t)n * n;
    float* diag = reinterpret_cast<float*>(sd);                       // (the pivots' array: free until the elimination has run)
    {
      cg_f32x4 acc[7];
      if (w == 0) cg_gram<0>(zc, wv, n, extra->gram_D, stage, acc, tid);
      else if (w == 1) cg_gram<1>(zc, wv, n, extra->gram_D, stage, acc, tid);
      else if (w == 2) cg_gram<2>(zc, wv, n, extra->gram_D, stage, acc, tid);
      else cg_gram<3>(zc, wv, n, extra->gram_D, stage, acc, tid);
      // (cg_gram ends on a barrier: nobody reads the panels any more)
      if (w == 0) cg_store<0>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
      else if (w == 1) cg_store<1>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
      else if (w == 2) cg_store<2>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
      else cg_store<3>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
    }
    __syncthreads();
    CHOL_PHASE(5);
    {
      const float gam = extra->g2[sidx];
      constexpr int NQ = (4 * KC * KC + 255) / 256;
      const int n4 = n >> 2, tot4 = n * n4;
      const int di4 = 256 / n4, dj4 = 256 - di4 * n4;
      int i = tid / n4, j4 = tid - i * n4;
      // every LDS read of the thread first (each element is read by the thread that overwrites it; the diagonal has its own copy)
      float gij[NQ][4], gi_[NQ], gj_[NQ][4];
      int qi[NQ], qj[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const bool ok = tid + 256 * u < tot4;
        qi[u] = ok ? i : -1; qj[u] = ok ? (j4 << 2) : 0;
        const int ii = ok ? i : 0, jj = qj[u];
        gi_[u] = diag[ii];
#pragma unroll
        for (int t = 0; t < 4; ++t) { gij[u][t] = stage[ii * LS + jj + t]; gj_[u][t] = diag[jj + t]; }
        i += di4; j4 += dj4;
        if (j4 >= n4) { j4 -= n4; ++i; }
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int ii = qi[u], jj = qj[u];
        if (ii >= 0) {
          float4 k4;
          k4.x = ii == jj ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u][0] - 2.f * gij[u][0]));
          k4.y = ii == jj + 1 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u][1] - 2.f * gij[u][1]));
          k4.z = ii == jj + 2 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u][2] - 2.f * gij[u][2]));
          k4.w = ii == jj + 3 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u][3] - 2.f * gij[u][3]));
          *reinterpret_cast<float4*>(Kout + (int64_t)ii * n + jj) = k4;
          float* sp = stage + ii * LS + jj;
          sp[0] = k4.x; sp[1] = k4.y; sp[2] = k4.z; sp[3] = k4.w;
        }
      }
    }
    CHOL_PHASE(8);
    __syncthreads();
    CHOL_PHASE(9);
    if constexpr (kBlk) { in_stage = true; } else
    {
      // every LDS read first, then the selects: written as one loop the compiler waits for each row's pair of reads before
      // it issues the next (25 LDS round trips in a row: 1.6 us of the chain's load phase)
      float sa_[KC], sb_[KC];
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int ic = min(4 * k + w, n - 1);
        sa_[k] = stage[ic * LS + min(ca, n - 1)]; sb_[k] = stage[ic * LS + min(cb, n - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int i = 4 * k + w;
        v[k].x = (i < n && minea) ? (R)sa_[k] + (i == ca ? (R)eps : R(0)) : chol_pad<R>(i, ca, n);
        v[k].y = (i < n && mineb) ? (R)sb_[k] + (i == cb ? (R)eps : R(0)) : chol_pad<R>(i, cb, n);
      }
    }
    CHOL_PHASE(7);
    if (!in_stage) __syncthreads();      // `stage` is reused for the results
  }
  }
  if constexpr (kBlk) {
    if (!gram_done && extra && extra->su_Lu && b >= extra->first) {
      // S_u = Lu Lu^T (gp_utils.py:182: the covariance of q(u) from its Cholesky vector) as a Gram matrix on the matrix core, by the
      // workgroup that factorises it: the 100-long dot products of the prologue's S_u role (391 slow workgroups of the front
      // launch, whose slots the norm role waited for) are gone
      gram_done = true;
      const float* lu = extra->su_Lu + (b - extra->first) * (int64_t)n * n;
      float* diag = reinterpret_cast<float*>(sd);
      {
        cg_f32x4 acc[7];
        if (w == 0) cg_gram<0, true>(lu, nullptr, n, n, stage, acc, tid);
        else if (w == 1) cg_gram<1, true>(lu, nullptr, n, n, stage, acc, tid);
        else if (w == 2) cg_gram<2, true>(lu, nullptr, n, n, stage, acc, tid);
        else cg_gram<3, true>(lu, nullptr, n, n, stage, acc, tid);
        if (w == 0) cg_store<0>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
        else if (w == 1) cg_store<1>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
        else if (w == 2) cg_store<2>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
        else cg_store<3>(acc, stage, LS, diag, n, lane, reinterpret_cast<float*>(sq));
      }
      __syncthreads();
      in_stage = true;
    }
  }
  if (gram_done) {
  } else if (extra && extra->part && b < extra->first) {
    // K-split partial Gram matrices -> kernel matrix on the way in (CholExtra, common.h).  Summation order and formula of
    // t0_combine_norm_kernel (elbo_t0.hip).  Loads on clamped indices, all of a row's in flight together.
    const float* part = extra->part + b * (int64_t)n * n;
    const int nsplit = extra->nsplit;
    const int64_t sS = extra->sSplit;
    float* Kout = extra->Kout + b * (int64_t)n * n;
    if (tid < n) {
      float g = 0.f, v[kCholPartMax];
#pragma unroll
      for (int q = 0; q < kCholPartMax; ++q) v[q] = part[min(q, nsplit - 1) * sS + (int64_t)tid * (n + 1)];
#pragma unroll
      for (int q = 0; q < kCholPartMax; ++q) if (q < nsplit) g += v[q];
      stage[tid] = g;
    }
    __syncthreads();
    CHOL_PHASE(5);
    const float gam = extra->g2[b / extra->part_C];
    if ((n & 3) == 0 && (sS & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(Kout) & 15) == 0) {
      // Wide path: the whole matrix as float4 (16 KC^2 / 1024 per thread and split, all in flight together: a wave has at most
      // 64 memory operations outstanding, so 200 dword loads per thread are more than three full round trips), kernel values
      // on the float4 elements, K out as float4 rows, and into LDS, from where every thread picks its (row, column) entries.
      constexpr int NQ = (4 * KC * KC + 255) / 256;
      const int n4 = n >> 2, tot4 = n * n4;
      const int di4 = 256 / n4, dj4 = 256 - di4 * n4;
      float4 acc[NQ];
      int qi[NQ], qj[NQ];
      {
        int i = tid / n4, j = tid - i * n4;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
          const bool ok = tid + 256 * u < tot4;
          qi[u] = ok ? i : -1; qj[u] = j << 2;
          const int64_t off = ok ? (int64_t)i * n + (j << 2) : 0;
          float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < kCholPartMax; ++q) {
            // (the unused splits are loaded too -- split 0 again -- and multiplied away: a load the compiler can move into
            // a branch on `use` is followed by s_waitcnt vmcnt(0), i.e. every load becomes its own round trip)
            const bool use = q < nsplit;
            const float m = use ? 1.f : 0.f;
            const float4 t4 = *reinterpret_cast<const float4*>(part + (use ? q : 0) * sS + off);
            a4.x = fmaf(t4.x, m, a4.x); a4.y = fmaf(t4.y, m, a4.y); a4.z = fmaf(t4.z, m, a4.z); a4.w = fmaf(t4.w, m, a4.w);
          }
          acc[u] = a4;
          i += di4; j += dj4;
          if (j >= n4) { j -= n4; ++i; }
        }
      }
#ifdef VARGP_CHOL_PHASES
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CHOL_PHASE(6);
#endif
      // the norms move out of the staging area (the matrix goes there): each thread keeps the ones its elements need
      float gi_[NQ];
      float4 gj_[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int i = max(qi[u], 0), j = qj[u];
        gi_[u] = stage[i];
        gj_[u] = make_float4(stage[j], stage[j + 1], stage[j + 2], stage[j + 3]);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int i = qi[u], j = qj[u];
        if (i >= 0) {
          float4 k4;
          k4.x = i == j ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u].x - 2.f * acc[u].x));
          k4.y = i == j + 1 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u].y - 2.f * acc[u].y));
          k4.z = i == j + 2 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u].z - 2.f * acc[u].z));
          k4.w = i == j + 3 ? gam : gam * expf(-0.5f * (gi_[u] + gj_[u].w - 2.f * acc[u].w));
          *reinterpret_cast<float4*>(Kout + (int64_t)i * n + j) = k4;
          float* sp = stage + i * LS + j;
          sp[0] = k4.x; sp[1] = k4.y; sp[2] = k4.z; sp[3] = k4.w;
        }
      }
      CHOL_PHASE(8);
      __syncthreads();
      CHOL_PHASE(9);
      if constexpr (kBlk) { in_stage = true; } else
      {
        // every LDS read first, then the selects: written as one loop the compiler waits for each row's pair of reads before
        // it issues the next (25 LDS round trips in a row: 1.6 us of the chain's load phase)
        float sa_[KC], sb_[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int ic = min(4 * k + w, n - 1);
          sa_[k] = stage[ic * LS + min(ca, n - 1)]; sb_[k] = stage[ic * LS + min(cb, n - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int i = 4 * k + w;
          v[k].x = (i < n && minea) ? (R)sa_[k] + (i == ca ? (R)eps : R(0)) : chol_pad<R>(i, ca, n);
          v[k].y = (i < n && mineb) ? (R)sb_[k] + (i == cb ? (R)eps : R(0)) : chol_pad<R>(i, cb, n);
        }
      }
    } else {
    const int cac = min(ca, n - 1), cbc = min(cb, n - 1);
    const float gja = stage[cac], gjb = stage[cbc];
    // all partial sums of the thread's rows in flight first (the stores to Kout below would otherwise fence every row's
    // loads behind the previous row's stores: KC dependent round trips), then the kernel values
    float ga_[KC], gb_[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w, ic = min(i, n - 1);
      float pa[kCholPartMax], pb[kCholPartMax];
#pragma unroll
      for (int q = 0; q < kCholPartMax; ++q) {
        const float* pq = part + min(q, nsplit - 1) * sS + (int64_t)ic * n;
        pa[q] = pq[cac];
        if (SETS == 2) pb[q] = pq[cbc];
      }
      float ga = 0.f, gb = 0.f;
#pragma unroll
      for (int q = 0; q < kCholPartMax; ++q)
        if (q < nsplit) { ga += pa[q]; if (SETS == 2) gb += pb[q]; }
      ga_[k] = ga; gb_[k] = gb;
    }
#ifdef VARGP_CHOL_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHOL_PHASE(6);
#endif
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w, ic = min(i, n - 1);
      const float ga = ga_[k], gb = gb_[k];
      const float gii = stage[ic];
      const float ka = i == ca ? gam : gam * expf(-0.5f * (gii + gja - 2.f * ga));
      const float kb = i == cb ? gam : gam * expf(-0.5f * (gii + gjb - 2.f * gb));
      v[k].x = chol_pad<R>(i, ca, n); v[k].y = chol_pad<R>(i, cb, n);
      if (i < n) {
        if (minea) { v[k].x = (R)ka + (i == ca ? (R)eps : R(0)); Kout[(int64_t)i * n + ca] = ka; }
        if (mineb) { v[k].y = (R)kb + (i == cb ? (R)eps : R(0)); Kout[(int64_t)i * n + cb] = kb; }
      }
    }
    }
    CHOL_PHASE(7);
    if (!in_stage) __syncthreads();      // `stage` is reused for the results
  } else if (extra && extra->symmetric_input && !kBlk) {
    // both triangles valid: row i lies across the lanes, coalesced as it is.  Unconditional loads on clamped indices, all in
    // flight together, then selects (a load inside a bounds branch is its own round trip: 2 KC of them)
    float ra[KC], rb[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int64_t ro = (int64_t)min(4 * k + w, n - 1) * lda;
      ra[k] = A[ro + min(ca, n - 1)];
      rb[k] = SETS == 2 ? A[ro + min(cb, n - 1)] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      v[k].x = (i < n && minea) ? (R)ra[k] + (i == ca ? (R)eps : R(0)) : chol_pad<R>(i, ca, n);
      v[k].y = (i < n && mineb) ? (R)rb[k] + (i == cb ? (R)eps : R(0)) : chol_pad<R>(i, cb, n);
    }
  } else {
    // only the lower triangle is trusted: the matrix comes in through LDS (coalesced global reads), then every
    // thread picks its entries (rows 4k + w) mirrored
    if ((n & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) {
      constexpr int NQ = (4 * KC * KC + 255) / 256;      // float4 loads, all in flight before the first LDS store
      const int n4 = n >> 2, tot4 = n * n4;
      const int di4 = 256 / n4, dj4 = 256 - di4 * n4;
      float4 tv[NQ];
      int to[NQ];
      int i = tid / n4, j = tid - i * n4;
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const bool ok = tid + 256 * u < tot4;
        tv[u] = *reinterpret_cast<const float4*>(A + (ok ? (int64_t)i * lda + (j << 2) : 0));
        to[u] = ok ? i * LS + (j << 2) : -1;
        i += di4; j += dj4;
        if (j >= n4) { j -= n4; ++i; }
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u)
        if (to[u] >= 0) { float* sp = stage + to[u]; sp[0] = tv[u].x; sp[1] = tv[u].y; sp[2] = tv[u].z; sp[3] = tv[u].w; }
    } else {
      // every global load of the thread in flight before the first LDS store (a load behind a store waits for it; one
      // round trip per element was 40 round trips): clamped indices, the out-of-range elements dropped at the store
      constexpr int NL = (16 * KC * KC + 255) / 256;
      float tv[NL];
      int to[NL];
      int i = tid / n, j = tid - i * n;
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        const bool ok = i < n;
        tv[u] = A[(int64_t)(ok ? i : n - 1) * lda + (ok ? j : 0)];
        to[u] = ok ? i * LS + j : -1;
        i += di; j += dj;
        if (j >= n) { j -= n; ++i; }
      }
#ifdef VARGP_CHOL_PHASES
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CHOL_PHASE(5);
#endif
#pragma unroll
      for (int u = 0; u < NL; ++u)
        if (to[u] >= 0) stage[to[u]] = tv[u];
    }
    __syncthreads();
    CHOL_PHASE(6);
    if constexpr (kBlk) { in_stage = true; } else
    {
      float sa_[KC], sb_[KC];          // (reads first, selects behind them: see the partial-sum path)
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int ic = min(4 * k + w, n - 1), cac = min(ca, n - 1), cbc = min(cb, n - 1);
        sa_[k] = stage[max(ic, cac) * LS + min(ic, cac)]; sb_[k] = stage[max(ic, cbc) * LS + min(ic, cbc)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int i = 4 * k + w;
        v[k].x = (i < n && minea) ? (R)sa_[k] + (i == ca ? (R)eps : R(0)) : chol_pad<R>(i, ca, n);
        v[k].y = (i < n && mineb) ? (R)sb_[k] + (i == cb ? (R)eps : R(0)) : chol_pad<R>(i, cb, n);
      }
    }
  }

  CHOL_PHASE(1);
  int fail = 0;
  bool blk_done = false;
  if constexpr (kBlk) {
    if (in_stage) {                              // (uniform)
#if VARGP_CHOL_BLOCK4
      float* dump16 = reinterpret_cast<float*>(&rows4[0][0][0]);        // (the row buffers of the register elimination: unused here)
#else
      float* dump16 = reinterpret_cast<float*>(&prow[0][0]);
#endif
      cb16_factor(stage, LS, sq, sd, n, eps, tid, reinterpret_cast<int*>(red), dump16, fail);
      blk_done = true;
    }
  }
  if (!blk_done) {
#if VARGP_CHOL_BLOCK4
  {
    // four pivots per barrier (chol4_steps); the LDS of the rank-1 version's row buffers holds the block's four rows
    const Chol4Ctx<R> c4{&rows4[0][0][0], n, lane, w};
    chol4_publish<R, KC, SETS, 0>(c4, v);
#if !defined(VARGP_CHOL_EXP) || VARGP_CHOL_EXP != 1      // (tuning builds: 1 = no elimination at all -- load / store time only)
    chol4_steps<R, KC, SETS, 0>(c4, v, fail);
#endif
    // the pivots d_i, for the final scaling: entry (i, i) of the finished rows, each in one lane of its owner
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      if (lane == (i & 63) && i < n) sd[i] = (double)(4 * k >= 64 ? v[k].y : v[k].x);
    }
    __syncthreads();
  }
#else
  const Chol3Ctx<R> cx{&prow[0][0], &qrow[0][0], dpiv, sd, n, lane, w, flag};
#if VARGP_CHOL_FLAGSYNC
  if (tid < 4) flag[tid] = 0;
  __syncthreads();
#endif
  if (w == 0) chol3_publish<R, KC, SETS, 0>(cx, v);   // row 0 has no predecessor to publish it
#ifdef VARGP_CHOL_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  chol3_steps<R, KC, SETS, 0>(cx, v, fail VARGP_STAMP_ARGS);
  __syncthreads();
#endif
#ifdef VARGP_CHOL_STAMPS
  if (tid == 0 && b == 0) for (int i = 0; i < 8; ++i) g_chol_stamps[i] = acc_[i];
#endif
  }
  CHOL_PHASE(2);
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
  if (!blk_done) {
  if (tid < n) {   // sq = sqrt(d), sd <- 1 / sqrt(d): once per pivot, not per entry; hardware estimate + two Newton steps
    const double d = sd[tid];                       // (the IEEE sqrt / division sequences are ~10x longer; d > 0 here)
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    sq[tid] = d * r;
    sd[tid] = r;
  }
  __syncthreads();
  // Entry (i, e) of the register file: e < i -> T_ie sqrt(d_i); e == i -> d_i; e > i -> L_ei sqrt(d_i).
  // Both factors leave through ONE staging matrix: entry (i, e), e != i, scaled by 1 / sqrt(d_i) goes to stage[e][i] -- for
  // e > i that is L_ei in the lower triangle, for e < i it is T_ie, kept transposed in the upper triangle; the diagonals are
  // sq (L) and sd (T).  Rows then go out as float4 (fixed trip counts: a store loop with a run-time bound waits for every
  // iteration's stores before the next).
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int i = 4 * k + w;
    if (i < n) {
      const double isi = sd[i];
      if (minea && ca != i) stage[ca * LS + i] = (float)((double)v[k].x * isi);
      if (mineb && cb != i) stage[cb * LS + i] = (float)((double)v[k].y * isi);
    }
  }
  __syncthreads();
  }
  CHOL_PHASE(3);
  const bool diag_only = extra && extra->diag_only_before_first && b < extra->first;   // only diag(L) is wanted (zeros elsewhere)
  // optional extra destination(s) for L (CholExtra, common.h)
  float* xb = nullptr;
  int ldx = 0, ncopy = 0;
  int64_t sxc = 0;
  if (extra && b >= extra->first) {
    xb = extra->base + (b - extra->first) * extra->stride_b;
    ldx = extra->ld; ncopy = extra->ncopy; sxc = extra->stride_copy;
  }
  // (every LDS read below is unconditional -- clamped indices -- and the triangle / diagonal logic is selects: a read the
  // compiler can move into a branch gets its own s_waitcnt, one LDS round trip per element)
  if ((n & 3) == 0 && (ldl & 3) == 0 && (reinterpret_cast<uintptr_t>(L) & 15) == 0 &&
      (!T || ((ldt & 3) == 0 && (reinterpret_cast<uintptr_t>(T) & 15) == 0)) &&
      (!xb || ((ldx & 3) == 0 && (sxc & 3) == 0 && (reinterpret_cast<uintptr_t>(xb) & 15) == 0))) {
    constexpr int NQ = (4 * KC * KC + 255) / 256;
    const int n4 = n >> 2, tot4 = n * n4;
    const int di4 = 256 / n4, dj4 = 256 - di4 * n4;
    int i = tid / n4, j4 = tid - i * n4;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const bool ok = tid + 256 * u < tot4;
      const int ii = ok ? i : 0, j = ok ? (j4 << 2) : 0;
      const float* lr = stage + ii * LS + j;          // L: row ii, columns j .. j+3
      const float* tc = stage + j * LS + ii;          // T (kept transposed): rows j .. j+3 of the staging matrix, column ii
      const float l0 = lr[0], l1 = lr[1], l2 = lr[2], l3 = lr[3];
      const float t0 = tc[0], t1 = tc[LS], t2 = tc[2 * LS], t3 = tc[3 * LS];
      const float dl = (float)sq[ii], dt = (float)sd[ii];
      float4 l4, t4;
      l4.x = j < ii ? (diag_only ? 0.f : l0) : (j == ii ? dl : 0.f);
      l4.y = j + 1 < ii ? (diag_only ? 0.f : l1) : (j + 1 == ii ? dl : 0.f);
      l4.z = j + 2 < ii ? (diag_only ? 0.f : l2) : (j + 2 == ii ? dl : 0.f);
      l4.w = j + 3 < ii ? (diag_only ? 0.f : l3) : (j + 3 == ii ? dl : 0.f);
      t4.x = j < ii ? t0 : (j == ii ? dt : 0.f);
      t4.y = j + 1 < ii ? t1 : (j + 1 == ii ? dt : 0.f);
      t4.z = j + 2 < ii ? t2 : (j + 2 == ii ? dt : 0.f);
      t4.w = j + 3 < ii ? t3 : (j + 3 == ii ? dt : 0.f);
      if (ok) {
        *reinterpret_cast<float4*>(L + (int64_t)ii * ldl + j) = l4;
        if (T) *reinterpret_cast<float4*>(T + (int64_t)ii * ldt + j) = t4;
        // (a store loop with a run-time trip count waits for each round's stores: the first copies are straight-line code)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < ncopy) *reinterpret_cast<float4*>(xb + c * sxc + (int64_t)ii * ldx + j) = l4;
        for (int c = 4; c < ncopy; ++c) *reinterpret_cast<float4*>(xb + c * sxc + (int64_t)ii * ldx + j) = l4;
      }
      i += di4; j4 += dj4;
      if (j4 >= n4) { j4 -= n4; ++i; }
    }
  } else {
    constexpr int NL = (16 * KC * KC + 255) / 256;
    int i = tid / n, j = tid - i * n;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const bool ok = i < n;
      const int ii = ok ? i : 0, jj = ok ? j : 0;
      const float lraw = stage[ii * LS + jj], traw = stage[jj * LS + ii];
      const float lv = jj < ii ? (diag_only ? 0.f : lraw) : (jj == ii ? (float)sq[ii] : 0.f);
      const float tv = jj < ii ? traw : (jj == ii ? (float)sd[ii] : 0.f);
      if (ok) {
        L[(int64_t)ii * ldl + jj] = lv;
        if (T) T[(int64_t)ii * ldt + jj] = tv;
        for (int c = 0; c < ncopy; ++c) xb[c * sxc + (int64_t)ii * ldx + jj] = lv;
      }
      i += di; j += dj;
      if (j >= n) { j -= n; ++i; }
    }
  }
  if (diag_only) return;   // (no logdet in this mode: the merged launch never asks for it)
  CHOL_PHASE(4);
  if (logdet) {   // sum_j log L_jj
    float acc = 0.f;
    for (int j = tid; j < n; j += 256) acc += (float)log(sq[j]);
    const float tot = block_sum<256>(acc, red);
    if (tid == 0) { if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot; }
  }
}

}  // namespace vargp
