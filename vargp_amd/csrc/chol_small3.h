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
__device__ __forceinline__ void chol3_publish(const Chol3Ctx<R>& cx, const R (&va)[KC], const R (&vb)[KC]) {
  constexpr bool second = J >= 64;
  constexpr int jl = second ? J - 64 : J;
  R* prow = cx.prow + (J % kCholBufs) * 128;
  R* qrow = cx.qrow + (J % kCholBufs) * 128;
  R pa = va[J / 4], pb = vb[J / 4];
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
__device__ __forceinline__ void chol3_steps(const Chol3Ctx<R>& cx, R (&va)[KC], R (&vb)[KC],
                                            int& fail VARGP_STAMP_PARAMS) {
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
#define VARGP_CHOL3_UPDATE(k)                                                \
    do {                                                                       \
      va[k] = fma(pr[k], qa, va[k]);                                           \
      if constexpr (SETS == 2) vb[k] = fma(pr[k], qb, vb[k]);                  \
    } while (0)
    // look-ahead: the next pivot row first, published at once
    if constexpr (has_next) {
      VARGP_CHOL3_UPDATE(k1);
      if (w == w1 && J + 1 < cx.n) chol3_publish<R, KC, SETS, J + 1>(cx, va, vb);
    }
    STAMP(1);
#pragma unroll
    for (int k = kj; k < KC; ++k) {
      if (!(has_next && k == k1)) VARGP_CHOL3_UPDATE(k);
    }
#undef VARGP_CHOL3_UPDATE
#ifdef VARGP_CHOL_STAMPS
    asm volatile("" ::"v"(va[KC - 1]), "v"(vb[KC - 1]));
    STAMP(4);
#endif
    chol3_steps<R, KC, SETS, J + 1>(cx, va, vb, fail VARGP_STAMP_ARGS);
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
  __shared__ R prow[kCholBufs][128], qrow[kCholBufs][128];
  __shared__ R dpiv[kCholBufs];
  __shared__ int flag[4];
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
  R va[KC], vb[KC];
  if (extra && extra->part && b < extra->first) {
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
    const float gam = extra->g2[b / extra->part_C];
    const int cac = min(ca, n - 1), cbc = min(cb, n - 1);
    const float gja = stage[cac], gjb = stage[cbc];
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
      const float gii = stage[ic];
      const float ka = i == ca ? gam : gam * expf(-0.5f * (gii + gja - 2.f * ga));
      const float kb = i == cb ? gam : gam * expf(-0.5f * (gii + gjb - 2.f * gb));
      va[k] = R(0); vb[k] = R(0);
      if (i < n) {
        if (minea) { va[k] = (R)ka + (i == ca ? (R)eps : R(0)); Kout[(int64_t)i * n + ca] = ka; }
        if (mineb) { vb[k] = (R)kb + (i == cb ? (R)eps : R(0)); Kout[(int64_t)i * n + cb] = kb; }
      }
    }
    __syncthreads();      // `stage` is reused for the results
  } else if (extra && extra->symmetric_input) {
    // both triangles valid: row i lies across the lanes, coalesced as it is
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      va[k] = R(0); vb[k] = R(0);
      if (i < n) {
        if (minea) va[k] = (R)A[(int64_t)i * lda + ca] + (i == ca ? (R)eps : R(0));
        if (mineb) vb[k] = (R)A[(int64_t)i * lda + cb] + (i == cb ? (R)eps : R(0));
      }
    }
  } else {
    // only the lower triangle is trusted: the matrix comes in through LDS (coalesced global reads), then every
    // thread picks its entries (rows 4k + w) mirrored
    {
      int i = tid / n, j = tid - i * n;
      for (int e = tid; e < n * n; e += 256) {
        stage[i * LS + j] = A[(int64_t)i * lda + j];
        i += di; j += dj;
        if (j >= n) { j -= n; ++i; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      va[k] = R(0); vb[k] = R(0);
      if (i < n) {
        if (minea) {
          const int hi = i > ca ? i : ca, lo = i > ca ? ca : i;
          va[k] = (R)stage[hi * LS + lo] + (i == ca ? (R)eps : R(0));
        }
        if (mineb) {
          const int hi = i > cb ? i : cb, lo = i > cb ? cb : i;
          vb[k] = (R)stage[hi * LS + lo] + (i == cb ? (R)eps : R(0));
        }
      }
    }
  }

  int fail = 0;
  const Chol3Ctx<R> cx{&prow[0][0], &qrow[0][0], dpiv, sd, n, lane, w, flag};
#if VARGP_CHOL_FLAGSYNC
  if (tid < 4) flag[tid] = 0;
  __syncthreads();
#endif
  if (w == 0) chol3_publish<R, KC, SETS, 0>(cx, va, vb);   // row 0 has no predecessor to publish it
#ifdef VARGP_CHOL_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  chol3_steps<R, KC, SETS, 0>(cx, va, vb, fail VARGP_STAMP_ARGS);
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
  // Entry (i, e) of the register file: e < i -> T_ie sqrt(d_i); e == i -> d_i; e > i -> L_ei sqrt(d_i).
  // T first: its rows lie across the lanes, so the stores are coalesced as they are.
  if (T) {
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      if (i < n) {
        const double isi = sd[i];
        if (minea) T[(int64_t)i * ldt + ca] = ca < i ? (float)((double)va[k] * isi) : (ca == i ? (float)isi : 0.f);
        if (mineb) T[(int64_t)i * ldt + cb] = cb < i ? (float)((double)vb[k] * isi) : (cb == i ? (float)isi : 0.f);
      }
    }
  }
  if (extra && extra->diag_only_before_first && b < extra->first) {
    // only diag(L) is wanted: rows straight from registers, zeros off the diagonal
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int i = 4 * k + w;
      if (i < n) {
        const float si = (float)sq[i];
        if (minea) L[(int64_t)i * ldl + ca] = ca == i ? si : 0.f;
        if (mineb) L[(int64_t)i * ldl + cb] = cb == i ? si : 0.f;
      }
    }
    return;   // (no logdet in this mode: the merged launch never asks for it)
  }
  // L is the transpose of the row tails: through LDS (lower entry (e, i) from the tail of row i, zero at (i, e)),
  // then coalesced rows out
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    const int i = 4 * k + w;
    if (i < n) {
      const double si = sq[i], isi = sd[i];
#pragma unroll
      for (int half = 0; half < SETS; ++half) {
        const int e = half ? cb : ca;
        if (half ? mineb : minea) {
          const double v = (double)(half ? vb[k] : va[k]);
          if (e > i) { stage[e * LS + i] = (float)(v * isi); stage[i * LS + e] = 0.f; }
          else if (e == i) stage[i * LS + i] = (float)si;
        }
      }
    }
  }
  __syncthreads();
  // optional extra destination(s) for L (CholExtra, common.h)
  float* xb = nullptr;
  int ldx = 0, ncopy = 0;
  int64_t sxc = 0;
  if (extra && b >= extra->first) {
    xb = extra->base + (b - extra->first) * extra->stride_b;
    ldx = extra->ld; ncopy = extra->ncopy; sxc = extra->stride_copy;
  }
  if ((n & 3) == 0 && (ldl & 3) == 0 && (reinterpret_cast<uintptr_t>(L) & 15) == 0 &&
      (!xb || ((ldx & 3) == 0 && (sxc & 3) == 0 && (reinterpret_cast<uintptr_t>(xb) & 15) == 0))) {
    const int n4 = n >> 2;
    for (int q = tid; q < n * n4; q += 256) {
      const int i = q / n4, j = (q - i * n4) << 2;
      const float* sp = stage + i * LS + j;
      const float4 v4 = make_float4(sp[0], sp[1], sp[2], sp[3]);
      *reinterpret_cast<float4*>(L + (int64_t)i * ldl + j) = v4;
      for (int c = 0; c < ncopy; ++c) *reinterpret_cast<float4*>(xb + c * sxc + (int64_t)i * ldx + j) = v4;
    }
  } else {
    int i = tid / n, j = tid - i * n;
    for (int e = tid; e < n * n; e += 256) {
      const float v = stage[i * LS + j];
      L[(int64_t)i * ldl + j] = v;
      for (int c = 0; c < ncopy; ++c) xb[c * sxc + (int64_t)i * ldx + j] = v;
      i += di; j += dj;
      if (j >= n) { j -= n; ++i; }
    }
  }
  if (logdet) {   // sum_j log L_jj
    float acc = 0.f;
    for (int j = tid; j < n; j += 256) acc += (float)log(sq[j]);
    const float tot = block_sum<256>(acc, red);
    if (tid == 0) { if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot; }
  }
}

}  // namespace vargp
