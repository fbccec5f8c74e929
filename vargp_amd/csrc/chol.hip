// Batched Cholesky (+ jitter) with explicit inverse factor T = L^-1 and log-determinant
// (reference: gp_utils.cholesky, var_gp/gp_utils.py:5-11, plus every triangular_solve against it).
//
// n <= 100: one workgroup per matrix, the matrix in registers (fp64), forward elimination on [A | I] yields L and
//   T = L^-1 together.  n <= 50: rows across threads, pivot row through LDS (chol_inv_small_kernel, below);
//   50 < n <= 100: columns across waves, rows across lanes, pivot row by v_readlane (chol_small3.h).
// n > 100: blocked right-looking on 96-wide panels; the diagonal blocks use the LDS kernel, the
//   panel solves and trailing updates are MFMA GEMMs (gemm.hip).
// Backward (any n) is four GEMMs (see chol_inv_bwd_impl).
#include "common.h"
#include <vector>
#ifdef VARGP_CHOL_STAMPS   // per-phase cycle accounting of chol3_body (wave 0 of block 0), tuning builds only
__device__ unsigned long long g_chol_stamps[8];
extern "C" void vargp_debug_chol_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_stamps), 64); }
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_[i] += t_ - last_; last_ = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#include <math.h>
#include <functional>
#include <type_traits>
#include <stdlib.h>
#include "chol_small3.h"

namespace vargp {

constexpr int kCholP = 5;        // threads per matrix row of chol_inv_small_kernel
// A co-launched GEMM pays while it is about as long as a diagonal block's pivot chain (~50 us); a much longer one is
// faster on its own with the big tiles (measured: Permuted-MNIST K_uf, 5040 tiles: 481 us merged vs 394 + 57 separate)
static bool co_gemm_is_comparable(const GemmParams& p, int nbatch) {
  return (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nbatch <= 1536;
}
// Packed lower-triangular index.
__device__ __forceinline__ int pk(int i, int j) { return i * (i + 1) / 2 + j; }
constexpr int kSmallMax = 100;  // largest n the register-resident kernel takes (packed triangle <= 512*10)
constexpr int kNbMax = 100;     // widest panel of the blocked algorithm (multiple of 4, <= kSmallMax)
// panel width: 100 (two panels at n = 200, four at 400) unless VARGP_CHOL_PANEL says otherwise (tuning aid)
static int panel_width() {
  static const int v = [] { const char* e = getenv("VARGP_CHOL_PANEL"); const int x = e ? atoi(e) : kNbMax; return (x >= 32 && x <= kNbMax && x % 4 == 0) ? x : kNbMax; }();
  return v;
}

// Outer panel of the two-level blocked factorisation (multiple of the inner width nb): 2 nb from n = 12 nb on, else one level.
// Measured at n = 2048 (MI355X, tests/native/chol_err.py; fp32 LAPACK: L 4.8e-6, T 1.35e-5 relative to fp64):
//   one level      L 7.0e-6  T 2.0e-5   factorisation + inverse of 10 matrices 0.168 of the f32 MFMA peak, Permuted-MNIST t=9 18.8 steps/s
//   outer 2 nb     L 9.6e-6  T 2.8e-5   0.187, 19.6
//   outer 4 nb     L 1.24e-5 T 3.6e-5   0.201, 20.0
// The longer the K of the trailing update, the longer the sequential fp32 accumulation inside the MFMA (K = 100 chunks added
// into the matrix act as a blocked summation): 4 nb is past the "within 2x of fp32 LAPACK" bar the tests hold, 2 nb is inside.
// VARGP_CHOL_PANEL2 (tuning aid): 0 = one level, otherwise the outer width in units of nb.
static int outer_panel_width(int n, int nb) {
  static const int v = [] { const char* e = getenv("VARGP_CHOL_PANEL2"); return e ? atoi(e) : -1; }();
  if (v == 0) return nb;
  if (v > 0) return v * nb < n ? v * nb : nb;
  return n >= 12 * nb ? 2 * nb : nb;
}

template <typename F> __device__ __forceinline__ F rcp_of(F d);
template <> __device__ __forceinline__ double rcp_of<double>(double d) { return fast_rcp(d); }
template <> __device__ __forceinline__ float rcp_of<float>(float d) { return 1.f / d; }

// One workgroup per matrix, n <= kSmallMax.  In-place Gauss-Jordan on [A | I], which yields L = chol(A)
// and T = L^-1 from one elimination.  The matrix lives in REGISTERS, row-wise and FULL (both triangles):
// thread (r, part) = (tid / P, tid % P) owns the entries e = k*P + part (k < K) of row r.
//   slot (i,e) holds the Schur-complement entry A_ie until column e has been eliminated (step e), afterwards
//   entry (i,e) of the unit-lower inverse.  Because the Schur complement stays symmetric, at step j ROW j
//   of the register file is the whole pivot vector: inverse row j for e < j, the pivot d_j at e = j, column
//   j of A (= row j) for e > j.  So a step is
//     publish  the P threads of row j copy their K slots to LDS, no predicates (p[j] := 1, d_j aside);
//     barrier; every row i > j does  v <- v + (-p[i]/d_j) * p[e]  on all its slots: one LDS read + one FMA
//              per slot; slot (i,j) restarts from 0 and becomes the inverse entry; rows i <= j are finished
//              (they run the same FMAs with a zero multiplier).
//   The slot holding column j, j / P, is wave-uniform and changes every P steps: the elimination runs in
//   groups of P statically unrolled steps with that slot kept in a named register, so no register array
//   is ever indexed dynamically.
//   end: L_ie = A_ie(final) / sqrt(d_e), T_ie = v_ie / sqrt(d_i), T_ii = 1/sqrt(d_i).
// F = double: factor and explicit inverse are then at least as accurate as fp32 LAPACK potrf + trsm on
// ill-conditioned K_uu; the chain of n dependent pivots, not the flops, bounds the kernel.
template <typename F, int K>
__global__ __launch_bounds__(512) void chol_inv_small_kernel(const float* __restrict__ A, int lda, int64_t strideA,
                                                             float eps, float* __restrict__ L, int ldl,
                                                             int64_t strideL, float* __restrict__ T, int ldt,
                                                             int64_t strideT, float* __restrict__ logdet,
                                                             int32_t* __restrict__ info, int info_base, int n,
                                                             int logdet_accumulate) {
  constexpr int P = kCholP;
  __shared__ F pbuf[2][K * P + 8];
  __shared__ F dpiv[2], dinv[2];
  __shared__ F sd[K * P + 8];
  __shared__ float Lp[(K * P) * (K * P + 1) / 2];   // final (unscaled) columns of A, packed
  __shared__ float red[8];

  const int tid = threadIdx.x, NT = blockDim.x;
  const int r = tid / P, part = tid % P;
  const int64_t b = blockIdx.x;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;
  const bool mine = r < n;

  F v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * P + part;
    v[k] = F(0);
    if (mine && e < n) {   // only the lower triangle of the input is trusted: mirror it
      const int hi = r > e ? r : e, lo = r > e ? e : r;
      v[k] = (F)A[(int64_t)hi * lda + lo] + (e == r ? (F)eps : F(0));
    }
  }

  int fail = 0;
  for (int jb = 0; jb * P < n && !fail; ++jb) {
    F vcur = F(0);       // slot jb of this thread: the entries (r, jb*P + part), i.e. this group's pivot columns
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k == jb) { asm volatile("" ::: "memory"); vcur = v[k]; }
    }
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
      const int j = jb * P + ps;
      if (j >= n || fail) break;                           // uniform
      F* p = pbuf[j & 1];
      if (__builtin_amdgcn_ballot_w64(r == j) != 0) {       // only the wave that holds row j
        if (r == j) {
#pragma unroll
          for (int k = 0; k < K; ++k) p[k * P + part] = v[k];
          p[jb * P + part] = vcur;                          // the live copy of slot jb
          if (part == ps) { dpiv[j & 1] = vcur; dinv[j & 1] = rcp_of<F>(vcur); p[j] = F(1); }
        }
      }
      __syncthreads();
      // every LDS read of the step is issued up front (one round trip; with one or two waves per SIMD nothing else
      // hides the latency), then the FMAs consume them in order
      const F d = dpiv[j & 1], di = dinv[j & 1];
      const F pr = p[mine ? r : 0];                         // A_rj (= A_jr)
      const F* pp = p + part;
      F pv[K];
#pragma unroll
      for (int k = 0; k < K; ++k) pv[k] = pp[k * P];
      const F pc = pp[jb * P];
      __builtin_amdgcn_sched_group_barrier(0x100, K + 4, 0);   // DS reads first
      if (!(d > F(0))) { fail = j + 1; break; }             // uniform
      if (tid == 0) sd[j] = d;
      const bool below = mine && r > j;
      if (below && part == 0) Lp[pk(r, j)] = (float)pr;     // park column j of A (unscaled L column)
      const F m = below ? -pr * di : F(0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = fma(m, pv[k], v[k]);
      // the group's slot: same update, except that (i,j) itself restarts as an inverse entry (0 + m * 1)
      vcur = (part == ps) ? m : fma(m, pc, vcur);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k == jb) { asm volatile("" ::: "memory"); v[k] = vcur; }
    }
  }
  __syncthreads();
  if (fail) {
    if (tid == 0 && info) { if (info[b] == 0) info[b] = info_base + fail; }
    const float qnan = __builtin_nanf("");
    for (int e = tid; e < n * n; e += NT) {
      const int i = e / n, j = e % n;
      L[(int64_t)i * ldl + j] = qnan;
      if (T) T[(int64_t)i * ldt + j] = qnan;
    }
    if (logdet && tid == 0) logdet[b] = qnan;
    return;
  }
  float ld_acc = 0.f;
  if (mine) {
    const F si = sqrt(sd[r]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e = k * P + part;
      if (e < r) {
        L[(int64_t)r * ldl + e] = (float)((F)Lp[pk(r, e)] / sqrt(sd[e]));
        if (T) T[(int64_t)r * ldt + e] = (float)(v[k] / si);
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
  if (logdet) {
    ld_acc = wave_sum(ld_acc);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = ld_acc;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int w = 0; w < (NT + 63) / 64; ++w) tot += red[w];
      if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot;
    }
  }
}

// stand-alone launch of chol3_body (chol_small3.h): columns across waves, rows across lanes
template <int KC, int SETS, class R = double>
__global__ __launch_bounds__(256) void chol_inv_small3_kernel(const float* __restrict__ A, int lda, int64_t strideA,
                                                              float eps, float* __restrict__ L, int ldl,
                                                              int64_t strideL, float* __restrict__ T, int ldt,
                                                              int64_t strideT, float* __restrict__ logdet,
                                                              int32_t* __restrict__ info, int info_base, int n,
                                                              int logdet_accumulate) {
  __shared__ float stage[chol3_stage_floats<KC>()];
  chol3_body<KC, SETS, R>(blockIdx.x, A, lda, strideA, eps, L, ldl, strideL, T, ldt, strideT, logdet, info, info_base, n,
                          logdet_accumulate, stage);
}

static int launch_small(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T,
                        int ldt, int64_t sT, float* logdet, int32_t* info, int info_base, int nbatch, int n,
                        int ld_acc, hipStream_t st) {
  ProfScope prof("chol_inv_small", st);
  const int nt = (int)round_up((int64_t)n * kCholP, 64);
#define VARGP_CHOL_LAUNCH(K)                                                                                     \
  hipLaunchKernelGGL((chol_inv_small_kernel<double, K>), dim3(nbatch), dim3(nt), 0, st, A, lda, sA, eps, L, ldl, sL, \
                     T, ldt, sT, logdet, info, info_base, n, ld_acc)
  // VARGP_CHOL_F32_ALONE=1 (tuning / measurement only): the stand-alone launch in the fp32 arithmetic of chol_small3.h
  static const int f32_alone = [] { const char* e = getenv("VARGP_CHOL_F32_ALONE"); return e ? atoi(e) : 0; }();
#define VARGP_CHOL3_LAUNCH(KC, SETS)                                                                                          \
  do {                                                                                                                          \
    if (f32_alone)                                                                                                              \
      hipLaunchKernelGGL((chol_inv_small3_kernel<KC, SETS, float>), dim3(nbatch), dim3(256), 0, st, A, lda, sA, eps, L, ldl, sL, T, \
                         ldt, sT, logdet, info, info_base, n, ld_acc);                                                            \
    else                                                                                                                        \
      hipLaunchKernelGGL((chol_inv_small3_kernel<KC, SETS, double>), dim3(nbatch), dim3(256), 0, st, A, lda, sA, eps, L, ldl, sL, T, \
                         ldt, sT, logdet, info, info_base, n, ld_acc);                                                            \
  } while (0)
  // measured (batch 30, MI355X): rows-across-threads kernel 10 / 22 us at n = 20 / 40; columns-across-waves kernel
  // 48 / 65 us at n = 64 / 100
  if (n <= 20) VARGP_CHOL_LAUNCH(4);
  else if (n <= 40) VARGP_CHOL_LAUNCH(8);
  else if (n <= 50) VARGP_CHOL_LAUNCH(10);
  else if (n <= 64) VARGP_CHOL3_LAUNCH(16, 1);
  else VARGP_CHOL3_LAUNCH(25, 2);   // n <= 100
#undef VARGP_CHOL_LAUNCH
#undef VARGP_CHOL3_LAUNCH
  return check_launch("chol_inv_small");
}

// Prologue of the blocked path, by nb x nb block (nb = panel width): blocks above the diagonal, L[b] = 0, T[b] = 0 -- the
// products that read the factors clip K per TILE, so the zeros next to the diagonal blocks must be real zeros.  Everything of
// L and T on / below the block diagonal is written later (diagonal blocks by the pivot-chain kernel incl. their zeros, the
// rest by the panel products).  There is no working COPY of A any more (round 5: it was a pass over the whole matrix, 128 us of
// the 2.3 ms at n = 2048 x 10): block column 0 is read from A itself, and the first update of every other region of the
// trailing matrix reads A and writes the workspace W (D = A, C = W); the jitter is added by the pivot-chain kernels.
__global__ void chol_zero_upper_kernel(float* __restrict__ L, float* __restrict__ T, int n, int nb) {
  const int64_t b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = e / n, j = e % n;
  if (j / nb > i / nb) {
    const int64_t o = b * n * n + e;
    L[o] = 0.f;
    if (T) T[o] = 0.f;
  }
}

// out = tril(gL) - tril(G2)   (either input may be null)
__global__ void tril_combine_kernel(const float* __restrict__ gL, const float* __restrict__ G2,
                                    float* __restrict__ out, int n, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e % ((int64_t)n * n);
  const int i = r / n, j = r % n;
  float v = 0.f;
  if (j <= i) v = (gL ? gL[e] : 0.f) - (G2 ? G2[e] : 0.f);
  out[e] = v;
}

// out_ij = half * P[max(i,j)][min(i,j)]   ( half = 0.5: (Phi(P) + Phi(P)^T) / 2 )
__global__ void phi_sym_kernel(const float* __restrict__ Pm, float* __restrict__ out, int n, int64_t total, float half) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t b = e / ((int64_t)n * n), r = e % ((int64_t)n * n);
  const int i = r / n, j = r % n;
  const int hi = i > j ? i : j, lo = i > j ? j : i;
  out[e] = half * Pm[b * n * n + (int64_t)hi * n + lo];
}

// square batched GEMM helper on dense [nbatch, n, n] buffers (or sub-blocks with explicit ld)
static int sq_gemm(const float* A, int lda, int64_t sA, int tA, int triA, const float* B, int ldb, int64_t sB, int tB,
                   int triB, float* C, int ldc, int64_t sC, const float* D, float alpha, float beta, int M, int N,
                   int K, int triC, int nbatch, hipStream_t st) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.D = D;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldc;
  p.nb1 = 1; p.nb2 = 1;
  p.sA[0] = sA; p.sB[0] = sB; p.sC[0] = sC; p.sD[0] = sC;
  p.alpha = alpha; p.beta = D ? beta : 0.f;
  p.triA = triA; p.triB = triB; p.triC = triC;
  return launch_gemm(p, tA, tB, nbatch, false, st);
}

}  // namespace vargp

using namespace vargp;

extern "C" size_t vargp_chol_workspace_bytes(int nbatch, int n, int backward) {
  const size_t nn = (size_t)nbatch * n * n * sizeof(float);
  if (backward) return 2 * nn + 256;
  if (n <= kSmallMax) return 256;
  const int wide = n < 4 * kNbMax ? n : 4 * kNbMax;                                       // widest (outer) panel any setting uses
  return nn + 2 * (size_t)nbatch * n * wide * sizeof(float) + 256;                        // W + two scratch strips (n x widest panel)
}

extern "C" int vargp_chol_inv_fwd(const float* A, float eps, float* L, float* T, float* logdet, int32_t* info,
                                  int nbatch, int n, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  return vargp::chol_inv_fwd_impl(A, eps, L, T, logdet, info, nbatch, n, ws, ws_bytes, true, as_stream(stream));
}

// zero_info = false: the caller has already cleared the status words (the fused ELBO program does it in its prologue).
// co[0 .. nco) / co_nbatch / co_done: RBF kernel-matrix GEMMs of the caller that do not depend on the factorisation (e.g.
// row slices of K_uf).  Diagonal block k shares ONE launch with co[k] if it qualifies (50 < width <= 100, T wanted, no
// logdet): its pivot chain keeps nbatch of the 256 CUs busy for ~50 us, the GEMM runs on the others.  *co_done = number of
// GEMMs consumed (always a prefix co[0 .. *co_done)); the caller launches the rest itself.
int vargp::chol_inv_fwd_impl(const float* A, float eps, float* L, float* T, float* logdet, int32_t* info, int nbatch,
                             int n, void* ws, size_t ws_bytes, bool zero_info, hipStream_t st, const GemmParams* co,
                             int co_nbatch, int* co_done, int nco, bool chain_f32) {
  VARGP_REQUIRE(A && L, "chol_inv_fwd: null pointer");
  VARGP_REQUIRE(nbatch > 0 && n > 0, "chol_inv_fwd: bad dims");
  const int64_t nn = (int64_t)n * n;
  int ndone = 0;
  if (co_done) *co_done = 0;
  if (!co) nco = 0;
  if (info && zero_info) zero_async(info, sizeof(int32_t) * nbatch, st);
  if (n <= kSmallMax) {
    if (nco > 0 && T && !logdet && info && chol_rbf_gemm_applicable(n, co[0]) && co_gemm_is_comparable(co[0], co_nbatch)) {
      if (co_done) *co_done = 1;
      return launch_chol_rbf_gemm_ld(A, n, nn, eps, L, n, nn, T, n, nn, info, nbatch, n, co[0], co_nbatch, st, chain_f32);
    }
    return launch_small(A, n, nn, eps, L, n, nn, T, n, nn, logdet, info, 0, nbatch, n, 0, st);
  }

  VARGP_REQUIRE(ws && ws_bytes >= vargp_chol_workspace_bytes(nbatch, n, 0), "chol_inv_fwd: workspace too small");
  float* W = reinterpret_cast<float*>(ws);
  float* tmp = W + (int64_t)nbatch * nn;
  const int kNbSmall = panel_width();
  const int NB2 = outer_panel_width(n, kNbSmall);       // == kNbSmall: one level
  const int64_t stmp = (int64_t)n * NB2;
  float* Tout = T;
  // zeros above the block diagonal of L and T: a third role of the FIRST pivot-chain launch (which leaves most of the chip idle)
  // when that launch has roles; else the plain kernel
  ZeroJobs zj{};
  zj.j[0] = ZeroJob{L, (int64_t)nbatch * n, n, n, n, kNbSmall};
  if (Tout) zj.j[1] = ZeroJob{Tout, (int64_t)nbatch * n, n, n, n, kNbSmall};
  static const int zrole_env = [] { const char* e = getenv("VARGP_CHOL_ZERO_ROLE"); return e ? atoi(e) : 1; }();   // tuning aid
  const int kb0 = kNbSmall < n ? kNbSmall : n;
  const bool co0 = nco > 0 && Tout && !logdet && info && chol_rbf_gemm_applicable(kb0, co[0]) && co_gemm_is_comparable(co[0], co_nbatch);
  const GemmParams none{};
  const bool zero_in_chain = zrole_env && Tout && info && !logdet && (co0 || (kb0 > 50 && kb0 <= 100));
  if (!zero_in_chain)
    hipLaunchKernelGGL(chol_zero_upper_kernel, dim3(cdiv(nn, 256), nbatch), dim3(256), 0, st, L, Tout, n, kNbSmall);
  int rc = VARGP_OK;
  // batched product on sub-blocks of the [nbatch, n, n] buffers (ld n, batch stride nn) or of tmp (ld ldtmp, stride stmp)
  auto mk = [&](const float* A_, int lda, const float* B_, int ldb, float* C_, int ldc, const float* D_, float alpha, float beta,
                int M_, int N_, int K_, int triA, int triB, int triC) {
    GemmParams p{};
    p.A = A_; p.B = B_; p.C = C_; p.D = D_;
    p.M = M_; p.N = N_; p.K = K_; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldc;
    p.nb1 = 1; p.nb2 = 1;
    p.sA[0] = (A_ == tmp) ? stmp : nn; p.sB[0] = (B_ == tmp) ? stmp : nn;
    p.sC[0] = (C_ == tmp) ? stmp : nn; p.sD[0] = p.sC[0];
    p.alpha = alpha; p.beta = D_ ? beta : 0.f;
    p.triA = triA; p.triB = triB; p.triC = triC;
    return p;
  };
  auto wgs = [&](const GemmParams& p) { return (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * nbatch; };
  static const int trow_pair = [] { const char* e = getenv("VARGP_CHOL_TROW_PAIR"); return e ? atoi(e) : 1; }();   // tuning aid
  // Right-looking, two levels: outer panels [K0, K1) of width NB2 (200 from n = 1200 on; NB2 == nb: the plain one-level
  // algorithm), inside them panels [k0, k1) of the register kernel's width nb.  After diagonal block k:
  //   (a1) the column of L below it, to the BOTTOM of the matrix:  L[k1:n, k] = W[k1:n, k] T_kk^T;
  //   (a2) the rest of the OUTER PANEL's columns only:            W[k1:n, k1:K1] -= L[k1:n, k] L[k1:K1, k]^T;
  //   (b)  block row k of T = L^-1 inside the outer block:        T[k, K0:k0] = -T_kk (L[k, K0:k0] T[K0:k0, K0:k0]);
  // and after the last block of an outer panel
  //   (A2) the trailing matrix, ONCE per outer panel with K = NB2: W[K1:n, K1:n] -= L[K1:n, K] L[K1:n, K]^T  (lower tiles);
  //   (B)  block row K of T to the left of the outer block:        T[K, 0:K0] = -T_KK (L[K, 0:K0] T[0:K0, 0:K0]).
  // Every L entry goes through the same arithmetic as in the one-level algorithm (panel solves against the fp64-accurate
  // 100 x 100 inverse blocks); what changes is that the trailing matrix -- all the remaining n^2 entries, read and written --
  // and the growing left part of T are passed over half as often, with K = 200 instead of 100 (which pads to 128 in 64-deep slabs).
  // Independent products share launches when they are mid-size: (a1 || b1), (a2 || b2), (A2 || B1).
  auto run_pair = [&](const GemmParams* x, int xA, int xB, const GemmParams* y, int yA, int yB, const char* tag) -> int {
    if (x && y && trow_pair && wgs(*x) + wgs(*y) <= 4096) return launch_gemm_pair2(*x, xA, xB, nbatch, *y, yA, yB, nbatch, st, tag);
    if (x) { const int r = launch_gemm(*x, xA, xB, nbatch, false, st, tag); if (r) return r; }
    if (y) { const int r = launch_gemm(*y, yA, yB, nbatch, false, st, tag); if (r) return r; }
    return VARGP_OK;
  };
  const int NBo = Tout ? NB2 : kNbSmall;
  // Look-ahead for T = L^-1 (stand-alone factorisation: no co-running GEMM of the caller's).  The block rows of T to the LEFT of
  // an outer block -- B1(K) = L[K, 0:K0] T[0:K0, 0:K0] into scratch, then B2(K) = -T_KK B1(K) -- feed nothing of the factorisation
  // of L, only each other in order (B1(K) reads the rows B2(K - 1) wrote; one scratch buffer).  They wait in a FIFO and every
  // pivot-chain launch (nbatch workgroups on 256 CUs for ~40 us) takes the next one along as its second role
  // (chol_nn_gemm_kernel); what is left at the end is launched plainly.  n = 2048 x 10: 2.56 -> 2.31 ms.  VARGP_CHOL_LOOKAHEAD=0: off.
  static const int la_env = [] { const char* e = getenv("VARGP_CHOL_LOOKAHEAD"); return e ? atoi(e) : 1; }();   // tuning aid
  const bool la = la_env && Tout && !logdet && info && nco == 0 && n >= 800;     // (n = 600: 2 % slower with it; 1000: 3 % faster; 1400: 8 %)
  std::vector<GemmParams> pend;
  size_t pend_i = 0;
  int b1_early = -1;      // outer panel (its K0) whose B1 already ran next to the first panel's trailing update
  for (int K0 = 0; K0 < n; K0 += NBo) {
    const int K1 = (K0 + NBo < n) ? K0 + NBo : n;
    if (la && K0 > 0) {     // B1 of this outer panel: everything it reads exists (behind the queued B2 of the previous panel)
      GemmParams B1q = mk(L + (int64_t)K0 * n, n, Tout, n, tmp + (int64_t)nbatch * stmp, K0, nullptr, 1.f, 0.f, K1 - K0, K0, K0, 0, 1, 0);
      B1q.sC[0] = stmp; B1q.sD[0] = stmp;
      pend.push_back(B1q);
    }
    for (int k0 = K0; k0 < K1; k0 += kNbSmall) {
      const int k1 = (k0 + kNbSmall < K1) ? k0 + kNbSmall : K1, kb = k1 - k0;
      const int rem = n - k1;                      // rows below the block, to the bottom
      const int ncol = K1 - k1;                    // columns of the outer panel still to come
      const int64_t dkk = (int64_t)k0 * n + k0;
      // diagonal block: L_kk, T_kk.  The panel solve needs T_kk even when the caller skips T: park it in tmp's head then.
      float* Tkk = Tout ? Tout + dkk : tmp;
      const int ldt = Tout ? n : kb;
      const int64_t sT = Tout ? nn : stmp;
      const int kpanel = k0 / kNbSmall;
      // block column 0 has seen no update: it is read from A itself; the first update of a region of the trailing matrix
      // reads A and writes W (`first`: nothing has touched W[k1:, k1:] yet)
      const float* Wk = k0 == 0 ? A : W;
      const bool first = k0 == 0;
      if (kpanel < nco && Tout && !logdet && info && chol_rbf_gemm_applicable(kb, co[kpanel]) &&
          co_gemm_is_comparable(co[kpanel], co_nbatch)) {
        // (chol3_body reports a failing pivot as info_base + j + 1; the merged kernel has no info_base: only the first
        // panel's index is exact, later panels report the index within the panel -- non-zero is what callers test)
        rc = launch_chol_rbf_gemm_ld(Wk + dkk, n, nn, eps, L + dkk, n, nn, Tkk, ldt, sT, info, nbatch, kb, co[kpanel], co_nbatch, st, chain_f32,
                                     (k0 == 0 && zero_in_chain) ? &zj : nullptr);
        ++ndone;
      } else {
        if (k0 == 0 && zero_in_chain) {       // first block, nothing to take along but the zero-fills
          rc = launch_chol_nn_gemm(Wk + dkk, n, nn, eps, L + dkk, n, nn, Tkk, ldt, sT, info, k0, nbatch, kb, none, 0, st, &zj);
        } else if (la && pend_i < pend.size() && chol_nn_gemm_applicable(kb, pend[pend_i])) {
          rc = launch_chol_nn_gemm(Wk + dkk, n, nn, eps, L + dkk, n, nn, Tkk, ldt, sT, info, k0, nbatch, kb, pend[pend_i], nbatch, st);
          ++pend_i;
        } else {
          rc = launch_small(Wk + dkk, n, nn, eps, L + dkk, n, nn, Tkk, ldt, sT, logdet, info, k0, nbatch, kb, k0 > 0, st);
        }
      }
      if (rc) return rc;
      float* W22 = W + (int64_t)k1 * n + k1;
      const float* D22 = first ? A + (int64_t)k1 * n + k1 : W22;
      if (!Tout) {
        // no T wanted (one level): only (a) runs, T_kk parked in tmp with its own strides
        if (rem > 0) {
          rc = sq_gemm(Wk + (int64_t)k1 * n + k0, n, nn, 0, 0, tmp, kb, stmp, 1, 2, L + (int64_t)k1 * n + k0, n, nn,
                       nullptr, 1.f, 0.f, rem, kb, kb, 0, nbatch, st);
          if (rc) return rc;
          rc = sq_gemm(L + (int64_t)k1 * n + k0, n, nn, 0, 0, L + (int64_t)k1 * n + k0, n, nn, 1, 0, W22, n, nn, D22,
                       -1.f, 1.f, rem, rem, kb, 2, nbatch, st);
          if (rc) return rc;
        }
        continue;
      }
      const int kl = k0 - K0;                      // columns of the outer block to the left of this one
      const int ldtmp = kl > 0 ? kl : 1;
      const GemmParams a1 = mk(Wk + (int64_t)k1 * n + k0, n, Tkk, n, L + (int64_t)k1 * n + k0, n, nullptr, 1.f, 0.f, rem, kb, kb, 0, 2, 0);
      const GemmParams b1 = mk(L + (int64_t)k0 * n + K0, n, Tout + (int64_t)K0 * n + K0, n, tmp, ldtmp, nullptr, 1.f, 0.f, kb, kl, kl, 0, 1, 0);
      const GemmParams a2 = mk(L + (int64_t)k1 * n + k0, n, L + (int64_t)k1 * n + k0, n, W22, n, D22, -1.f, 1.f, rem, ncol, kb, 0, 0, 2);
      const GemmParams b2 = mk(Tkk, n, tmp, ldtmp, Tout + (int64_t)k0 * n + K0, n, nullptr, -1.f, 0.f, kb, kl, kb, 1, 0, 0);
      // the outer panel's own pair (A2 || B2) follows its last block; B1, which only needs earlier outer panels, rides with
      // that block's products (its scratch is the second half of tmp: b1 / b2 of this block use the first)
      const bool last = k1 == K1;
      const int KB = K1 - K0, REM = n - K1;
      const int ldTMP = K0 > 0 ? K0 : 1;
      float* tmpB = tmp + (int64_t)nbatch * stmp;
      GemmParams B1 = mk(L + (int64_t)K0 * n, n, Tout, n, tmpB, ldTMP, nullptr, 1.f, 0.f, KB, K0, K0, 0, 1, 0);
      B1.sC[0] = stmp; B1.sD[0] = stmp;
      const bool doB = last && K0 > 0 && !la && b1_early != K0;     // (look-ahead: B1 went into the queue at the top of the panel)
      if (kl > 0) {
        rc = run_pair(rem > 0 ? &a1 : nullptr, 0, 1, &b1, 0, 0, "chol_panel");
        if (rc) return rc;
        rc = run_pair((rem > 0 && ncol > 0) ? &a2 : (doB ? &B1 : nullptr), (rem > 0 && ncol > 0) ? 0 : 0,
                      (rem > 0 && ncol > 0) ? 1 : 0, &b2, 0, 0, "chol_panel");
        if (rc) return rc;
        if (doB && rem > 0 && ncol > 0) {            // (cannot happen: the last block has ncol == 0)
          rc = launch_gemm(B1, 0, 0, nbatch, false, st, "chol_trow");
          if (rc) return rc;
        }
      } else {
        rc = run_pair(rem > 0 ? &a1 : nullptr, 0, 1, doB ? &B1 : nullptr, 0, 0, "chol_panel");
        if (rc) return rc;
        if (rem > 0 && ncol > 0) {
          rc = launch_gemm(a2, 0, 1, nbatch, false, st, "chol_panel");
          if (rc) return rc;
        }
      }
      if (last) {
        float* TKK = Tout + (int64_t)K0 * n + K0;
        float* W22o = W + (int64_t)K1 * n + K1;
        // (the first outer panel's trailing update is the first touch of W[K1:, K1:])
        const float* D22o = K0 == 0 ? A + (int64_t)K1 * n + K1 : W22o;
        const GemmParams A2 = mk(L + (int64_t)K1 * n + K0, n, L + (int64_t)K1 * n + K0, n, W22o, n, D22o, -1.f, 1.f, REM, REM, KB, 0, 0, 2);
        GemmParams B2 = mk(TKK, n, tmpB, ldTMP, Tout + (int64_t)K0 * n, n, nullptr, -1.f, 0.f, KB, K0, KB, 1, 0, 0);
        B2.sB[0] = stmp;
        if (la && K0 > 0) pend.push_back(B2);
        // First outer panel: there is no B2 to keep A2 company, but B1 of the NEXT panel -- L[K1:K1n, 0:K1] T[0:K1, 0:K1] -- reads
        // only what exists now (the panel's columns of L, its block of T): it takes the free seat, and the next panel's last block
        // finds it done (Split-MNIST t = 1, two panels: six launches of the factorisation become five)
        GemmParams B1n{};
        const bool early = K0 == 0 && !la && REM > 0;
        if (early) {
          const int K1n = (K1 + NBo < n) ? K1 + NBo : n;
          B1n = mk(L + (int64_t)K1 * n, n, Tout, n, tmpB, K1, nullptr, 1.f, 0.f, K1n - K1, K1, K1, 0, 1, 0);
          B1n.sC[0] = stmp; B1n.sD[0] = stmp;
          b1_early = K1;
        }
        rc = run_pair(REM > 0 ? &A2 : nullptr, 0, 1, early ? &B1n : ((K0 > 0 && !la) ? &B2 : nullptr), 0, 0, "chol_trailing");
        if (rc) return rc;
      }
    }
  }
  for (; pend_i < pend.size(); ++pend_i) {      // block rows of T no pivot chain was left to take along
    rc = launch_gemm(pend[pend_i], 0, 0, nbatch, false, st, "chol_trow");
    if (rc) return rc;
  }
  if (co_done) *co_done = ndone;
  return check_launch("chol_inv_fwd");
}

extern "C" int vargp_chol_inv_bwd(const float* L, const float* T, const float* gL, const float* gT, float* gA,
                                  int nbatch, int n, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  return vargp::chol_inv_bwd_impl(L, T, gL, gT, gA, nbatch, n, ws, ws_bytes, false, as_stream(stream), false);
}

// With T = L^-1 (dT = -T dL T) the total gradient on L is  gL_tot = tril(gL) - tril(Y),  Y = T^T gT T^T, and
//   gA = T^T S T,   S = (Phi(P) + Phi(P)^T) / 2,   P = L^T gL_tot,   Phi = lower triangle with halved diagonal.
// Only the lower triangle of P is used, and there  L^T tril(Y) = L^T Y - L^T triu_strict(Y)  agrees with
// L^T Y = (T L)^T gT T^T = gT T^T  (upper x strictly-upper is strictly upper).  So
//   P = L^T tril(gL) - gT T^T :   four GEMMs in all (gT T^T, L^T gL, T^T S, (.) T) instead of five.
// gl_lower: gL is already lower-triangular with stored zeros above the diagonal (skips the masking pass).
// First product of the backward, w1 = gT T^T, optionally sharing its launch with an unrelated product `other` of the
// caller (transposition flags oA / oB, batch onb) that is ready at the same time.
int vargp::chol_inv_bwd_first(const float* T, const float* gT, int nbatch, int n, void* ws, size_t ws_bytes,
                              const GemmParams* other, int oA, int oB, int onb, hipStream_t st) {
  VARGP_REQUIRE(T && gT && ws && ws_bytes >= vargp_chol_workspace_bytes(nbatch, n, 1), "chol_inv_bwd_first: bad arguments");
  const int64_t nn = (int64_t)n * n;
  float* w1 = reinterpret_cast<float*>(ws);
  GemmParams p{};
  p.A = gT; p.B = T; p.C = w1;
  p.M = n; p.N = n; p.K = n; p.lda = n; p.ldb = n; p.ldc = n; p.ldd = n;
  p.nb1 = 1; p.nb2 = 1;
  p.sA[0] = nn; p.sB[0] = nn; p.sC[0] = nn; p.sD[0] = nn;
  p.alpha = 1.f;
  p.triB = 2;
  if (other) return launch_gemm_pair2(p, 0, 1, nbatch, *other, oA, oB, onb, st, "chol_bwd1_pair");
  return launch_gemm(p, 0, 1, nbatch, false, st);
}

int vargp::chol_inv_bwd_impl(const float* L, const float* T, const float* gL, const float* gT, float* gA, int nbatch,
                             int n, void* ws, size_t ws_bytes, bool gl_lower, hipStream_t st, bool first_done) {
  VARGP_REQUIRE(L && T && gA && ws, "chol_inv_bwd: null pointer");
  VARGP_REQUIRE(ws_bytes >= vargp_chol_workspace_bytes(nbatch, n, 1), "chol_inv_bwd: workspace too small");
  const int64_t nn = (int64_t)n * n, total = nn * nbatch;
  float* w1 = reinterpret_cast<float*>(ws);
  float* w2 = w1 + total;
  int rc;
  if (gT && !first_done) {   // w1 = gT T^T  (first_done: the caller ran chol_inv_bwd_first already)
    rc = chol_inv_bwd_first(T, gT, nbatch, n, ws, ws_bytes, nullptr, 0, 0, 0, st);
    if (rc) return rc;
  }
  float* Sm = nullptr;
  if (gL) {
    const float* gLl = gL;
    if (!gl_lower) {
      hipLaunchKernelGGL(tril_combine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, gL, (const float*)nullptr, gA, n,
                         total);
      gLl = gA;
    }
    // S = (Phi(P) + Phi(P)^T) / 2 with P = L^T tril(gL) - w1, written symmetrically by the GEMM's epilogue from the
    // lower triangle of 0.5 P (tiles above the diagonal are not computed)
    GemmParams p{};
    p.A = L; p.B = gLl; p.C = w2; p.D = gT ? w1 : nullptr;
    p.M = n; p.N = n; p.K = n; p.lda = n; p.ldb = n; p.ldc = n; p.ldd = n;
    p.nb1 = 1; p.nb2 = 1;
    p.sA[0] = nn; p.sB[0] = nn; p.sC[0] = nn; p.sD[0] = nn;
    p.alpha = 0.5f; p.beta = gT ? -0.5f : 0.f;
    p.triA = 2; p.triB = 1; p.triC = 2; p.symout = 1;
    rc = launch_gemm(p, 1, 0, nbatch, false, st);
    if (rc) return rc;
    Sm = w2;
  } else {
    VARGP_REQUIRE(gT, "chol_inv_bwd: neither gL nor gT given");
    // only gT: S = -(Phi(w1) + Phi(w1)^T) / 2
    Sm = w2;
    hipLaunchKernelGGL(phi_sym_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w1, Sm, n, total, -0.5f);
  }
  float* tmp = (Sm == w1) ? w2 : w1;
  rc = sq_gemm(T, n, nn, 1, 2, Sm, n, nn, 0, 0, tmp, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);
  if (rc) return rc;
  rc = sq_gemm(tmp, n, nn, 0, 0, T, n, nn, 0, 1, gA, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);
  if (rc) return rc;
  return check_launch("chol_inv_bwd");
}

// Triangular solve against a factor whose inverse came with it (SURVEY §8b lists trsm_lower as an op of its own; here
// it is, by design, a GEMM with T = L^-1):  X = L^-1 B = T B.
extern "C" int vargp_trsm_lower_fwd(const float* T, const float* B, float* X, int nbatch, int n, int nrhs,
                                    vargp_stream_t stream) {
  VARGP_REQUIRE(T && B && X && nbatch > 0 && n > 0 && nrhs > 0, "trsm_lower_fwd: bad arguments");
  return sq_gemm(T, n, (int64_t)n * n, 0, 1, B, nrhs, (int64_t)n * nrhs, 0, 0, X, nrhs, (int64_t)n * nrhs, nullptr, 1.f, 0.f, n,
                 nrhs, n, 0, nbatch, as_stream(stream));
}

// Adjoint of X = L^-1 B (torch.triangular_solve backward, gp_utils.py:89-134 call sites):  gB = L^-T gX = T^T gX,
// gL = -tril(gB X^T)  (either output may be NULL)
extern "C" int vargp_trsm_lower_bwd(const float* T, const float* X, const float* gX, float* gB, float* gL, int nbatch, int n,
                                    int nrhs, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  VARGP_REQUIRE(T && X && gX && nbatch > 0 && n > 0 && nrhs > 0, "trsm_lower_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  float* gBp = gB;
  if (!gBp) {
    VARGP_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)nbatch * n * nrhs, "trsm_lower_bwd: workspace too small");
    gBp = reinterpret_cast<float*>(ws);
  }
  int rc = sq_gemm(T, n, (int64_t)n * n, 1, 2, gX, nrhs, (int64_t)n * nrhs, 0, 0, gBp, nrhs, (int64_t)n * nrhs, nullptr, 1.f,
                   0.f, n, nrhs, n, 0, nbatch, st);
  if (rc || !gL) return rc;
  return sq_gemm(gBp, nrhs, (int64_t)n * nrhs, 0, 0, X, nrhs, (int64_t)n * nrhs, 1, 0, gL, n, (int64_t)n * n, nullptr, -1.f, 0.f,
                 n, n, nrhs, 1, nbatch, st);
}
