// Batched Cholesky (+ jitter) with explicit inverse factor T = L^-1 and log-determinant
// (reference: gp_utils.cholesky, var_gp/gp_utils.py:5-11, plus every triangular_solve against it).
//
// n <= 128: one 256-thread workgroup per matrix, matrix resident in LDS (odd row stride).
//   factorisation: right-looking with deferred column scaling (one barrier per column):
//       a_ik -= a_ij a_kj / d_j  (j < k <= i),  L_ij = a_ij / sqrt(d_j)
//   inverse: forward substitution, one lane group per column of T; groups never interact, so the
//   sweep needs no workgroup barrier.
// n > 128: blocked right-looking on 128-wide panels; the diagonal blocks use the LDS kernel, the
//   panel solves and trailing updates are MFMA GEMMs (gemm.hip).
// Backward (any n) is five GEMMs (see vargp_chol_inv_bwd).
#include "common.h"
#include <math.h>

namespace vargp {

constexpr int kNbSmall = 128;

__global__ __launch_bounds__(256) void chol_inv_small_kernel(const float* __restrict__ A, int lda, int64_t strideA,
                                                             float eps, float* __restrict__ L, int ldl,
                                                             int64_t strideL, float* __restrict__ T, int ldt,
                                                             int64_t strideT, float* __restrict__ logdet,
                                                             int32_t* __restrict__ info, int info_base, int n,
                                                             int logdet_accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LD = n | 1;
  float* sa = smem;            // n x LD  working matrix / L
  float* sd = sa + n * LD;     // n       sqrt of pivots
  float* st = sd + ((n + 3) & ~3);  // n x LD  T (only if T != null)
  __shared__ float red[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  A += b * strideA;
  L += b * strideL;
  if (T) T += b * strideT;

  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    if (j <= i) sa[i * LD + j] = A[(int64_t)i * lda + j] + (i == j ? eps : 0.f);
  }

  int fail = 0;
  for (int j = 0; j < n; ++j) {
    __syncthreads();
    const float d = sa[j * LD + j];
    if (!(d > 0.f)) { fail = j + 1; break; }  // uniform: every thread reads the same pivot
    const float inv = 1.f / d;
    for (int k = j + 1 + lane; k < n; k += 64) {
      const float ckj = sa[k * LD + j] * inv;
      for (int i = j + 1 + wave; i < n; i += 4) {
        if (i >= k) sa[i * LD + k] = fmaf(-sa[i * LD + j], ckj, sa[i * LD + k]);
      }
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
    if (logdet && tid == 0) logdet[b] = qnan;
    return;
  }
  for (int j = tid; j < n; j += 256) sd[j] = sqrtf(sa[j * LD + j]);
  __syncthreads();
  float ld_acc = 0.f;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    float v = 0.f;
    if (j < i) v = sa[i * LD + j] / sd[j];
    else if (j == i) { v = sd[j]; ld_acc += logf(v); }
    if (j <= i) sa[i * LD + j] = v;
    L[(int64_t)i * ldl + j] = v;
  }
  if (logdet) {
    const float tot = block_sum<256>(ld_acc, red);
    if (tid == 0) { if (logdet_accumulate) logdet[b] += tot; else logdet[b] = tot; }
  }
  if (!T) return;
  __syncthreads();

  // T = L^-1, column c handled by P adjacent lanes of one wave
  int np2 = 16;
  while (np2 < n) np2 <<= 1;
  const int P = 256 / np2;
  const int c = tid / P, part = tid % P;
  if (c < n) {
    for (int i = 0; i < n; ++i) {
      if (i < c) {
        if (part == 0) st[i * LD + c] = 0.f;
      } else {
        float acc = 0.f;
        for (int k = c + part; k < i; k += P) acc = fmaf(sa[i * LD + k], st[k * LD + c], acc);
        for (int off = P >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        const float t = ((i == c ? 1.f : 0.f) - acc) / sa[i * LD + i];
        if (part == 0) st[i * LD + c] = t;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    T[(int64_t)i * ldt + j] = (j <= i) ? st[i * LD + j] : 0.f;
  }
}

static size_t small_lds_bytes(int n, bool want_T) {
  const int LD = n | 1;
  return sizeof(float) * ((size_t)n * LD + ((n + 3) & ~3) + (want_T ? (size_t)n * LD : 0));
}

static int launch_small(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T,
                        int ldt, int64_t sT, float* logdet, int32_t* info, int info_base, int nbatch, int n,
                        int ld_acc, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chol_inv_small_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
      set_error("chol: cannot raise dynamic LDS limit");
      return VARGP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(chol_inv_small_kernel, dim3(nbatch), dim3(256), small_lds_bytes(n, T != nullptr), st, A, lda, sA,
                     eps, L, ldl, sL, T, ldt, sT, logdet, info, info_base, n, ld_acc);
  return check_launch("chol_inv_small");
}

// W[b] = lower(A[b]) + eps I  (dense copy incl. upper part, which is never read)
__global__ void copy_jitter_kernel(const float* __restrict__ A, float* __restrict__ W, int n, float eps) {
  const int64_t b = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = e / n, j = e % n;
  W[b * n * n + e] = A[b * n * n + e] + (i == j ? eps : 0.f);
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

// out_ij = 0.5 * P[max(i,j)][min(i,j)]   ( = (Phi(P) + Phi(P)^T) / 2 )
__global__ void phi_sym_kernel(const float* __restrict__ Pm, float* __restrict__ out, int n, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t b = e / ((int64_t)n * n), r = e % ((int64_t)n * n);
  const int i = r / n, j = r % n;
  const int hi = i > j ? i : j, lo = i > j ? j : i;
  out[e] = 0.5f * Pm[b * n * n + (int64_t)hi * n + lo];
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
  if (n <= kNbSmall) return 256;
  return nn + (size_t)nbatch * n * kNbSmall * sizeof(float) + 256;
}

extern "C" int vargp_chol_inv_fwd(const float* A, float eps, float* L, float* T, float* logdet, int32_t* info,
                                  int nbatch, int n, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  VARGP_REQUIRE(A && L, "chol_inv_fwd: null pointer");
  VARGP_REQUIRE(nbatch > 0 && n > 0, "chol_inv_fwd: bad dims");
  hipStream_t st = as_stream(stream);
  const int64_t nn = (int64_t)n * n;
  if (info) (void)hipMemsetAsync(info, 0, sizeof(int32_t) * nbatch, st);
  if (n <= kNbSmall)
    return launch_small(A, n, nn, eps, L, n, nn, T, n, nn, logdet, info, 0, nbatch, n, 0, st);

  VARGP_REQUIRE(ws && ws_bytes >= vargp_chol_workspace_bytes(nbatch, n, 0), "chol_inv_fwd: workspace too small");
  float* W = reinterpret_cast<float*>(ws);
  float* tmp = W + (int64_t)nbatch * nn;
  const int64_t stmp = (int64_t)n * kNbSmall;
  hipLaunchKernelGGL(copy_jitter_kernel, dim3(cdiv(nn, 256), nbatch), dim3(256), 0, st, A, W, n, eps);
  (void)hipMemsetAsync(L, 0, sizeof(float) * nbatch * nn, st);
  // the blocked inverse needs the factor even if the caller does not want T: use tmp-free path
  float* Tout = T;
  if (Tout) (void)hipMemsetAsync(Tout, 0, sizeof(float) * nbatch * nn, st);
  int rc;
  for (int k0 = 0; k0 < n; k0 += kNbSmall) {
    const int kb = (n - k0 < kNbSmall) ? n - k0 : kNbSmall;
    const int k1 = k0 + kb, rem = n - k1;
    const int64_t dkk = (int64_t)k0 * n + k0;
    // diagonal block: L_kk, T_kk.  The panel solve needs T_kk even when the caller skips T: park
    // it in tmp's head in that case.
    float* Tkk = Tout ? Tout + dkk : tmp;
    const int ldt = Tout ? n : kb;
    const int64_t sT = Tout ? nn : stmp;
    rc = launch_small(W + dkk, n, nn, 0.f, L + dkk, n, nn, Tkk, ldt, sT, logdet, info, k0, nbatch, kb, k0 > 0, st);
    if (rc) return rc;
    if (rem > 0) {
      if (!Tout) {
        // T_kk sits in tmp: panel result must go elsewhere -> write directly into L (inputs are W, tmp)
      }
      // L_ik = W_ik T_kk^T   (rem x kb)
      rc = sq_gemm(W + (int64_t)k1 * n + k0, n, nn, 0, 0, Tkk, ldt, sT, 1, 2, L + (int64_t)k1 * n + k0, n, nn,
                   nullptr, 1.f, 0.f, rem, kb, kb, 0, nbatch, st);
      if (rc) return rc;
      // W_22 -= L_panel L_panel^T  (lower tiles only, in place)
      float* W22 = W + (int64_t)k1 * n + k1;
      rc = sq_gemm(L + (int64_t)k1 * n + k0, n, nn, 0, 0, L + (int64_t)k1 * n + k0, n, nn, 1, 0, W22, n, nn, W22,
                   -1.f, 1.f, rem, rem, kb, 2, nbatch, st);
      if (rc) return rc;
    }
  }
  if (Tout) {
    // off-diagonal blocks of T, last block column first:  T[j1:, j] = -T[j1:, j1:] (L[j1:, j] T_jj)
    const int nblk = cdiv(n, kNbSmall);
    for (int jb = nblk - 2; jb >= 0; --jb) {
      const int j0 = jb * kNbSmall, j1 = j0 + kNbSmall, rem = n - j1;
      rc = sq_gemm(L + (int64_t)j1 * n + j0, n, nn, 0, 0, Tout + (int64_t)j0 * n + j0, n, nn, 0, 1, tmp, kNbSmall,
                   stmp, nullptr, 1.f, 0.f, rem, kNbSmall, kNbSmall, 0, nbatch, st);
      if (rc) return rc;
      rc = sq_gemm(Tout + (int64_t)j1 * n + j1, n, nn, 0, 1, tmp, kNbSmall, stmp, 0, 0,
                   Tout + (int64_t)j1 * n + j0, n, nn, nullptr, -1.f, 0.f, rem, kNbSmall, rem, 0, nbatch, st);
      if (rc) return rc;
    }
  }
  return check_launch("chol_inv_fwd");
}

extern "C" int vargp_chol_inv_bwd(const float* L, const float* T, const float* gL, const float* gT, float* gA,
                                  int nbatch, int n, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  VARGP_REQUIRE(L && T && gA && ws, "chol_inv_bwd: null pointer");
  VARGP_REQUIRE(ws_bytes >= vargp_chol_workspace_bytes(nbatch, n, 1), "chol_inv_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const int64_t nn = (int64_t)n * n, total = nn * nbatch;
  float* w1 = reinterpret_cast<float*>(ws);
  float* w2 = w1 + total;
  int rc;
  const float* G2 = nullptr;
  if (gT) {
    // d<gT, T> = -<T^T gT T^T, dL>
    rc = sq_gemm(T, n, nn, 1, 2, gT, n, nn, 0, 0, w1, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);  // T^T gT
    if (rc) return rc;
    rc = sq_gemm(w1, n, nn, 0, 0, T, n, nn, 1, 2, w2, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);  // . T^T
    if (rc) return rc;
    G2 = w2;
  }
  hipLaunchKernelGGL(tril_combine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, gL, G2, gA, n, total);
  // P = L^T gL_tot ; Psym = (Phi(P) + Phi(P)^T)/2 ; gA = T^T Psym T
  rc = sq_gemm(L, n, nn, 1, 2, gA, n, nn, 0, 1, w1, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);
  if (rc) return rc;
  hipLaunchKernelGGL(phi_sym_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w1, w2, n, total);
  rc = sq_gemm(T, n, nn, 1, 2, w2, n, nn, 0, 0, w1, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);
  if (rc) return rc;
  rc = sq_gemm(w1, n, nn, 0, 0, T, n, nn, 0, 1, gA, n, nn, nullptr, 1.f, 0.f, n, n, n, 0, nbatch, st);
  if (rc) return rc;
  return check_launch("chol_inv_bwd");
}
