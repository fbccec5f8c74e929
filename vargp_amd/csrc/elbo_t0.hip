// Glue kernels of the fused task-0 ELBO path (vargp_amd/fused.py): packing the small right-hand sides that
// share the factor T = Lz^-1 into one GEMM operand, and the MVN-KL of q(u) = N(m, Lu Lu^T) against the prior
// p(u | theta) = N(0, Lz Lz^T) (reference: var_gp/vargp.py:156-190) computed straight from that GEMM's output.
//
//   Rsmall[c] = [ m_c | 0 0 0 | LS_c | Lu_c ]            (M x NR, NR = 4 + 2M), one per class
//   Q[s,c]    = T[s,c] . Rsmall[c] = [ a | 0 0 0 | G | G2 ],  a = Lz^-1 m, G = Lz^-1 L_S, G2 = Lz^-1 Lu
//   kl[s,c]   = sum log diag Lz - sum log diag Lu + 0.5 (|G2|_F^2 + |a|^2 - M);   kl_u = (1/S) sum_{s,c} kl[s,c]
#include "common.h"

namespace vargp {

__global__ void pack_rsmall_kernel(const float* __restrict__ m, const float* __restrict__ LS,
                                   const float* __restrict__ Lu, float* __restrict__ R, int M, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int NR = 4 + 2 * M;
  const int col = e % NR;
  const int64_t row = e / NR;            // c * M + i
  const int64_t c = row / M;
  const int i = row % M;
  float v = 0.f;
  if (col == 0) v = m[row];
  else if (col >= 4 && col < 4 + M) v = LS[(c * M + i) * M + (col - 4)];
  else if (col >= 4 + M) v = Lu[(c * M + i) * M + (col - 4 - M)];
  R[e] = v;
}

// grid (ceil(M / kKlRows), S*C): a block reduces kKlRows rows of one (s, c); kl_u accumulated with one atomic per
// block (pre-zeroed by the caller)
constexpr int kKlRows = 8;
__global__ __launch_bounds__(256) void kl_t0_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Lz,
                                                        const float* __restrict__ Lu, float* __restrict__ kl_u, int S,
                                                        int C, int M) {
  __shared__ float red[4];
  const int NR = 4 + 2 * M;
  const int64_t b = blockIdx.y;          // s * C + c
  const int c = b % C;
  const int i0 = blockIdx.x * kKlRows, i1 = min(M, i0 + kKlRows);
  const float* q = Q + b * M * NR;
  float acc = 0.f;
  for (int e = threadIdx.x; e < (i1 - i0) * M; e += 256) {
    const int i = i0 + e / M, j = e % M;
    if (j <= i) { const float v = q[(int64_t)i * NR + 4 + M + j]; acc = fmaf(v, v, acc); }
  }
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    const float a = q[(int64_t)i * NR];
    acc = fmaf(a, a, acc);
    acc += 2.f * (logf(Lz[(b * M + i) * M + i]) - logf(Lu[((int64_t)c * M + i) * M + i])) - 1.f;
  }
  const float t = block_sum<256>(acc, red);
  if (threadIdx.x == 0) atomicAdd(kl_u, 0.5f * t / (float)S);
}

// gQ[:, 0] = ga + g a / S ; gQ[:, 1..3] = 0 ; gQ[:, 4+M..] = g tril(G2) / S  (the G block is written by a GEMM);
// gLz = diag(g / (S Lz_ii)) ; gLu_diag[c, i] = -g / Lu_ii  (summed over s analytically)
__global__ __launch_bounds__(256) void kl_t0_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Lz,
                                                        const float* __restrict__ Lu, const float* __restrict__ ga,
                                                        const float* __restrict__ gkl, float* __restrict__ gQ,
                                                        float* __restrict__ gLz, float* __restrict__ gLu, int S, int C,
                                                        int M) {
  const int NR = 4 + 2 * M;
  const int64_t b = blockIdx.y;
  const int s = b / C, c = b % C;
  const int i0 = blockIdx.x * kKlRows, i1 = min(M, i0 + kKlRows);
  const float g = gkl[0] / (float)S;
  const float* q = Q + b * M * NR;
  float* gq = gQ + b * M * NR;
  for (int e0 = threadIdx.x; e0 < (i1 - i0) * M; e0 += 256) {
    const int i = i0 + e0 / M, j = e0 % M;
    const int e = i * M + j;
    gq[(int64_t)i * NR + 4 + M + j] = (j <= i) ? g * q[(int64_t)i * NR + 4 + M + j] : 0.f;
    gLz[b * M * M + e] = (i == j) ? g / Lz[b * M * M + e] : 0.f;
    if (s == 0) gLu[(int64_t)c * M * M + e] = (i == j) ? -gkl[0] / Lu[(int64_t)c * M * M + e] : 0.f;
  }
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    gq[(int64_t)i * NR] = ga[b * M + i] + g * q[(int64_t)i * NR];
    gq[(int64_t)i * NR + 1] = 0.f; gq[(int64_t)i * NR + 2] = 0.f; gq[(int64_t)i * NR + 3] = 0.f;
  }
}

// gtheta[s, D] += 2 gamma_s^2 sum_c gkd[s, c]   (kdiag = gamma^2 = exp(2 theta_D), var_gp/kernels.py:58-60)
__global__ void kdiag_bwd_kernel(const float* __restrict__ theta, const float* __restrict__ gkd,
                                 float* __restrict__ gtheta, int S, int C, int D) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc += gkd[s * C + c];
  gtheta[(int64_t)s * (D + 1) + D] += 2.f * expf(2.f * theta[(int64_t)s * (D + 1) + D]) * acc;
}

}  // namespace vargp

using namespace vargp;

extern "C" int vargp_pack_rsmall(const float* m, const float* LS, const float* Lu, float* R, int C, int M,
                                 vargp_stream_t stream) {
  VARGP_REQUIRE(m && LS && Lu && R && C > 0 && M > 0, "pack_rsmall: bad arguments");
  const int64_t total = (int64_t)C * M * (4 + 2 * M);
  hipLaunchKernelGGL(pack_rsmall_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), m, LS, Lu, R, M, total);
  return check_launch("pack_rsmall");
}
extern "C" int vargp_kl_t0_fwd(const float* Q, const float* Lz, const float* Lu, float* kl_u, int S, int C, int M,
                               vargp_stream_t stream) {
  VARGP_REQUIRE(Q && Lz && Lu && kl_u && S > 0 && C > 0 && M > 0, "kl_t0_fwd: bad arguments");
  zero_async(kl_u, sizeof(float), as_stream(stream));
  hipLaunchKernelGGL(kl_t0_fwd_kernel, dim3(cdiv(M, kKlRows), S * C), dim3(256), 0, as_stream(stream), Q, Lz, Lu, kl_u, S, C,
                     M);
  return check_launch("kl_t0_fwd");
}
extern "C" int vargp_kl_t0_bwd(const float* Q, const float* Lz, const float* Lu, const float* ga, const float* gkl,
                               float* gQ, float* gLz, float* gLu, int S, int C, int M, vargp_stream_t stream) {
  VARGP_REQUIRE(Q && Lz && Lu && ga && gkl && gQ && gLz && gLu, "kl_t0_bwd: null pointer");
  hipLaunchKernelGGL(kl_t0_bwd_kernel, dim3(cdiv(M, kKlRows), S * C), dim3(256), 0, as_stream(stream), Q, Lz, Lu, ga, gkl,
                     gQ, gLz, gLu, S, C, M);
  return check_launch("kl_t0_bwd");
}
extern "C" int vargp_kdiag_bwd(const float* theta, const float* gkd, float* gtheta, int S, int C, int D,
                               vargp_stream_t stream) {
  VARGP_REQUIRE(theta && gkd && gtheta && S > 0 && C > 0, "kdiag_bwd: bad arguments");
  hipLaunchKernelGGL(kdiag_bwd_kernel, dim3(cdiv(S, 64)), dim3(64), 0, as_stream(stream), theta, gkd, gtheta, S, C, D);
  return check_launch("kdiag_bwd");
}
