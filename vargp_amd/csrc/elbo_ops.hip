// Small fused kernels around the GEMM/Cholesky core of the ELBO: packed-triangle <-> matrix,
// predictive mean/variance column reductions, MVN-KL reduction, log-det, Monte-Carlo softmax
// likelihood.  All are HBM/latency-bound elementwise or reduction kernels: coalesced loads along the
// contiguous dim, wave64 shuffles for the reductions.
#include "common.h"

namespace vargp {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ---- vec2tril (var_gp/gp_utils.py:22-49) ------------------------------------------------------
__global__ void vec2tril_fwd_kernel(const float* __restrict__ vec, float* __restrict__ tril, int m, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t mm = (int64_t)m * m, b = e / mm, r = e % mm;
  const int i = r / m, j = r % m;
  float v = 0.f;
  if (j <= i) {
    v = vec[b * ((int64_t)m * (m + 1) / 2) + (int64_t)i * (i + 1) / 2 + j];
    if (i == j) v = softplus_f(v);
  }
  tril[e] = v;
}
__global__ void vec2tril_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ gtril,
                                    float* __restrict__ gvec, int m, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t mm = (int64_t)m * m, b = e / mm, r = e % mm;
  const int i = r / m, j = r % m;
  if (j > i) return;
  const int64_t idx = b * ((int64_t)m * (m + 1) / 2) + (int64_t)i * (i + 1) / 2 + j;
  float g = gtril[e];
  if (i == j) { const float x = vec[idx]; g *= (x > 20.f) ? 1.f : sigmoid_f(x); }
  gvec[idx] = g;
}
__global__ void mat2trilvec_kernel(const float* __restrict__ mat, float* __restrict__ vec, int m, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t mm = (int64_t)m * m, b = e / mm, r = e % mm;
  const int i = r / m, j = r % m;
  if (j <= i) vec[b * ((int64_t)m * (m + 1) / 2) + (int64_t)i * (i + 1) / 2 + j] = mat[e];
}

// ---- predictive diag (var_gp/gp_utils.py:178-186) ---------------------------------------------
// grid (ceil(B/64), nb): 64 columns x 4 row lanes per block, coalesced across the column index
__global__ __launch_bounds__(256) void pdiag_fwd_kernel(const float* __restrict__ P, const float* __restrict__ W,
                                                        const float* __restrict__ a, int64_t a_stride,
                                                        int64_t a_bstride, const float* __restrict__ kd,
                                                        float* __restrict__ mu, float* __restrict__ var, int M, int B) {
  __shared__ float red[3][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int64_t b = blockIdx.y;
  float m0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (col < B) {
    const float* p = P + b * M * B + col;
    const float* w = W + b * M * B + col;
    const float* av = a + b * a_bstride;
#pragma unroll 4
    for (int m = ry; m < M; m += 4) {
      const float pv = p[(int64_t)m * B], wv = w[(int64_t)m * B];
      m0 = fmaf(pv, av[m * a_stride], m0);
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
}
// grid (M, nb): one block per row; gP, gW elementwise, ga row reduction; block m == 0 also reduces gkd
__global__ __launch_bounds__(256) void pdiag_bwd_kernel(const float* __restrict__ P, const float* __restrict__ W,
                                                        const float* __restrict__ a, int64_t a_stride,
                                                        int64_t a_bstride, const float* __restrict__ gmu,
                                                        const float* __restrict__ gvar, float* __restrict__ gP,
                                                        float* __restrict__ gW, float* __restrict__ ga,
                                                        float* __restrict__ gkd, int M, int B) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int64_t off = (b * M + m) * B;
  const float am = a[b * a_bstride + m * a_stride];
  float acc = 0.f, accv = 0.f;
  for (int col = threadIdx.x; col < B; col += 256) {
    const float gm = gmu[b * B + col], gv = gvar[b * B + col];
    const float pv = P[off + col], wv = W[off + col];
    gP[off + col] = am * gm - 2.f * pv * gv;
    gW[off + col] = 2.f * wv * gv;
    acc = fmaf(pv, gm, acc);
    accv += gv;
  }
  const float t = block_sum<256>(acc, red);
  if (threadIdx.x == 0) ga[b * M + m] = t;
  if (m == 0) {
    const float tv = block_sum<256>(accv, red);
    if (threadIdx.x == 0) gkd[b] = tv;
  }
}

// ---- MVN KL reduction and log-det --------------------------------------------------------------
__global__ __launch_bounds__(256) void mvn_kl_fwd_kernel(const float* __restrict__ G, const float* __restrict__ d,
                                                         const float* __restrict__ ldp, const float* __restrict__ ldq,
                                                         float* __restrict__ kl, int M) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const float* g = G + b * M * M;
  float acc = 0.f;
  for (int e = threadIdx.x; e < M * M; e += 256) { const float v = g[e]; acc = fmaf(v, v, acc); }
  for (int e = threadIdx.x; e < M; e += 256) { const float v = d[b * M + e]; acc = fmaf(v, v, acc); }
  const float t = block_sum<256>(acc, red);
  if (threadIdx.x == 0) kl[b] = ldp[b] - ldq[b] + 0.5f * (t - (float)M);
}
__global__ void mvn_kl_bwd_kernel(const float* __restrict__ G, const float* __restrict__ d,
                                  const float* __restrict__ gkl, float* __restrict__ gG, float* __restrict__ gd,
                                  int M, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t per = (int64_t)M * M + M, b = e / per, r = e % per;
  const float g = gkl[b];
  if (r < (int64_t)M * M) gG[b * M * M + r] = g * G[b * M * M + r];
  else gd[b * M + (r - (int64_t)M * M)] = g * d[b * M + (r - (int64_t)M * M)];
}
__global__ __launch_bounds__(64) void logdet_fwd_kernel(const float* __restrict__ L, float* __restrict__ out, int n) {
  const int64_t b = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += logf(L[b * n * n + (int64_t)i * n + i]);
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[b] = acc;
}
__global__ void logdet_bwd_kernel(const float* __restrict__ L, const float* __restrict__ g, float* __restrict__ gL,
                                  int n, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t nn = (int64_t)n * n, b = e / nn, r = e % nn;
  const int i = r / n, j = r % n;
  gL[e] = (i == j) ? g[b] / L[e] : 0.f;
}

// ---- Monte-Carlo softmax likelihood (var_gp/likelihoods.py:13-63) -----------------------------
// log-sum-exp over classes of f_c = mu[s,c,b] + sqrt(var[s,c,b]) eps[s,f,c,b]
__device__ __forceinline__ float lse_classes(const float* __restrict__ mu, const float* __restrict__ var,
                                             const float* __restrict__ eps, int s, int f, int b, int F, int C, int B) {
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) {
    const int64_t i = ((int64_t)s * C + c) * B + b;
    const float v = mu[i] + sqrtf(var[i]) * eps[(((int64_t)s * F + f) * C + c) * B + b];
    mx = fmaxf(mx, v);
  }
  float se = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t i = ((int64_t)s * C + c) * B + b;
    const float v = mu[i] + sqrtf(var[i]) * eps[(((int64_t)s * F + f) * C + c) * B + b];
    se += expf(v - mx);
  }
  return mx + logf(se);
}
// one thread per (s, f, b); grid over S*F*B
__global__ __launch_bounds__(256) void softmax_nll_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              const float* __restrict__ eps, const int64_t* __restrict__ y,
                                                              float* __restrict__ nll, int S, int F, int C, int B) {
  __shared__ float red[4];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (e < (int64_t)S * F * B) {
    const int b = e % B, f = (e / B) % F, s = e / ((int64_t)B * F);
    const float lse = lse_classes(mu, var, eps, s, f, b, F, C, B);
    const int c = (int)y[b];
    const int64_t i = ((int64_t)s * C + c) * B + b;
    const float fy = mu[i] + sqrtf(var[i]) * eps[(((int64_t)s * F + f) * C + c) * B + b];
    contrib = -(fy - lse) / (float)(S * F);
  }
  const float t = block_sum<256>(contrib, red);
  if (threadIdx.x == 0) atomicAdd(nll, t);
}
// one thread per (s, c, b): sums over f of (softmax_c - 1[c == y]) and its eps-weighted version
__global__ __launch_bounds__(256) void softmax_nll_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              const float* __restrict__ eps, const int64_t* __restrict__ y,
                                                              const float* __restrict__ gnll, float* __restrict__ gmu,
                                                              float* __restrict__ gvar, int S, int F, int C, int B) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * C * B) return;
  const int b = e % B, c = (e / B) % C, s = e / ((int64_t)B * C);
  const float sd = sqrtf(var[e]), m = mu[e];
  const float ind = ((int)y[b] == c) ? 1.f : 0.f;
  float a0 = 0.f, a1 = 0.f;
  for (int f = 0; f < F; ++f) {
    const float lse = lse_classes(mu, var, eps, s, f, b, F, C, B);
    const float ep = eps[(((int64_t)s * F + f) * C + c) * B + b];
    const float p = expf(m + sd * ep - lse) - ind;
    a0 += p;
    a1 = fmaf(p, ep, a1);
  }
  const float sc = gnll[0] / (float)(S * F);
  gmu[e] = sc * a0;
  gvar[e] = sc * a1 * 0.5f / sd;
}
// C <= CMAX: one thread per (s, f, b) keeps the class vector in registers and adds its share of the two
// gradients with float atomics (gmu / gvar pre-zeroed by the launcher); S*F*B threads instead of S*B
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_nll_bwd_small_kernel(const float* __restrict__ mu,
                                                                    const float* __restrict__ var,
                                                                    const float* __restrict__ eps,
                                                                    const int64_t* __restrict__ y,
                                                                    const float* __restrict__ gnll,
                                                                    float* __restrict__ gmu, float* __restrict__ gvar,
                                                                    int S, int F, int C, int B) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * F * B) return;
  const int b = e % B, f = (e / B) % F, s = e / ((int64_t)B * F);
  const int yb = (int)y[b];
  float sd[CMAX], ev[CMAX], v[CMAX], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const int64_t i = ((int64_t)s * C + c) * B + b;
    sd[c] = c < C ? sqrtf(var[i]) : 1.f;
    ev[c] = c < C ? eps[(((int64_t)s * F + f) * C + c) * B + b] : 0.f;
    v[c] = c < C ? mu[i] + sd[c] * ev[c] : -INFINITY;
    mx = fmaxf(mx, v[c]);
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { v[c] = c < C ? expf(v[c] - mx) : 0.f; se += v[c]; }
  const float sc = gnll[0] / (float)(S * F) / se;
  const float sc1 = gnll[0] / (float)(S * F);
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

// probs[b, c] = mean_{s,f} softmax_c ; one thread per (b, c)
__global__ __launch_bounds__(256) void softmax_predict_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              const float* __restrict__ eps, float* __restrict__ probs,
                                                              int S, int F, int C, int B) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)C * B) return;
  const int b = e % B, c = e / B;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) {
    const int64_t i = ((int64_t)s * C + c) * B + b;
    const float m = mu[i], sd = sqrtf(var[i]);
    for (int f = 0; f < F; ++f) {
      const float lse = lse_classes(mu, var, eps, s, f, b, F, C, B);
      acc += expf(m + sd * eps[(((int64_t)s * F + f) * C + c) * B + b] - lse);
    }
  }
  probs[(int64_t)b * C + c] = acc / (float)(S * F);
}

// ---- variational hyper-parameters (var_gp/kernels.py:62-77) -----------------------------------
// theta[s,d] = mean[d] + eps[s,d] * exp(0.5 * logvar[d])
__global__ void hyper_sample_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                        const float* __restrict__ eps, float* __restrict__ theta, int S, int D1) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * D1) return;
  const int d = e % D1;
  theta[e] = mean[d] + eps[e] * expf(0.5f * logvar[d]);
}
__global__ void hyper_sample_bwd_kernel(const float* __restrict__ logvar, const float* __restrict__ eps,
                                        const float* __restrict__ gtheta, float* __restrict__ gmean,
                                        float* __restrict__ glogvar, int S, int D1) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D1) return;
  const float hs = 0.5f * expf(0.5f * logvar[d]);
  float gm = 0.f, gv = 0.f;
  for (int s = 0; s < S; ++s) {
    const float g = gtheta[s * D1 + d];
    gm += g;
    gv = fmaf(g * hs, eps[s * D1 + d], gv);
  }
  gmean[d] = gm;
  glogvar[d] = gv;
}
// kl = sum_d 0.5 * (exp(v - v0) + (m - m0)^2 / exp(v0) - 1 - (v - v0));  single block
__global__ __launch_bounds__(256) void hyper_kl_fwd_kernel(const float* __restrict__ m, const float* __restrict__ v,
                                                           const float* __restrict__ m0, const float* __restrict__ v0,
                                                           float* __restrict__ kl, int D1) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int d = threadIdx.x; d < D1; d += 256) {
    const float dv = v[d] - v0[d], dm = m[d] - m0[d];
    acc += 0.5f * (expf(dv) + dm * dm * expf(-v0[d]) - 1.f - dv);
  }
  const float t = block_sum<256>(acc, red);
  if (threadIdx.x == 0) kl[0] = t;
}
__global__ void hyper_kl_bwd_kernel(const float* __restrict__ m, const float* __restrict__ v,
                                    const float* __restrict__ m0, const float* __restrict__ v0,
                                    const float* __restrict__ gkl, float* __restrict__ gm, float* __restrict__ gv,
                                    int D1) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D1) return;
  const float g = gkl[0];
  gm[d] = g * (m[d] - m0[d]) * expf(-v0[d]);
  gv[d] = g * 0.5f * (expf(v[d] - v0[d]) - 1.f);
}

}  // namespace vargp

using namespace vargp;
#define GRID1(total) dim3(cdiv((total), 256)), dim3(256), 0, as_stream(stream)

// ---- bias + ReLU of the deep-kernel feature map (var_gp/kernels.py:80-96: Linear -> ReLU -> Linear -> ReLU -> Linear) --
// y[r, c] = act(x[r, c] + bias[c]); one thread per element, rows x cols row-major
__global__ void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
                                    int64_t total, int cols, int relu) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const float v = x[e] + bias[e % cols];
  y[e] = relu ? fmaxf(v, 0.f) : v;
}
// gx = gy * (y > 0) (ReLU) or gy; gbias[c] = sum_r gx[r, c].  64 columns x 4 row lanes per block, kBiasRows rows per block;
// column sums meet in gbias with float atomics (pre-zeroed by the launcher).
constexpr int kBiasRows = 64;
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                           float* __restrict__ gx, float* __restrict__ gbias, int64_t rows,
                                                           int cols, int relu) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int64_t r0 = (int64_t)blockIdx.y * kBiasRows;
  float acc = 0.f;
  if (col < cols) {
    for (int64_t r = r0 + ry; r < rows && r < r0 + kBiasRows; r += 4) {
      const int64_t e = r * cols + col;
      const float g = (relu && !(y[e] > 0.f)) ? 0.f : gy[e];
      gx[e] = g;
      acc += g;
    }
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < cols) atomicAdd(&gbias[col], red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

extern "C" int vargp_bias_act_fwd(const float* x, const float* bias, float* y, int64_t rows, int cols, int relu,
                                  vargp_stream_t stream) {
  VARGP_REQUIRE(x && bias && y && rows >= 0 && cols > 0, "bias_act_fwd: bad arguments");
  const int64_t total = rows * cols;
  if (total == 0) return VARGP_OK;
  hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, bias, y, total, cols, relu);
  return check_launch("bias_act_fwd");
}

extern "C" int vargp_bias_act_bwd(const float* y, const float* gy, float* gx, float* gbias, int64_t rows, int cols, int relu,
                                  vargp_stream_t stream) {
  VARGP_REQUIRE(y && gy && gx && gbias && rows >= 0 && cols > 0, "bias_act_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  zero_async(gbias, sizeof(float) * cols, st);
  if (rows == 0) return VARGP_OK;
  hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(cdiv(cols, 64), cdiv(rows, kBiasRows)), dim3(256), 0, st, y, gy, gx, gbias, rows,
                     cols, relu);
  return check_launch("bias_act_bwd");
}

extern "C" int vargp_vec2tril_fwd(const float* vec, float* tril, int nbatch, int m, vargp_stream_t stream) {
  VARGP_REQUIRE(vec && tril && nbatch > 0 && m > 0, "vec2tril_fwd: bad arguments");
  const int64_t total = (int64_t)nbatch * m * m;
  hipLaunchKernelGGL(vec2tril_fwd_kernel, GRID1(total), vec, tril, m, total);
  return check_launch("vec2tril_fwd");
}
extern "C" int vargp_vec2tril_bwd(const float* vec, const float* gtril, float* gvec, int nbatch, int m,
                                  vargp_stream_t stream) {
  VARGP_REQUIRE(vec && gtril && gvec && nbatch > 0 && m > 0, "vec2tril_bwd: bad arguments");
  const int64_t total = (int64_t)nbatch * m * m;
  hipLaunchKernelGGL(vec2tril_bwd_kernel, GRID1(total), vec, gtril, gvec, m, total);
  return check_launch("vec2tril_bwd");
}
extern "C" int vargp_mat2trilvec(const float* mat, float* vec, int nbatch, int m, vargp_stream_t stream) {
  VARGP_REQUIRE(mat && vec && nbatch > 0 && m > 0, "mat2trilvec: bad arguments");
  const int64_t total = (int64_t)nbatch * m * m;
  hipLaunchKernelGGL(mat2trilvec_kernel, GRID1(total), mat, vec, m, total);
  return check_launch("mat2trilvec");
}

extern "C" int vargp_predictive_diag_fwd(const float* P, const float* W, const float* a, int64_t a_stride,
                                         int64_t a_bstride, const float* kdiag, float* mu, float* var, int nbatch,
                                         int M, int B, vargp_stream_t stream) {
  VARGP_REQUIRE(P && W && a && kdiag && mu && var && nbatch > 0 && M > 0 && B > 0, "predictive_diag_fwd: bad arguments");
  VARGP_REQUIRE(nbatch <= 65535, "predictive_diag_fwd: batch too large");
  hipLaunchKernelGGL(pdiag_fwd_kernel, dim3(cdiv(B, 64), nbatch), dim3(256), 0, as_stream(stream), P, W, a, a_stride,
                     a_bstride, kdiag, mu, var, M, B);
  return check_launch("predictive_diag_fwd");
}
extern "C" int vargp_predictive_diag_bwd(const float* P, const float* W, const float* a, int64_t a_stride,
                                         int64_t a_bstride, const float* gmu, const float* gvar, float* gP, float* gW,
                                         float* ga, float* gkdiag, int nbatch, int M, int B, vargp_stream_t stream) {
  VARGP_REQUIRE(P && W && a && gmu && gvar && gP && gW && ga && gkdiag, "predictive_diag_bwd: null pointer");
  VARGP_REQUIRE(nbatch <= 65535 && M <= 65535 * 32, "predictive_diag_bwd: dims too large");
  hipLaunchKernelGGL(pdiag_bwd_kernel, dim3(M, nbatch), dim3(256), 0, as_stream(stream), P, W, a, a_stride, a_bstride, gmu,
                     gvar, gP, gW, ga, gkdiag, M, B);
  return check_launch("predictive_diag_bwd");
}

extern "C" int vargp_mvn_kl_fwd(const float* G, const float* d, const float* logdet_p, const float* logdet_q, float* kl,
                                int nbatch, int M, vargp_stream_t stream) {
  VARGP_REQUIRE(G && d && logdet_p && logdet_q && kl && nbatch > 0 && M > 0, "mvn_kl_fwd: bad arguments");
  hipLaunchKernelGGL(mvn_kl_fwd_kernel, dim3(nbatch), dim3(256), 0, as_stream(stream), G, d, logdet_p, logdet_q, kl, M);
  return check_launch("mvn_kl_fwd");
}
extern "C" int vargp_mvn_kl_bwd(const float* G, const float* d, const float* gkl, float* gG, float* gd, int nbatch,
                                int M, vargp_stream_t stream) {
  VARGP_REQUIRE(G && d && gkl && gG && gd && nbatch > 0 && M > 0, "mvn_kl_bwd: bad arguments");
  const int64_t total = (int64_t)nbatch * ((int64_t)M * M + M);
  hipLaunchKernelGGL(mvn_kl_bwd_kernel, GRID1(total), G, d, gkl, gG, gd, M, total);
  return check_launch("mvn_kl_bwd");
}
extern "C" int vargp_logdet_tril_fwd(const float* L, float* logdet, int nbatch, int n, vargp_stream_t stream) {
  VARGP_REQUIRE(L && logdet && nbatch > 0 && n > 0, "logdet_tril_fwd: bad arguments");
  hipLaunchKernelGGL(logdet_fwd_kernel, dim3(nbatch), dim3(64), 0, as_stream(stream), L, logdet, n);
  return check_launch("logdet_tril_fwd");
}
extern "C" int vargp_logdet_tril_bwd(const float* L, const float* g, float* gL, int nbatch, int n,
                                     vargp_stream_t stream) {
  VARGP_REQUIRE(L && g && gL && nbatch > 0 && n > 0, "logdet_tril_bwd: bad arguments");
  const int64_t total = (int64_t)nbatch * n * n;
  hipLaunchKernelGGL(logdet_bwd_kernel, GRID1(total), L, g, gL, n, total);
  return check_launch("logdet_tril_bwd");
}

extern "C" int vargp_softmax_nll_fwd(const float* mu, const float* var, const float* eps, const int64_t* y, float* nll,
                                     int S, int F, int C, int B, vargp_stream_t stream) {
  VARGP_REQUIRE(mu && var && eps && y && nll && S > 0 && F > 0 && C > 0 && B > 0, "softmax_nll_fwd: bad arguments");
  zero_async(nll, sizeof(float), as_stream(stream));
  const int64_t total = (int64_t)S * F * B;
  hipLaunchKernelGGL(softmax_nll_fwd_kernel, GRID1(total), mu, var, eps, y, nll, S, F, C, B);
  return check_launch("softmax_nll_fwd");
}
extern "C" int vargp_softmax_nll_bwd(const float* mu, const float* var, const float* eps, const int64_t* y,
                                     const float* gnll, float* gmu, float* gvar, int S, int F, int C, int B,
                                     vargp_stream_t stream) {
  VARGP_REQUIRE(mu && var && eps && y && gnll && gmu && gvar, "softmax_nll_bwd: null pointer");
  if (C <= 16) {
    const int64_t total = (int64_t)S * F * B;
    zero_async(gmu, sizeof(float) * (size_t)S * C * B, as_stream(stream));
    zero_async(gvar, sizeof(float) * (size_t)S * C * B, as_stream(stream));
    hipLaunchKernelGGL(softmax_nll_bwd_small_kernel<16>, GRID1(total), mu, var, eps, y, gnll, gmu, gvar, S, F, C, B);
  } else {
    const int64_t total = (int64_t)S * C * B;
    hipLaunchKernelGGL(softmax_nll_bwd_kernel, GRID1(total), mu, var, eps, y, gnll, gmu, gvar, S, F, C, B);
  }
  return check_launch("softmax_nll_bwd");
}
extern "C" int vargp_softmax_predict(const float* mu, const float* var, const float* eps, float* probs, int S, int F,
                                     int C, int B, vargp_stream_t stream) {
  VARGP_REQUIRE(mu && var && eps && probs && S > 0 && F > 0 && C > 0 && B > 0, "softmax_predict: bad arguments");
  const int64_t total = (int64_t)C * B;
  hipLaunchKernelGGL(softmax_predict_kernel, GRID1(total), mu, var, eps, probs, S, F, C, B);
  return check_launch("softmax_predict");
}

extern "C" int vargp_hyper_sample_fwd(const float* mean, const float* logvar, const float* eps, float* theta, int S,
                                      int D1, vargp_stream_t stream) {
  VARGP_REQUIRE(mean && logvar && eps && theta && S > 0 && D1 > 0, "hyper_sample_fwd: bad arguments");
  hipLaunchKernelGGL(hyper_sample_fwd_kernel, GRID1((int64_t)S * D1), mean, logvar, eps, theta, S, D1);
  return check_launch("hyper_sample_fwd");
}
extern "C" int vargp_hyper_sample_bwd(const float* logvar, const float* eps, const float* gtheta, float* gmean,
                                      float* glogvar, int S, int D1, vargp_stream_t stream) {
  VARGP_REQUIRE(logvar && eps && gtheta && gmean && glogvar && S > 0 && D1 > 0, "hyper_sample_bwd: bad arguments");
  hipLaunchKernelGGL(hyper_sample_bwd_kernel, GRID1((int64_t)D1), logvar, eps, gtheta, gmean, glogvar, S, D1);
  return check_launch("hyper_sample_bwd");
}
extern "C" int vargp_hyper_kl_fwd(const float* mean, const float* logvar, const float* prior_mean,
                                  const float* prior_logvar, float* kl, int D1, vargp_stream_t stream) {
  VARGP_REQUIRE(mean && logvar && prior_mean && prior_logvar && kl && D1 > 0, "hyper_kl_fwd: bad arguments");
  hipLaunchKernelGGL(hyper_kl_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), mean, logvar, prior_mean,
                     prior_logvar, kl, D1);
  return check_launch("hyper_kl_fwd");
}
extern "C" int vargp_hyper_kl_bwd(const float* mean, const float* logvar, const float* prior_mean,
                                  const float* prior_logvar, const float* gkl, float* gmean, float* glogvar, int D1,
                                  vargp_stream_t stream) {
  VARGP_REQUIRE(mean && logvar && prior_mean && prior_logvar && gkl && gmean && glogvar, "hyper_kl_bwd: null pointer");
  hipLaunchKernelGGL(hyper_kl_bwd_kernel, GRID1((int64_t)D1), mean, logvar, prior_mean, prior_logvar, gkl, gmean,
                     glogvar, D1);
  return check_launch("hyper_kl_bwd");
}
