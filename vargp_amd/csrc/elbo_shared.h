// Device code and small host helpers shared by the two native ELBO programs (elbo_t0.hip: first task; elbo_tn.hip:
// tasks t > 0 and any M): softplus / sigmoid, the counter-based normal generator, the fused softmax likelihood kernel,
// the hyper-parameter backward, the K-split rule and the flat batched-GEMM descriptor.  `static`: one copy per
// translation unit.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace vargp {

__device__ __forceinline__ float softplus_t0(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_t0(float x) { return 1.f / (1.f + expf(-x)); }

// ---- counter-based normal generator (Philox4x32-10 + Box-Muller) ---------------------------------------------------
// Element g of noise stream `stream` at step `step` is a pure function of (seed, stream, g, step): group g/4 is one
// Philox block, whose four 32-bit words make two Box-Muller pairs.  A rank that evaluates samples [s0, s0 + S) of a
// global draw simply offsets g, so every rank sees its slice of ONE global tensor without communication.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void normal4(uint64_t seed, uint32_t stream, uint64_t group, uint32_t step, float (&out)[4]) {
  uint32_t c[4] = {(uint32_t)group, (uint32_t)(group >> 32), stream, step};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u0 = ((float)c[2 * h] + 1.f) * 2.3283064365386963e-10f;       // (0, 1]
    const float u1 = (float)c[2 * h + 1] * 2.3283064365386963e-10f;            // [0, 1]
    const float r = sqrtf(-2.f * logf(u0));
    float sn, cs;
    sincospif(2.f * u1, &sn, &cs);
    out[2 * h] = r * cs; out[2 * h + 1] = r * sn;
  }
}
// the single element g of a stream (recomputes its group)
__device__ __forceinline__ float normal1(uint64_t seed, uint32_t stream, uint64_t g, uint32_t step) {
  float v[4];
  normal4(seed, stream, g >> 2, step, v);
  const int l = (int)(g & 3);
  return l == 0 ? v[0] : (l == 1 ? v[1] : (l == 2 ? v[2] : v[3]));
}
constexpr uint32_t kStreamTheta = 0, kStreamF = 1;

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
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const int64_t i = ((int64_t)s * C + c) * B + b;
      sd[c] = c < C ? sqrtf(var[i]) : 1.f;
      ev[c] = c < C ? eps[(((int64_t)s * F + f) * C + c) * B + b] : 0.f;
      v[c] = c < C ? mu[i] + sd[c] * ev[c] : -INFINITY;
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
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D1) return;
  const float hs = map_est ? 0.f : 0.5f * expf(0.5f * logvar[d]);
  float gm = 0.f, gv = 0.f;
  for (int s = 0; s < S; ++s) {
    float g = gtheta[s * D1 + d];
    if (d == D1 - 1) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc += gkd[s * C + c];
      g += 2.f * g2[s] * acc;
    }
    gm += g;
    if (!map_est) gv = fmaf(g * hs, eps[s * D1 + d], gv);
  }
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


}  // namespace vargp
