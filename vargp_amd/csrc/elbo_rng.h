// Scalar device helpers of the native ELBO programs: softplus / sigmoid and the counter-based normal generator.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

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

}  // namespace vargp
