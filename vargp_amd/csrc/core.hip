// Error plumbing and small generic kernels (outer-sum reduction, fused Yogi step).
#include "common.h"
#include <stdarg.h>

namespace vargp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return VARGP_ELAUNCH;
  }
  return VARGP_OK;
}

__global__ void sum_outer_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t outer, int64_t inner) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= inner) return;
  float acc = 0.f;
  for (int64_t r = 0; r < outer; ++r) acc += in[r * inner + i];
  out[i] = acc;
}

// Yogi (Zaheer et al. 2018): v <- v - (1-b2) sign(v - g^2) g^2 ; p <- p - lr/bias1 * m / (sqrt(v/bias2) + eps)
__global__ void yogi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bias1,
                            float bias2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i], g2 = gi * gi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  float vi = v[i];
  const float df = vi - g2;
  const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
  vi = vi - (1.f - b2) * sg * g2;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bias2) + eps;
  p[i] -= (lr / bias1) * mi / denom;
}

}  // namespace vargp

using namespace vargp;

extern "C" int vargp_version(void) { return 100; }
extern "C" const char* vargp_last_error(void) { return g_err; }

extern "C" int vargp_sum_outer(const float* in, float* out, int64_t outer, int64_t inner, vargp_stream_t stream) {
  VARGP_REQUIRE(in && out && outer > 0 && inner > 0, "sum_outer: bad arguments");
  hipLaunchKernelGGL(sum_outer_kernel, dim3(cdiv(inner, 256)), dim3(256), 0, as_stream(stream), in, out, outer, inner);
  return check_launch("sum_outer");
}

extern "C" int vargp_yogi_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float bias1, float bias2, vargp_stream_t stream) {
  VARGP_REQUIRE(p && g && m && v && n > 0, "yogi_step: bad arguments");
  hipLaunchKernelGGL(yogi_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1,
                     beta2, eps, bias1, bias2);
  return check_launch("yogi_step");
}
