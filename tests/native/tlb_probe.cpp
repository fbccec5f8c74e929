// How long does ONE workgroup wait for a batch of independent 16-byte loads, as a function of how many distinct pages
// the batch touches?  (Round 4: the factorising workgroups of the merged first-task launch spend ~7 us on one round trip of 20
// float4 loads per thread.)  256 threads, thread t loads float4 number u at base + u * stride + 16 t: each load instruction
// covers 4 KB.  Reports 100 MHz wall-clock ticks for a cold pass (after a 512 MB sweep by another kernel) and a warm pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NL>
__global__ void probe(const float4* __restrict__ buf, long stride16, unsigned long long* out, float* sink) {
  const float4* p = buf + (long)blockIdx.x * ((stride16 <= 256 ? NL * stride16 : 0) + 8192) + threadIdx.x;   // (blocks 128 KB apart; contiguous ranges do not overlap)
  float4 v[NL];
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int u = 0; u < NL; ++u) v[u] = p[u * stride16];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < NL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 12345.678f) sink[0] = acc;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x * 2 + pass] = t1 - t0;
  }
}
__global__ void sweep(float* buf, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) buf[i] += 1.f;
}

int main() {
  const long big = 512l << 20;
  float *buf, *trash, *sink;
  unsigned long long* out;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&trash, big)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(buf, 0, big));
  const long strides[] = {0, 4096, 65536, 2l << 20};
  for (int nb : {1, 40}) {
    for (long st : strides) {
      double cold = 0, warm = 0;
      const int reps = 5;
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, trash, big / 4);
        hipLaunchKernelGGL((probe<20>), dim3(nb), dim3(256), 0, 0, (const float4*)buf, st / 16, out, sink);
        unsigned long long h[80];
        CK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
        double c = 0, w = 0;
        for (int b = 0; b < nb; ++b) { c += h[2 * b]; w += h[2 * b + 1]; }
        cold += c / nb; warm += w / nb;
      }
      printf("workgroups %2d  20 float4 loads / thread, %8ld B between loads: cold %.2f us  warm %.2f us\n", nb, st,
             cold / reps / 100.0, warm / reps / 100.0);
    }
  }
  return 0;
}
