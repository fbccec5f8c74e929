// Micro-measurements behind the Cholesky kernel design (run on the GPU box):
//   cycles per wave64 v_fma_f64 (independent accumulators), per broadcast ds_read2_b64, per __syncthreads round.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_fma(double* out, uint64_t* cyc, int iters, double m) {
  double a[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) a[k] = threadIdx.x + k;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 20; ++k) a[k] = fma(m, a[k], 1.0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 20; ++k) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_fma32(float* out, uint64_t* cyc, int iters, float m) {
  float a[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) a[k] = threadIdx.x + k;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 20; ++k) a[k] = fmaf(m, a[k], 1.0f);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int k = 0; k < 20; ++k) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_lds(double* out, uint64_t* cyc, int iters) {
  __shared__ double buf[128];
  if (threadIdx.x < 128) buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int part = threadIdx.x % 5;
  double s = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 20; ++k) s += buf[part + 5 * k];
    asm volatile("" ::: "memory");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// all 64 lanes read the SAME address (full broadcast), 25 values + 50 FMAs per iteration: the column-per-wave step
__global__ void k_lds_same(double* out, uint64_t* cyc, int iters) {
  __shared__ double buf[256];
  buf[threadIdx.x & 255] = threadIdx.x * 1e-3;
  __syncthreads();
  double a[25], b[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) { a[k] = threadIdx.x + k; b[k] = threadIdx.x * 0.5 + k; }
  const double ma = 1e-3 * threadIdx.x, mb = -2e-3 * threadIdx.x;
  const int w = threadIdx.x >> 6;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const double* p = buf + ((it & 1) * 128) + w;
    double pv[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) pv[k] = p[4 * k];
    __builtin_amdgcn_sched_group_barrier(0x100, 25, 0);
#pragma unroll
    for (int k = 0; k < 25; ++k) { a[k] = fma(ma, pv[k], a[k]); b[k] = fma(mb, pv[k], b[k]); }
    asm volatile("" ::: "memory");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 25; ++k) s += a[k] + b[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_bar(double* out, uint64_t* cyc, int iters) {
  __shared__ double buf[2][8];
  double s = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x == (it & 127)) buf[it & 1][0] = s + it;
    __syncthreads();
    s += buf[it & 1][0];
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// column-per-wave step body: pivot-row values broadcast from one lane's registers (v_readlane -> SGPR operand)
__device__ __forceinline__ double bcast(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__global__ void k_bcast(double* out, uint64_t* cyc, int iters) {
  double a[25], b[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) { a[k] = threadIdx.x + k; b[k] = threadIdx.x * 0.5 + k; }
  const double ma = 1e-3 * threadIdx.x, mb = -2e-3 * threadIdx.x;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int lane = it & 63;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
      const double pv = bcast(a[k], lane);
      a[k] = fma(ma, pv, a[k]);
      b[k] = fma(mb, pv, b[k]);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 25; ++k) s += a[k] + b[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; uint64_t* cyc; float* outf;
  hipMalloc(&out, 1 << 20); hipMalloc(&outf, 1 << 20); hipMalloc(&cyc, 4096);
  uint64_t h[64];
  const int iters = 2000;
  for (int nt : {64, 256, 512}) {
    for (int rep = 0; rep < 2; ++rep) k_fma<<<30, nt>>>(out, cyc, iters, 0.999);
    hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
    printf("fma_f64  threads %3d: %.2f memtime ticks per wave-instruction\n", nt, (double)h[0] / (iters * 20.0));
    for (int rep = 0; rep < 2; ++rep) k_fma32<<<30, nt>>>(outf, cyc, iters, 0.999f);
    hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
    printf("fma_f32  threads %3d: %.2f ticks per wave-instruction\n", nt, (double)h[0] / (iters * 20.0));
    for (int rep = 0; rep < 2; ++rep) k_lds<<<30, nt>>>(out, cyc, iters);
    hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
    printf("lds b64  threads %3d: %.2f ticks per broadcast read (+ add)\n", nt, (double)h[0] / (iters * 20.0));
    for (int rep = 0; rep < 2; ++rep) k_bar<<<30, nt>>>(out, cyc, iters);
    hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
    printf("barrier  threads %3d: %.1f ticks per write+barrier+read round\n", nt, (double)h[0] / iters);
  }
  for (int rep = 0; rep < 2; ++rep) k_bcast<<<30, 256>>>(out, cyc, iters);
  hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
  printf("bcast step (25 x [2 readlane + 2 fma_f64]), 256 threads: %.1f ticks per step\n", (double)h[0] / iters);
  for (int rep = 0; rep < 2; ++rep) k_lds_same<<<30, 256>>>(out, cyc, iters);
  hipMemcpy(h, cyc, 8 * 30, hipMemcpyDeviceToHost);
  printf("same-address step (25 broadcast ds_read_b64 + 50 fma_f64), 256 threads: %.1f ticks per step\n", (double)h[0] / iters);
  // wall-clock calibration of the tick
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k_fma<<<30, 256>>>(out, cyc, 20000, 0.999); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("tick calibration: %.1f ticks per us\n", (double)h[0] / (ms * 1e3));
  return 0;
}
