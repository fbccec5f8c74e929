// Device helpers shared by the LDS-resident kernels of the first-task backward (t0_bwd_mid.h: per (s, c, column tile);
// t0_bwd_mat.h: per matrix): f32 MFMA 32x32x2 fragments with the k-pairing of gemm.hip, compile-time loops, staging loads.
#pragma once
#include "common.h"
#include <type_traits>

namespace vargp {


constexpr int kBmKP = 104;      // padded inner dimension (M <= 104, multiple of 8)
constexpr int kBmSA = 108;      // row stride of the M x M operand (G, then T): 108 / 4 odd -> conflict-free b128 rows
constexpr int kBmST = 68;       // row stride of the M x 64 tiles
constexpr size_t kBwdMidLdsBytes =
    sizeof(float) * (kBmKP * kBmSA + 2 * kBmKP * kBmST + 128 /*a*/ + 64 + 64 /*gmu, gvar*/ + 64 /*column sums*/ + 8 +
                     3 * 4 * 64 /*deferred softmax: partial sums of the four f-groups*/);
// deferred softmax (t0_bwd_mid_kernel evaluates the likelihood of its tile): at most kBmSmC classes, 4 kBmSmF likelihood samples
constexpr int kBmSmC = 16, kBmSmF = 4;
struct BmSoftmax {
  const float *mu, *var, *eps;      // (S, C, B), (S, C, B), (S, F, C, B); eps == NULL: gmu / gvar come from memory
  const int64_t* y;                 // (B)
  float* nll;                       // scalar accumulator (cleared by the forward)
  int F;
};
typedef float bm_f32x16 __attribute__((ext_vector_type(16)));

// fragment of a K-contiguous operand ([index][k]): the lane's 4 consecutive k of one row, one ds_read_b128
__device__ __forceinline__ float4 bm_frag_kc(const float* __restrict__ rowp, int k) {
  return *reinterpret_cast<const float4*>(rowp + k);
}
// fragment of a k-major operand ([k][index]): 4 rows, same column
__device__ __forceinline__ float4 bm_frag_km(const float* __restrict__ colp, int k, int stride) {
  const float* p = colp + k * stride;
  return make_float4(p[0], p[stride], p[2 * stride], p[3 * stride]);
}
__device__ __forceinline__ void bm_mfma4(bm_f32x16& acc, const float4 a, const float4 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// Workgroup barrier for LDS hand-overs only: waits for this wave's LDS operations, NOT for its outstanding global loads, stores
// and atomics (__syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier drains them all: a prefetch issued in front of it is
// waited for on the spot, a round of float atomics likewise).  Nothing in this kernel hands data over through global memory.
__device__ __forceinline__ void bmm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compile-time loop: f(integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void bm_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    bm_for<I0 + 1, I1>(f);
  }
}
__host__ __device__ constexpr int bm_min(int a, int b) { return a < b ? a : b; }

constexpr int kBmNQ = kBmKP / 4;                          // float4 per row of an M x M operand (26)
constexpr int kBmNA = (kBmKP * kBmNQ + 255) / 256;        // float4 per thread of one M x M operand (11)

// one M x M matrix (row-major, row stride ld in global memory) -> registers (clamped indices).  32-bit BYTE offsets from the
// (uniform) base: the load then takes its address as SGPR pair + one VGPR, and the address arithmetic stays 32-bit -- with
// 64-bit element indices every load carried a v_mad_i64 / v_lshl_add_u64 chain (a third of the instructions in front of the
// first barrier of t0_bwd_mid_kernel).  A matrix spans far less than 4 GB.
// TRI: the matrix is lower triangular (T, G): a float4 wholly above the diagonal is not fetched -- its load is pointed at the
// row's diagonal float4, which is fetched anyway, and the value is dropped when it is stored (bm_store_mat<true>).  The staging
// fronts are bound by the cache lines a CU can have in flight (10.8k cycles to ISSUE the 45 loads of t0_fwd_fused_kernel's
// front, whatever the address arithmetic), so lines not asked for are time saved; the instruction count does not change.
template <bool TRI = false>
__device__ __forceinline__ void bm_load_mat(const float* __restrict__ base, int ld, int M, int tid, float4 (&dst)[kBmNA]) {
  const char* bp = reinterpret_cast<const char*>(base);
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) {
    const int e = min(tid + 256 * u, kBmKP * kBmNQ - 1);
    const int i = e / kBmNQ, j = (e - i * kBmNQ) * 4;
    const int ic = min(i, M - 1);
    const int jc = TRI ? min(j, min(M - 4, ic & ~3)) : min(j, M - 4);
    const unsigned off = 4u * (__umul24((unsigned)ic, (unsigned)ld) + (unsigned)jc);
    dst[u] = *reinterpret_cast<const float4*>(bp + off);
  }
}

}  // namespace vargp
