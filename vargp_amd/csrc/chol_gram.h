// Gram front of a pivot-chain workgroup (chol_small3.h, 64 < n <= 100, n % 4 == 0): the weighted Gram matrix
// G = (z sqrt(w)) (z sqrt(w))^T of ONE class's inducing points under ONE hyper-sample's 1/sigma^2, built by the workgroup that
// factorises K_uu = g2 exp(-(G_ii + G_jj - 2 G_ij) / 2) + eps I next -- the kernel matrix never leaves the CU before its factors do.
// Reference: kernels.py:24-44 (inputs scaled by the lengthscales, squared distances from the inner products), gp_utils.py:5-11.
//
// Why (round 6): with six or more hyper-samples the K-split Gram tiles of the front launch (64 x 64 MFMA tiles, M = 100 padded to
// 128: 1.64x the work, partial sums written and re-read) are a throughput problem, while a chain workgroup has time to spare
// under the K_uf product of its launch.  Here the Gram is 7 x 7 blocks of 16 x 16 (f32 MFMA 16x16x4: 100 -> 112 rows, 1.25x),
// lower triangle only (28 blocks, seven per wave), ONE staged operand panel (z sqrt(w): A and B fragments are the same rows).
//
// 256 threads = 4 waves.  Panel: 112 rows x 32 k, row stride 36 floats (b128 fragment reads of 16 rows x 4 k-quads hit 64
// distinct banks), double-buffered; both buffers and the sqrt(w) table live INSIDE the caller's staging matrix, which is only
// written after the K loop.  Fragment of a 16-row block for a group of 16 k: lane (l16, q) reads the float4 at row l16, k = 4 q
// .. 4 q + 3; component t feeds MFMA t, whose four inner indices are {t, 4 + t, 8 + t, 12 + t} -- the same permutation on both
// operands, so the sum over the 16 k is the plain one.
#pragma once
#include "common.h"
#include <type_traits>

namespace vargp {

typedef float cg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCgRows = 112, kCgBK = 32, kCgLS = kCgBK + 4;
constexpr int kCgPanel = kCgRows * kCgLS;           // floats per panel buffer
constexpr int kCgMaxD = 1024;                       // sqrt(w) table (floats of LDS behind the two panels)
constexpr int kCgLdsFloats = 2 * kCgPanel + kCgMaxD;

// wave WV: block row 6 - WV (7 - WV blocks: columns 0 .. 6 - WV) and, WV > 0, block row WV - 1 (WV blocks: columns 0 .. WV - 1)
template <int WV> struct CgTri {
  static constexpr int RA = 6 - WV, NA = 7 - WV, RB = WV > 0 ? WV - 1 : 0, NBk = WV;
  __host__ __device__ static constexpr int rb(int u) { return u < NA ? RA : RB; }
  __host__ __device__ static constexpr int cb(int u) { return u < NA ? u : u - NA; }
};

template <int I, int N, class F>
__device__ __forceinline__ void cg_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cg_for<I + 1, N>(f); }
}

__device__ __forceinline__ void cg_mfma4(cg_f32x4& acc, const float4 a, const float4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// z: the class's inducing points [n][D] (16-byte aligned, D % 4 == 0, D <= kCgMaxD); w: the sample's 1/sigma^2 [D];
// lds: kCgLdsFloats floats.  acc[u]: block (rb(u), cb(u)) of G, register r = row 16 rb + 4 (lane / 16) + r, column 16 cb + lane % 16.
// LOWER: z is lower triangular (n x n, zeros above the diagonal: the Cholesky factor of q(u)): a 16-row block has no entries in
// the k-groups to the right of it, so block (rb, cb) only takes the k-groups kb <= cb (<= rb) -- 84 of the 196 block products.
// PACKED (with LOWER): z is that factor's PACKED vector (row i: i + 1 entries at offset i (i + 1) / 2, gp_utils.py:22-49) and
// `diag` maps the stored diagonal entries to the factor's (softplus); D = n.
struct CgNoDiag { __device__ __forceinline__ float operator()(float v) const { return v; } };
template <int WV, bool LOWER = false, bool PACKED = false, class DiagF = CgNoDiag>
__device__ __forceinline__ void cg_gram(const float* __restrict__ z, const float* __restrict__ w, const int n, const int D,
                                        float* __restrict__ lds, cg_f32x4 (&acc)[7], const int tid, DiagF diag = DiagF()) {
  using W = CgTri<WV>;
  float* sw = lds + 2 * kCgPanel;
  const int lane = tid & 63, l16 = lane & 15, q = lane >> 4;
  const int nslab = (D + kCgBK - 1) / kCgBK;
  // staging: 112 rows x 8 float4 per slab = 896 float4, four rounds of 256 threads (the last one half full)
  int srow[4], sq[4];
  const float* zp[4];
  bool live[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    srow[u] = e >> 3; sq[u] = (e & 7) << 2;
    live[u] = e < kCgRows * 8;
    zp[u] = z + (int64_t)min(srow[u], n - 1) * D + sq[u];
  }
  float4 rg[4];
  auto load_slab = [&](int s) {
    const int k0 = min(s, nslab - 1) * kCgBK;
    if constexpr (PACKED) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = min(srow[u], n - 1), k = min(k0 + sq[u], D - 4);
        const float* rp = z + (int64_t)row * (row + 1) / 2;
        float e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = rp[min(k + t, row)];                 // (clamped: entries right of the diagonal are masked)
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = k + t < row ? e[t] : (k + t == row ? diag(e[t]) : 0.f);
        rg[u] = make_float4(e[0], e[1], e[2], e[3]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) rg[u] = *reinterpret_cast<const float4*>(zp[u] + min(k0, D - 4 - sq[u]));
    }
  };
  auto store_slab = [&](int s, float* __restrict__ buf) {
    const int k0 = s * kCgBK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + sq[u];
      const bool ok = srow[u] < n && k < D;                   // rows n .. 111 and the K overhang of the last slab: zeros
      const float4 s4 = *reinterpret_cast<const float4*>(&sw[min(k, D - 4)]);
      const float4 v = rg[u];
      const float4 o = ok ? make_float4(v.x * s4.x, v.y * s4.y, v.z * s4.z, v.w * s4.w) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[u]) *reinterpret_cast<float4*>(&buf[srow[u] * kCgLS + sq[u]]) = o;
    }
  };
  load_slab(0);
  for (int d = tid; d < D; d += 256) sw[d] = w ? sqrtf(w[d]) : 1.f;          // (w == NULL: the plain Gram matrix z z^T)
#pragma unroll
  for (int u = 0; u < 7; ++u) acc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  store_slab(0, lds);
  load_slab(1);
  __syncthreads();
  const int foff = l16 * kCgLS + 4 * q;
  for (int s = 0; s < nslab; ++s) {
    const float* buf = lds + (s & 1) * kCgPanel;
    float4 fb[2][W::NA];
    auto frag = [&](int h) {
      cg_for<0, W::NA>([&](auto ci) {
        constexpr int cbk = decltype(ci)::value;
        fb[h][cbk] = *reinterpret_cast<const float4*>(&buf[(16 * cbk) * kCgLS + foff + 16 * h]);
      });
    };
    frag(0);
    frag(1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kb = 2 * s + h;                            // (uniform) the k-group of 16 this half-slab holds
      // (a block row's own fragment is the column fragment of its diagonal block: RA = NA - 1, and RB < NA for every wave)
      cg_for<0, W::NA>([&](auto ui) {
        constexpr int u = decltype(ui)::value;
        if (!LOWER || u >= kb) cg_mfma4(acc[u], fb[h][W::RA], fb[h][u]);
      });
      if constexpr (W::NBk > 0)
        cg_for<0, W::NBk>([&](auto ui) {
          constexpr int u = decltype(ui)::value;
          if (!LOWER || u >= kb) cg_mfma4(acc[W::NA + u], fb[h][W::RB], fb[h][u]);
        });
    }
    if (s + 1 < nslab) store_slab(s + 1, lds + ((s + 1) & 1) * kCgPanel);
    load_slab(s + 2);
    // LDS-only barrier: __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier would wait for the loads just issued
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __syncthreads();      // (the caller overwrites the panels)
}

// acc -> G in the staging matrix (row stride LS, both triangles) and its diagonal in `diag` (>= n floats).  dump: 64 floats of LDS
// nobody reads -- entries that do not exist (rows / columns >= n, the upper half of a diagonal block) are stored THERE, one slot per
// lane, instead of being branched around: 56 predicated stores per lane compiled to 56 exec-mask branches (2.1 us per matrix)
template <int WV>
__device__ __forceinline__ void cg_store(const cg_f32x4 (&acc)[7], float* __restrict__ stage, const int LS, float* __restrict__ diag,
                                         const int n, const int lane, float* __restrict__ dump) {
  using W = CgTri<WV>;
  const int l16 = lane & 15, q = lane >> 4;
  float* const mine = dump + lane;
  cg_for<0, 7>([&](auto ui) {
    constexpr int u = decltype(ui)::value;
    constexpr int rb = W::rb(u), cb = W::cb(u);
    const int j = 16 * cb + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * rb + 4 * q + r;
      const bool ok = i < n && j < n && (rb != cb || j <= i);
      const float v = acc[u][r];
      *(ok ? &stage[i * LS + j] : mine) = v;
      *(ok ? &stage[j * LS + i] : mine) = v;
      if constexpr (rb == cb) *((ok && i == j) ? &diag[i] : mine) = v;
    }
  });
}

// acc -> a symmetric n x n matrix in memory (row stride ld, both triangles)
template <int WV>
__device__ __forceinline__ void cg_store_global(const cg_f32x4 (&acc)[7], float* __restrict__ out, const int ld, const int n, const int lane) {
  using W = CgTri<WV>;
  const int l16 = lane & 15, q = lane >> 4;
  cg_for<0, 7>([&](auto ui) {
    constexpr int u = decltype(ui)::value;
    constexpr int rb = W::rb(u), cb = W::cb(u);
    const int j = 16 * cb + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * rb + 4 * q + r;
      if (i < n && j < n && (rb != cb || j <= i)) {
        out[(int64_t)i * ld + j] = acc[u][r];
        out[(int64_t)j * ld + i] = acc[u][r];
      }
    }
  });
}

}  // namespace vargp
