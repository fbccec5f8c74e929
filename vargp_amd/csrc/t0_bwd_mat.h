// The per-matrix tail of the first-task backward as ONE LDS-resident workgroup per matrix (n = M <= 104, M % 4 == 0):
// everything between the tile kernel (t0_bwd_mid.h) and the kernel-matrix backward that is a chain of M x M x M products on
// ONE matrix -- six of them, each waiting for the one before -- and that, launched as batched GEMMs, costs seven launches
// of ~8-17 us in which the chip idles (40 matrices of 100 x 100: 160 tiles).  Here a workgroup keeps T = Lz^-1 and two
// scratch matrices in LDS and walks the chain; the launch is shared with the big product that does not depend on it
// (P_uf = W_uf x of the kernel-matrix backward: gemm.hip, t0_bwdmat_gemm_kernel), which hides it.
//
// K_uu role (matrix id = (s, c); reference: autograd of gp_utils.py:150-191 through torch.cholesky / triangular_solve):
//   gT   = gT_tiles + tril(ga m^T + gG L_S^T + gG2 Lu^T)                  (small columns of gT = tril(gQP RK^T), QP = T RK)
//   g_u_mean[c] += T^T ga (atomics over s),   gLu_part[s, c] = tril(T^T gG2)  (small columns of gRK = T^T gQP; the consumers sum over s)
//   w1 = gT T^T,  S = sym(0.5 (g I - tril(w1))),  gK = T^T S T              (Cholesky + inverse adjoint, chol.hip: gL is
//                                                                            diag(g / L_ii) here, so L^T tril(gL) = g I)
//   W_uu = 2 gK o K_uu, r_uu = its row sums, gtheta[s, D] += sum W_uu       (the K_uu role of the W = gK o K pass)
// S_u role (id = S C + c; q(u)'s covariance factor L_S = chol(Lu Lu^T + eps I), gp_utils.py:182):
//   gL = sum_s tril(T_s^T gG_s)   (the L_S block of gRK, recomputed here so that the role needs nothing of the K_uu roles: same
//   launch),  S = sym(0.5 tril(L_S^T gL)),  gS_u = T_S^T S T_S   -> gKS[S C + c]
//
// 256 threads = 4 waves; f32 MFMA 32x32x2 with the k-pairing of gemm.hip; results are 4 x 4 blocks of 32 x 32, dealt to the
// waves by static tables that balance the (triangular) K ranges.  Everything about a wave's blocks is a compile-time
// constant (the body is instantiated per wave): fully unrolled k loops with the next group's fragments requested before
// the current group's MFMAs, scalar control flow.
#pragma once
#include "t0_bwd_common.h"

namespace vargp {

constexpr int kMatS = 108;                                    // row stride of every M x M LDS matrix (108 / 4 odd)
constexpr int kMatN = kBmKP * kMatS;                          // floats per matrix
constexpr size_t kBwdMatLdsBytes = sizeof(float) * (3 * kMatN + 4 * 128 + 8);

// ---- block tables: up to four blocks per wave, one byte each (rb << 2 | cb, 0xFF = none) -------------------------------------
// family L: lower blocks, K range [0, 32 cb + 32)           (tril(X Y^T), X and Y lower triangular, both K-contiguous)
// family R: lower blocks, K range [32 rb, KP)               (T^T X with X lower triangular: k >= row and k >= col)
// family C: lower blocks, K range [32 cb, KP)               (X T with T lower triangular: k >= col)
// family F: all 16 blocks, K range [32 rb, KP)              (T^T S, S full)
enum { kFamL = 0, kFamR = 1, kFamC = 2, kFamF = 3 };
constexpr unsigned kNone = 0xFFu;
__host__ __device__ constexpr unsigned mat_blk(int rb, int cb) { return (unsigned)((rb << 2) | cb); }
__host__ __device__ constexpr unsigned mat_blk4(unsigned a, unsigned b, unsigned c, unsigned d) {
  return a | (b << 8) | (c << 16) | (d << 24);
}
__host__ __device__ constexpr unsigned mat_blocks(int family, int wave) {
  return family == kFamL
             ? (wave == 0   ? mat_blk4(mat_blk(3, 3), mat_blk(0, 0), mat_blk(1, 0), kNone)
                : wave == 1 ? mat_blk4(mat_blk(2, 2), mat_blk(1, 1), kNone, kNone)
                : wave == 2 ? mat_blk4(mat_blk(3, 2), mat_blk(2, 1), kNone, kNone)
                            : mat_blk4(mat_blk(3, 1), mat_blk(2, 0), mat_blk(3, 0), kNone))
         : family == kFamR
             ? (wave == 0   ? mat_blk4(mat_blk(0, 0), mat_blk(3, 0), mat_blk(3, 1), kNone)
                : wave == 1 ? mat_blk4(mat_blk(1, 0), mat_blk(2, 0), kNone, kNone)
                : wave == 2 ? mat_blk4(mat_blk(1, 1), mat_blk(2, 1), kNone, kNone)
                            : mat_blk4(mat_blk(2, 2), mat_blk(3, 2), mat_blk(3, 3), kNone))
         : family == kFamC
             ? (wave == 0   ? mat_blk4(mat_blk(0, 0), mat_blk(1, 1), kNone, kNone)
                : wave == 1 ? mat_blk4(mat_blk(1, 0), mat_blk(2, 1), kNone, kNone)
                : wave == 2 ? mat_blk4(mat_blk(2, 0), mat_blk(3, 1), kNone, kNone)
                            : mat_blk4(mat_blk(3, 0), mat_blk(2, 2), mat_blk(3, 2), mat_blk(3, 3)))
             : (wave == 0   ? mat_blk4(mat_blk(0, 0), mat_blk(0, 1), mat_blk(3, 0), mat_blk(3, 1))
                : wave == 1 ? mat_blk4(mat_blk(0, 2), mat_blk(0, 3), mat_blk(3, 2), mat_blk(3, 3))
                : wave == 2 ? mat_blk4(mat_blk(1, 0), mat_blk(1, 1), mat_blk(2, 0), mat_blk(2, 1))
                            : mat_blk4(mat_blk(1, 2), mat_blk(1, 3), mat_blk(2, 2), mat_blk(2, 3)));
}
template <int FAM, int WV, int U> struct MatBlk {
  static constexpr unsigned code = (mat_blocks(FAM, WV) >> (8 * U)) & 0xFFu;
  static constexpr bool valid = code != kNone;
  static constexpr int rb = valid ? (int)(code >> 2) : 0, cb = valid ? (int)(code & 3u) : 0;
  static constexpr int GE = kBmKP / 8;
  static constexpr int g0 = FAM == kFamL ? 0 : (FAM == kFamC ? 4 * cb : 4 * rb);
  static constexpr int g1 = FAM == kFamL ? bm_min(GE, 4 * cb + 4) : GE;
};

// one 32 x 32 block: acc += A[rows of RB][k] B[k][cols of CB] over the k-groups [G0, G1).  Operand layouts in LDS (stride
// kMatS): KC = [index][k] (one b128 per fragment), KM = [k][index] (four b32).
template <bool AKC, bool BKC, int RB, int CB, int G0, int G1>
__device__ __forceinline__ void mat_block(bm_f32x16& acc, const float* __restrict__ sA, const float* __restrict__ sB, int li, int lh) {
  const int ia = min(32 * RB + li, kBmKP - 1), ib = min(32 * CB + li, kBmKP - 1);
  const float* pa = AKC ? sA + ia * kMatS + 4 * lh : sA + (4 * lh) * kMatS + ia;
  const float* pb = BKC ? sB + ib * kMatS + 4 * lh : sB + (4 * lh) * kMatS + ib;
  auto fa = [&](int g) { return AKC ? bm_frag_kc(pa, 8 * g) : bm_frag_km(pa, 8 * g, kMatS); };
  auto fb = [&](int g) { return BKC ? bm_frag_kc(pb, 8 * g) : bm_frag_km(pb, 8 * g, kMatS); };
  float4 a = fa(G0), b = fb(G0);
  bm_for<G0, G1>([&](auto gi) {
    constexpr int g = decltype(gi)::value;
    const float4 ca = a, cbv = b;
    if constexpr (g + 1 < G1) { a = fa(g + 1); b = fb(g + 1); }
    bm_mfma4(acc, ca, cbv);
  });
}

// all blocks of wave WV for one product: acc[u] (+)= A B over the family's K ranges
template <bool AKC, bool BKC, int FAM, int WV>
__device__ __forceinline__ void mat_product(bm_f32x16 (&acc)[4], const float* __restrict__ sA, const float* __restrict__ sB,
                                            bool zero, int li, int lh) {
  bm_for<0, 4>([&](auto ui) {
    constexpr int u = decltype(ui)::value;
    using Bk = MatBlk<FAM, WV, u>;
    if constexpr (Bk::valid) {
      if (zero) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
      }
      mat_block<AKC, BKC, Bk::rb, Bk::cb, Bk::g0, Bk::g1>(acc[u], sA, sB, li, lh);
    }
  });
}
// f(u, rb, cb, r, i, j) for every accumulator element of the wave's blocks of a family (i, j: matrix indices of register r)
template <int FAM, int WV, class F>
__device__ __forceinline__ void mat_foreach(int li, int lh, F&& f) {
  bm_for<0, 4>([&](auto ui) {
    constexpr int u = decltype(ui)::value;
    using Bk = MatBlk<FAM, WV, u>;
    if constexpr (Bk::valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) f(u, Bk::rb, Bk::cb, r, 32 * Bk::rb + (r & 3) + 8 * (r >> 2) + 4 * lh, 32 * Bk::cb + li);
    }
  });
}

// registers -> LDS (zero padded), as bm_store_mat with stride kMatS
__device__ __forceinline__ void mat_store(float* __restrict__ dst, const float4 (&src)[kBmNA], int M, int tid) {
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) {
    const int e = tid + 256 * u;
    const int i = e / kBmNQ, j = (e - i * kBmNQ) * 4;
    const float4 v = (i < M && j < M) ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < kBmKP * kBmNQ) *reinterpret_cast<float4*>(&dst[i * kMatS + j]) = v;
  }
}

#ifdef BMAT_STAMPS   // tuning builds only: s_memtime of matrix 0, thread 0 after each phase (tests/native/bm_stamps.py mat)
__device__ unsigned long long g_bmat_stamps[4][24];     // [wave][stamp]
__device__ unsigned long long g_bmat_span[2][4];        // gemm.hip, t0_bwdmat_gemm_kernel
extern "C" void vargp_debug_bmat_span(unsigned long long* out, int reset) {
  if (reset) {
    const unsigned long long init[2][4] = {{~0ull, 0, 0, 0}, {~0ull, 0, 0, 0}};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bmat_span), init, sizeof(init));
  } else {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bmat_span), 64);
  }
}
extern "C" void vargp_debug_bmat_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bmat_stamps), 4 * 192); }
#define BMAT_STAMP(i) do { if ((threadIdx.x & 63) == 0 && id == 0) g_bmat_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
// the S_u role's own stamps (first class, wave 0) go into slots 17.. of wave 3's row, which the K_uu role leaves free
#define BSU_STAMP(i) do { if (threadIdx.x == 0 && id == a.S * a.C) g_bmat_stamps[3][17 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BMAT_STAMP(i) do { } while (0)
#define BSU_STAMP(i) do { } while (0)
#endif

// X[i][j] = X[j][i] = scale (diag delta_ij + acc) from the lower blocks of family FAM (rows / columns >= M: zero).  The direct
// write runs along rows (b32, conflict-free); the mirrored one of an off-diagonal block is one b128 per register quad -- the
// quad's four consecutive i are four consecutive words of row j, and rows are 27 quads apart (odd: conflict-free) -- instead
// of 16 b32 writes down a column, four lanes to a bank.  Diagonal blocks, where only j <= i counts, stay element by element.
template <int FAM, int WV>
__device__ __forceinline__ void mat_write_sym(float* __restrict__ X, const bm_f32x16 (&acc)[4], int M, float scale, float diag,
                                              int li, int lh) {
  bm_for<0, 4>([&](auto ui) {
    constexpr int u = decltype(ui)::value;
    using Bk = MatBlk<FAM, WV, u>;
    if constexpr (Bk::valid) {
      const int j = 32 * Bk::cb + li;
      if constexpr (Bk::rb != Bk::cb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i0 = 32 * Bk::rb + 8 * q + 4 * lh;
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = i0 + t < M ? scale * acc[u][4 * q + t] : 0.f;
          if (i0 < kBmKP) {                                  // (kBmKP % 4 == 0: the whole quad is inside; j <= 95 here)
#pragma unroll
            for (int t = 0; t < 4; ++t) X[(i0 + t) * kMatS + j] = v[t];
            *reinterpret_cast<float4*>(&X[j * kMatS + i0]) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 32 * Bk::rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (i < kBmKP && j <= i) {
            const float v = i < M ? scale * ((i == j ? diag : 0.f) + acc[u][r]) : 0.f;
            X[i * kMatS + j] = v;
            X[j * kMatS + i] = v;
          }
        }
      }
    }
  });
}

// ---- tail shared by both roles: S in X2, T in sT -> gA = T^T S T, symmetric, out[i][j] = 2 kmul[i][j] gA[i][j] (kmul == NULL: gA) ----
template <int WV>
__device__ __forceinline__ void mat_tail(float* __restrict__ sT, float* __restrict__ X1, float* __restrict__ X2,
                                         const float* __restrict__ kmul, float* __restrict__ out, float* __restrict__ rsum_out,
                                         float* __restrict__ tot_out, int M, int id, int tid, int lane, int li, int lh) {
  bm_f32x16 acc[4];
  // the factor the result is multiplied by, row by row (coalesced float4): requested now, used in the last pass
  float4 kr[kBmNA];
  if (kmul) bm_load_mat(kmul, M, M, tid, kr);
  mat_product<false, true, kFamF, WV>(acc, sT, X2, true, li, lh);                  // tmp = T^T S  (S symmetric: rows of X2 as the K-contiguous operand, one b128 per fragment)
  BMAT_STAMP(12);
  __syncthreads();                                     // (X1 may still be read by a slower wave's previous product)
  mat_foreach<kFamF, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
    if (i < kBmKP && j < kBmKP) X1[i * kMatS + j] = i < M ? acc[u][r] : 0.f;
  });
  __syncthreads();
  BMAT_STAMP(13);
  mat_product<true, false, kFamC, WV>(acc, X1, sT, true, li, lh);                  // gA = tmp T, lower blocks
  BMAT_STAMP(14);
  // (X2 = S was last read by the product before the previous barrier)
  mat_write_sym<kFamC, WV>(X2, acc, M, 1.f, 0.f, li, lh);
  __syncthreads();
  BMAT_STAMP(15);
  // rows out (coalesced float4); row sums: thread (i, h) = (tid / 2, tid % 2) sums half a row.  (LDS reads first, on clamped
  // indices, all in flight: inside the bounds branch each one is waited for.)
  float dsum = 0.f;
  float4 xv[kBmNA];
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) {
    const int e = min(tid + 256 * u, kBmKP * kBmNQ - 1);
    const int i = e / kBmNQ, j = (e - i * kBmNQ) * 4;
    xv[u] = *reinterpret_cast<const float4*>(&X2[i * kMatS + j]);
  }
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) {
    const int e = tid + 256 * u;
    const int i = e / kBmNQ, j = (e - i * kBmNQ) * 4;
    float4 v = xv[u];
    const bool ok = e < kBmKP * kBmNQ && i < M && j < M;
    if (kmul) {
      v.x *= 2.f * kr[u].x; v.y *= 2.f * kr[u].y; v.z *= 2.f * kr[u].z; v.w *= 2.f * kr[u].w;
      // the diagonal of W_uu counts for gamma only (K_ii = gamma^2: see rbf_w_self_kernel, rbf.hip): out of W_uu and r_uu
      const int q = i - j;
      const float dv = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
      dsum += (ok && q >= 0 && q < 4) ? dv : 0.f;
      v.x = q == 0 ? 0.f : v.x; v.y = q == 1 ? 0.f : v.y; v.z = q == 2 ? 0.f : v.z; v.w = q == 3 ? 0.f : v.w;
      if (ok) *reinterpret_cast<float4*>(&X2[i * kMatS + j]) = v;       // (for the row sums below; each thread its own elements)
    }
    if (ok) *reinterpret_cast<float4*>(&out[(int64_t)i * M + j]) = v;
  }
  if (rsum_out) {
    __syncthreads();
    const int i = tid >> 1, h = tid & 1;
    const float* pr = X2 + min(i, kBmKP - 1) * kMatS + 52 * h;       // columns [0, 52) / [52, 104): the padding is zero
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < 13; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(pr + 4 * q);
      a0 += v.x + v.z; a1 += v.y + v.w;
    }
    float t = a0 + a1;
    t += __shfl_xor(t, 1, 64);
    if (h == 0 && i < M) rsum_out[(int64_t)id * M + i] = t;
    float tot = ((h == 0 && i < M) ? t : 0.f) + dsum;
    tot = wave_sum(tot);
    if (lane == 0) atomicAdd(tot_out, tot);
  }
  BMAT_STAMP(16);
}

// K_uu role of wave WV
template <int WV>
__device__ __forceinline__ void mat_kuu(const BwdMatArgs& a, int id, float* __restrict__ lds, int tid, int lane, int li, int lh) {
  float* sT = lds;                  // T, row-major
  float* X1 = sT + kMatN;
  float* X2 = X1 + kMatN;
  float* sga = X2 + kMatN;          // [128]  ga
  float* smv = sga + 128;           // [128]  m
  float* srs = smv + 128;           // [256]  scratch
  const int M = a.M, LD = a.LD;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  const int64_t b = id;
  const int c = id % a.C;
  const float* Tb = a.TT + b * MM;
  const float* gq = a.gQP + b * MLD;
  const float* qp = a.QP + b * MLD;
  const float* rk = a.RK + b * MLD;
  const float g = a.seeds[1] / (float)a.S;
  bm_f32x16 acc[4];
  float4 r0[kBmNA], r1[kBmNA], r2[kBmNA];
  BMAT_STAMP(0);
  // ---- gG, L_S -> LDS for the first product; T is only needed after it and lands while it runs; gG2, Lu (second stage) and the
  //      tiles' gT are requested right behind
  bm_load_mat(gq + 4, LD, M, tid, r1);
  bm_load_mat(rk + 4, LD, M, tid, r2);
  bm_load_mat(Tb, M, M, tid, r0);
  // ga = the tiles' sum (atomics of t0_bwd_mid_kernel into a cleared column) + the KL's share g a
  const float gav = tid < 128 ? gq[(int64_t)min(tid, M - 1) * LD] + g * qp[(int64_t)min(tid, M - 1) * LD] : 0.f;
  const float mvv = tid < 128 ? rk[(int64_t)min(tid, M - 1) * LD] : 0.f;
  mat_store(X1, r1, M, tid);
  mat_store(X2, r2, M, tid);
  bm_load_mat(qp + 4 + M, LD, M, tid, r1);             // G2 (gG2 = g G2: KL; G2 = T Lu is lower triangular with stored zeros)
  bm_load_mat(rk + 4 + M, LD, M, tid, r2);             // Lu
  // the tiles' share of gT (accumulated by t0_bwd_mid_kernel's atomics), read in the accumulator layout: needed after two products
  float gtt[4][16];
  {
    const float* gTb = a.gTT + b * MM;
    mat_foreach<kFamL, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
      gtt[u][r] = gTb[(int64_t)min(i, M - 1) * M + min(j, M - 1)];
    });
  }
  __syncthreads();
  BMAT_STAMP(1);
  // ---- gT (small columns), part 1: gG L_S^T --------------------------------------------------------------------------------------
  mat_product<true, true, kFamL, WV>(acc, X1, X2, true, li, lh);
  BMAT_STAMP(2);
  mat_store(sT, r0, M, tid);
  if (tid < 128) { sga[tid] = tid < M ? gav : 0.f; smv[tid] = tid < M ? mvv : 0.f; }
  __syncthreads();                                     // everybody is done with gG and L_S; T, ga, m are in place
  BMAT_STAMP(3);
  if (a.gL_acc) {
    // many hyper-samples (the S_u roles run as a launch of their own, after this one): this sample's share of the gradient of
    // L_S, tril(T_s^T gG_s), added into the class's sum -- the S_u role would otherwise walk all S samples by itself
    bm_f32x16 acx[4];
    mat_product<false, false, kFamR, WV>(acx, sT, X1, true, li, lh);
    float* gl = a.gL_acc + (int64_t)c * MM;
    mat_foreach<kFamR, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
      if (i < M && (rb != cb || j <= i)) atomicAdd(&gl[(int64_t)i * M + j], acx[u][r]);
    });
    __syncthreads();                                   // everybody is done with gG (X1 is overwritten next)
  }
#pragma unroll
  for (int u = 0; u < kBmNA; ++u) { r1[u].x *= g; r1[u].y *= g; r1[u].z *= g; r1[u].w *= g; }
  mat_store(X1, r1, M, tid);
  mat_store(X2, r2, M, tid);
  {
    // g_u_mean[c] += T^T ga.  (T^T ga)[i] = sum_{k >= i} T[k][i] ga[k]: thread (i, h) takes the k of half h.  Fixed trip count,
    // every LDS read unconditional and in flight together (a run-time `for k = i..M` waited for each read: 2.3 us for this
    // mat-vec); rows k >= M of T and of ga are zero
    const int i = tid & 127, h = tid >> 7, ic = min(i, kBmKP - 1);
    constexpr int KH = kBmKP / 2;
    const float* tp = sT + (KH * h) * kMatS + ic;
    float tv[KH];
    float4 gk[KH / 4];
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) gk[q] = *reinterpret_cast<const float4*>(&sga[KH * h + 4 * q]);
#pragma unroll
    for (int k = 0; k < KH; ++k) tv[k] = tp[k * kMatS];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const int k0 = KH * h + 4 * q;
      s0 = fmaf(k0 >= i ? tv[4 * q] : 0.f, gk[q].x, s0);
      s1 = fmaf(k0 + 1 >= i ? tv[4 * q + 1] : 0.f, gk[q].y, s1);
      s0 = fmaf(k0 + 2 >= i ? tv[4 * q + 2] : 0.f, gk[q].z, s0);
      s1 = fmaf(k0 + 3 >= i ? tv[4 * q + 3] : 0.f, gk[q].w, s1);
    }
    srs[tid] = s0 + s1;
  }
  __syncthreads();
  BMAT_STAMP(5);
  if (tid < M) atomicAdd(&a.g_u_mean[(int64_t)c * M + tid], srs[tid] + srs[tid + 128]);
  BMAT_STAMP(6);
  // ---- part 2: + gG2 Lu^T;  gLu_part[s, c] = tril(T^T gG2) --------------------------------------------------------------------------------
  mat_product<true, true, kFamL, WV>(acc, X1, X2, false, li, lh);
  {
    bm_f32x16 acr[4];
    mat_product<false, false, kFamR, WV>(acr, sT, X1, true, li, lh);
    // this (s, c)'s share, plain stores (lower triangle; lanes run along a row): the consumers add the S shares of a class.
    // As float atomics into a per-class sum -- 5050 per matrix, three matrices per address -- they drained for 7.5 us AFTER
    // the last wave had retired (kernel 47.2 -> 39.7 us without them, chains alone)
    float* gl = a.gLu_part + b * MM;
    mat_foreach<kFamR, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
      if (i < M && (rb != cb || j <= i)) gl[(int64_t)i * M + j] = acr[u][r];
    });
  }
  BMAT_STAMP(7);
  __syncthreads();                                     // everybody is done with gG2 and Lu
  BMAT_STAMP(8);
  // ---- gT = tiles + tril(ga m^T + gG L_S^T + gG2 Lu^T) -> X1 (K-contiguous operand of w1 = gT T^T) ------------------------------
  // (ga and m read first, unconditionally -- i, j < 128 = their padded lengths: a read inside the bounds branch is waited for
  // element by element)
  float gai[4][16], mj[4];
  mat_foreach<kFamL, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
    gai[u][r] = sga[i];
    if (r == 0) mj[u] = smv[j];
  });
  mat_foreach<kFamL, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
    const float v = acc[u][r] + gtt[u][r] + gai[u][r] * mj[u];
    if (i < kBmKP && j < kBmKP) X1[i * kMatS + j] = (i < M && j <= i) ? v : 0.f;
  });
  __syncthreads();
  BMAT_STAMP(9);
  // ---- w1 = tril(gT T^T);  S = sym(0.5 (g I - w1)) -> X2 --------------------------------------------------------------------------
  mat_product<true, true, kFamL, WV>(acc, X1, sT, true, li, lh);
  BMAT_STAMP(10);
  mat_write_sym<kFamL, WV>(X2, acc, M, -0.5f, -g, li, lh);       // -0.5 (-g delta + w1) = 0.5 (g delta - w1)
  __syncthreads();
  BMAT_STAMP(11);
  mat_tail<WV>(sT, X1, X2, a.KS + b * MM, a.Wuu + b * MM, a.r_uu, &a.gtheta[(int64_t)(id / a.C) * (a.D + 1) + a.D], M, id, tid,
               lane, li, lh);
}

// S_u role of wave WV (class c): its own T_s^T gG_s for every hyper-sample, then the Cholesky adjoint of L_S
template <int WV>
__device__ __forceinline__ void mat_su(const BwdMatArgs& a, int id, float* __restrict__ lds, int tid, int lane, int li, int lh) {
  float* sT = lds;
  float* X1 = sT + kMatN;
  float* X2 = X1 + kMatN;
  const int M = a.M, LD = a.LD, SC = a.S * a.C;
  const int64_t MM = (int64_t)M * M, MLD = (int64_t)M * LD;
  const int c = id - SC;
  bm_f32x16 acc[4];
  float4 r0[kBmNA], r1[kBmNA];
  // gL = sum_s tril(T_s^T gG_s): lower blocks, accumulated over the samples in registers.  The next sample's operands (after
  // the last one: T_S and L_S) are requested before the current product, so that only the first round trip is exposed
  BSU_STAMP(0);
  if (a.gL_acc) {
    // the sum over the samples was accumulated by the K_uu roles of an earlier launch
    bm_load_mat(a.TT + (int64_t)id * MM, M, M, tid, r0);
    bm_load_mat(a.LL + (int64_t)id * MM, M, M, tid, r1);
    const float* gl = a.gL_acc + (int64_t)c * MM;
    mat_foreach<kFamR, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
      acc[u][r] = gl[(int64_t)min(i, M - 1) * M + min(j, M - 1)];
    });
  } else {
  bm_load_mat(a.TT + (int64_t)c * MM, M, M, tid, r0);
  bm_load_mat(a.gQP + (int64_t)c * MLD + 4, LD, M, tid, r1);
  }
  for (int s = 0; s < (a.gL_acc ? 0 : a.S); ++s) {
    if (s > 0) __syncthreads();                        // everybody is done with the previous sample's operands
    mat_store(sT, r0, M, tid);
    mat_store(X1, r1, M, tid);
    __syncthreads();
    const bool more = s + 1 < a.S;
    const int64_t bn = (int64_t)(s + 1) * a.C + c;
    bm_load_mat(a.TT + (more ? bn : (int64_t)id) * MM, M, M, tid, r0);
    bm_load_mat(more ? a.gQP + bn * MLD + 4 : a.LL + (int64_t)id * MM, more ? LD : M, M, tid, r1);
    mat_product<false, false, kFamR, WV>(acc, sT, X1, s == 0, li, lh);
  }
  BSU_STAMP(1);
  __syncthreads();
  // gL (lower, zeros above the diagonal) -> X2 as the k-major operand of P = tril(L_S^T gL)
  mat_foreach<kFamR, WV>(li, lh, [&](int u, int rb, int cb, int r, int i, int j) {
    if (i < kBmKP && j < kBmKP) X2[i * kMatS + j] = (i < M && j <= i) ? acc[u][r] : 0.f;
  });
  mat_store(sT, r0, M, tid);
  mat_store(X1, r1, M, tid);
  __syncthreads();
  BSU_STAMP(2);
  mat_product<false, false, kFamR, WV>(acc, X1, X2, true, li, lh);      // A[i][k] = L_S[k][i] (k >= i), B[k][j] = gL[k][j] (j <= k)
  __syncthreads();                                     // everybody is done with gL before S takes its place
  mat_write_sym<kFamR, WV>(X2, acc, M, 0.5f, 0.f, li, lh);
  __syncthreads();
  BSU_STAMP(3);
  mat_tail<WV>(sT, X1, X2, nullptr, a.gKS + (int64_t)id * MM, nullptr, nullptr, M, id, tid, lane, li, lh);
  BSU_STAMP(4);
}

// matrix id < S C: K_uu role for (s, c) = id;  id >= S C: S_u role for class id - S C
__device__ __forceinline__ void t0_bwd_mat_body(const BwdMatArgs& a, int id, float* __restrict__ lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  if (id < a.S * a.C) {
    if (wave == 0) mat_kuu<0>(a, id, lds, tid, lane, li, lh);
    else if (wave == 1) mat_kuu<1>(a, id, lds, tid, lane, li, lh);
    else if (wave == 2) mat_kuu<2>(a, id, lds, tid, lane, li, lh);
    else mat_kuu<3>(a, id, lds, tid, lane, li, lh);
  } else {
    if (wave == 0) mat_su<0>(a, id, lds, tid, lane, li, lh);
    else if (wave == 1) mat_su<1>(a, id, lds, tid, lane, li, lh);
    else if (wave == 2) mat_su<2>(a, id, lds, tid, lane, li, lh);
    else mat_su<3>(a, id, lds, tid, lane, li, lh);
  }
}

}  // namespace vargp
