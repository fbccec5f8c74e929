// Blocked Cholesky + inverse factor of ONE matrix (64 < n <= 100... 112) by one workgroup of four waves, fp32, on the f32 MFMA
// 16x16x4: the elimination of chol_small3.h (25 block steps of four pivots: a barrier, an LDS round trip, a 4 x 4 in-block
// elimination that every wave repeats and the wave's share of a rank-4 update on the vector units, ~2100 cycles each) re-cut
// into SEVEN steps of sixteen pivots whose panel / trailing / inverse updates are 16 x 16 x 16 block products on the matrix core.
// Reference: gp_utils.py:5-11 (torch.cholesky of K + eps I in fp32) and every triangular solve against it (T = L^-1).
//
// Right-looking on 16 x 16 blocks, L = chol(A), T = L^-1 by forward substitution on X (X = I at the start):
//   step k:  L_kk = chol(A_kk), W = L_kk^-1                (in-wave elimination on the 16 x 16 block, columns across lanes)
//            L_ik = A_ik W^T            (i > k)             panel
//            A_ij -= L_ik L_jk^T        (i >= j > k)        trailing update
//            T_kj = W X_kj  (j < k),  T_kk = W;   X_ij -= L_ik T_kj   (i > k, j <= k)      block row k of the inverse
// What makes it cheap is the register layout.  An accumulator block C of the 16x16x4 MFMA (lane (c, q) = (lane & 15, lane >> 4),
// register r: C[4 q + r][c]) IS a B operand as it stands (k-major: MFMA t takes k = 4 q + t from register t) and, dumped lane by
// lane to LDS and read back the same way (one b128 each, conflict-free), an A operand holding its TRANSPOSE -- as long as both
// operands of a product use that k permutation.  So the matrix is kept as the transposed upper blocks U_ki = A_ik^T (k <= i),
// block column i in one wave:
//            R_ki  = W U_ki                                 = L_ik^T         (A = W from LDS, B = own block)
//            U_ji -= R_kj^T R_ki                                            (A = the dump of R_kj, B = own block R_ki)
// and the inverse as block columns of X / T (column j in one wave):
//            T_kj  = W X_kj,   X_ij -= R_ki^T T_kj                           (A = W / the dump of R_ki, B = own block)
// -- one dump of the panel blocks per step feeds every product of the step, nothing else moves.  Two LDS-only barriers per step.
// Block columns per wave: U {6} {5, 0} {4, 1} {3, 2}, X {0} {1, 6} {2, 5} {3, 4}: seven blocks of each per wave (56 registers).
#pragma once
#include "common.h"
#include <type_traits>

namespace vargp {

#ifdef VARGP_CB16_STAMPS   // tuning builds: shader-clock stamps of workgroup 0, per wave and step: S1 start, behind B1, behind B2, S3 done
__device__ unsigned long long g_cb16_stamps[4][8][4];
extern "C" void vargp_debug_cb16_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cb16_stamps), sizeof(g_cb16_stamps)); }
#define CB_STAMP(step, i) do { if (lane == 0 && blockIdx.x == 0) g_cb16_stamps[WV][step][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CB_STAMP(step, i) do { } while (0)
#endif
typedef float cb_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCbNB = 7;                      // 16-row blocks (112 rows)
// exchange areas inside the caller's staging matrix (floats); all dead before the results are written there
constexpr int kCbRD = 0;                      // [7][256]  panel blocks R_k,j of the current step, dumped lane by lane
constexpr int kCbWB = kCbRD + kCbNB * 256;    // [16][20]  E[i][c] = entry (i, c) of the eliminated diagonal block / sqrt(d_i): W below the diagonal,
                                              //           L_kk^T above it (the diagonal entry itself is not used)
constexpr int kCbDD = kCbWB + 16 * 20;        // [256]     the diagonal block on its way into the factorising wave's column layout
constexpr int kCbED = kCbDD + 256;            // [16]      1 / sqrt(d_i) = W_ii = 1 / L_ii
constexpr int kCbLdsFloats = kCbED + 16;      // (the failure flag -- first failing pivot + 1 -- is a word of the caller's)

template <int I, int N, class F>
__device__ __forceinline__ void cb_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cb_for<I + 1, N>(f); }
}
__host__ __device__ constexpr int cb_owner(int c) { return c >= 3 ? 6 - c : c + 1; }      // wave that holds block column c of U
template <int WV> struct CbOwn {
  static constexpr int UA = 6 - WV, UB = WV > 0 ? WV - 1 : -1;      // block columns of U
  static constexpr int XA = WV, XB = WV > 0 ? 7 - WV : -1;          // block columns of X / T
  __host__ __device__ static constexpr bool owns_u(int c) { return c == UA || c == UB; }
  __host__ __device__ static constexpr bool owns_x(int c) { return c == XA || c == XB; }
  __host__ __device__ static constexpr int ui(int c, int k) { return c == UA ? k : (UA + 1) + k; }                  // block (k, c) of U, k <= c
  __host__ __device__ static constexpr int xi(int c, int i) { return c == XA ? i - XA : (7 - XA) + (i - XB); }      // block (i, c) of X, i >= c
};

__device__ __forceinline__ void cb_mfma4(cb_f32x4& acc, const float4 a, const cb_f32x4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
}
__device__ __forceinline__ void cb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float cb_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One wave's part.  stage: the n x n matrix (row stride LS, both triangles, no jitter) on entry -- every wave has loaded its
// blocks before the first barrier -- and the exchange areas afterwards; on exit (fail == 0) the factors in the convention of
// chol3_body's output loop: stage[r * LS + c] = L[r][c] for r > c and T[c][r] for r < c, sq[i] = L_ii, sd[i] = 1 / L_ii.
template <int WV>
__device__ __forceinline__ void cb16_wave(float* __restrict__ stage, const int LS, double* __restrict__ sq, double* __restrict__ sd,
                                          const int n, const float eps, const int lane, int* __restrict__ flag, float* __restrict__ dump, int& fail) {
  using O = CbOwn<WV>;
  const int c = lane & 15, q = lane >> 4;
  cb_f32x4 accU[kCbNB], accX[kCbNB];
  // ---- the wave's blocks of the (symmetric) matrix: U_ki = block (k, i), rows / columns >= n an identity, eps on the diagonal
  cb_for<0, kCbNB>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    if constexpr (O::owns_u(i)) {
      cb_for<0, i + 1>([&](auto ki) {
        constexpr int k = decltype(ki)::value;
        const int col = 16 * i + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * k + 4 * q + r;
          const int rc = min(row, n - 1), cc = min(col, n - 1);
          const float sv = stage[max(rc, cc) * LS + min(rc, cc)];          // (the lower triangle is the trusted one)
          accU[O::ui(i, k)][r] = (row < n && col < n) ? sv + (row == col ? eps : 0.f) : (row == col ? 1.f : 0.f);
        }
      });
    }
  });
#pragma unroll
  for (int u = 0; u < kCbNB; ++u) accX[u] = cb_f32x4{0.f, 0.f, 0.f, 0.f};
  if (WV == 0 && lane == 0) flag[0] = 0;
  cb_barrier();                                          // everybody has its blocks: `stage` turns into the exchange areas
  float* RD = stage + kCbRD;
  float* WB = stage + kCbWB;
  float* DD = stage + kCbDD;
  float* ED = stage + kCbED;
  const int nblk = (n + 15) >> 4;

  cb_for<0, kCbNB>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    if (k >= nblk || fail) return;                       // (uniform) beyond the matrix: identity blocks, nothing to do
    CB_STAMP(k, 0);
    // ---- S1: the owner of block column k factorises the diagonal block: L_kk, W = L_kk^-1 ---------------------------------------
    if constexpr (cb_owner(k) == WV) {
      // the block from accumulator layout (lane (c, q) holds rows 4 q .. 4 q + 3 of column c) to a column per lane (16 registers;
      // the four lane groups hold copies): through LDS, this wave only
      *reinterpret_cast<cb_f32x4*>(&DD[lane * 4]) = accU[O::ui(k, k)];
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(&DD[(g * 16 + c) * 4]);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
      }
      // forward elimination in place (chol_small3.h's scheme on a 16 x 16 block): entry (i, e) ends as  e < i: W_ie sqrt(d_i),
      // e == i: d_i,  e > i: L_ei sqrt(d_i);  the multiplier of row i for pivot j is p_i by symmetry (lane i of row j).
      // This is the serial part of a step (the other three waves wait at B1): nothing but the pivots' dependent chain and
      // one readlane + one FMA per remaining row in it; lane c keeps the pivot of ITS row (dmine) for the scaling below
      float dmine = 1.f;
      cb_for<0, 16>([&](auto ji) {
        constexpr int j = decltype(ji)::value;
        const float d = cb_lane(v[j], j);
        dmine = c == j ? d : dmine;
        float x = __builtin_amdgcn_rcpf(d);
        x = fmaf(x, fmaf(-d, x, 1.f), x);
        const float ndi = -x;
        float qv = v[j] * ndi;
        qv = c == j ? ndi * (1.f + d) : qv;              // column j counts as 1 + d: the inverse's entry restarts as the multiplier
        cb_for<j + 1, 16>([&](auto ri) {
          constexpr int r = decltype(ri)::value;
          const float m = cb_lane(v[j], r);
          v[r] = fmaf(m, qv, v[r]);
        });
      });
      // 1 / sqrt(d) per lane (its own row's pivot), rows scaled by their pivot's: E[i][c] = entry (i, c) / sqrt(d_i) -- W below the
      // diagonal, L_kk^T above it; the readers mask the triangle they want and take the diagonals from ED = 1 / sqrt(d)
      float rsm = __builtin_amdgcn_rsqf(dmine);
      rsm = rsm * fmaf(-0.5f * dmine * rsm, rsm, 1.5f);
      const bool good = dmine > 0.f;                     // (NaN: false)
      if (lane < 16) {
        ED[c] = rsm;
        if (16 * k + c < n) { sq[16 * k + c] = (double)(dmine * rsm); sd[16 * k + c] = (double)rsm; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float ri = cb_lane(rsm, i);
        if (lane < 16) WB[i * 20 + c] = v[i] * ri;
      }
      const unsigned long long okm = __ballot(good || lane >= 16);
      if (okm != ~0ull && lane == 0) flag[0] = 16 * k + (int)__builtin_ctzll(~okm) + 1;      // first failing pivot + 1
    }
    cb_barrier();                                        // B1: W (and a failure) is out
    fail = flag[0];
    if (fail) return;                                    // (uniform)
    CB_STAMP(k, 1);
    // ---- S2: panel R_ki = W U_ki (dumped for everybody), block row k of the inverse T_kj = W X_kj ---------------------------------
    // A fragment of W: row c, k = 4 q .. 4 q + 3 (lower triangle of E, 1 / sqrt(d) on the diagonal)
    float4 wf = *reinterpret_cast<const float4*>(&WB[c * 20 + 4 * q]);
    {
      const float dg = ED[c];
      const int k0 = 4 * q;
      wf.x = k0 < c ? wf.x : (k0 == c ? dg : 0.f);
      wf.y = k0 + 1 < c ? wf.y : (k0 + 1 == c ? dg : 0.f);
      wf.z = k0 + 2 < c ? wf.z : (k0 + 2 == c ? dg : 0.f);
      wf.w = k0 + 3 < c ? wf.w : (k0 + 3 == c ? dg : 0.f);
    }
    cb_for<k + 1, kCbNB>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      if constexpr (O::owns_u(i)) {
        cb_f32x4 r = {0.f, 0.f, 0.f, 0.f};
        cb_mfma4(r, wf, accU[O::ui(i, k)]);
        accU[O::ui(i, k)] = r;
        *reinterpret_cast<cb_f32x4*>(&RD[i * 256 + lane * 4]) = r;
      }
    });
    if constexpr (O::owns_u(k)) {                        // the diagonal block's place takes R_kk = L_kk^T: [4 q + r][c], upper part of E
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        const float e = WB[i * 20 + c];
        accU[O::ui(k, k)][r] = c > i ? e : 0.f;          // (the diagonal L_ii leaves through sq)
      }
    }
    cb_for<0, k>([&](auto ji) {
      constexpr int j = decltype(ji)::value;
      if constexpr (O::owns_x(j)) {
        cb_f32x4 t = {0.f, 0.f, 0.f, 0.f};
        cb_mfma4(t, wf, accX[O::xi(j, k)]);
        accX[O::xi(j, k)] = t;
      }
    });
    if constexpr (O::owns_x(k)) {                        // T_kk = W: [4 q + r][c], lower part of E with 1 / sqrt(d) on the diagonal
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        const float e = WB[i * 20 + c];
        accX[O::xi(k, k)][r] = c < i ? e : (c == i ? ED[i] : 0.f);
      }
    }
    cb_barrier();                                        // B2: the panel blocks are out
    CB_STAMP(k, 2);
    // ---- S3: trailing update U_ji -= R_kj^T R_ki, inverse X_ij -= R_ki^T T_kj (the next diagonal block's column first) ------------
    cb_for<k + 1, kCbNB>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      if constexpr (O::owns_u(i)) {
        cb_for<k + 1, i + 1>([&](auto ji) {
          constexpr int j = decltype(ji)::value;
          float4 a = *reinterpret_cast<const float4*>(&RD[j * 256 + lane * 4]);
          a = make_float4(-a.x, -a.y, -a.z, -a.w);
          cb_mfma4(accU[O::ui(i, j)], a, accU[O::ui(i, k)]);
        });
      }
    });
    cb_for<0, k + 1>([&](auto ji) {
      constexpr int j = decltype(ji)::value;
      if constexpr (O::owns_x(j)) {
        cb_for<k + 1, kCbNB>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          float4 a = *reinterpret_cast<const float4*>(&RD[i * 256 + lane * 4]);
          a = make_float4(-a.x, -a.y, -a.z, -a.w);
          cb_mfma4(accX[O::xi(j, i)], a, accX[O::xi(j, k)]);
        });
      }
    });
    CB_STAMP(k, 3);
  });
  CB_STAMP(7, 0);
  cb_barrier();                                          // the exchange areas are dead
  if (fail) return;
  float* const mine = dump + lane;                       // where the entries that do not exist go (64 floats nobody reads)
  // ---- the factors into `stage`: L below the diagonal (R_ki[4 q + r][c] = L[16 i + c][16 k + 4 q + r]), T transposed above it
  //      (T_ij[4 q + r][c] = T[16 i + 4 q + r][16 j + c] -> stage[16 j + c][16 i + 4 q + r])
  cb_for<0, kCbNB>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    if constexpr (O::owns_u(i)) {
      cb_for<0, i + 1>([&](auto ki) {
        constexpr int k = decltype(ki)::value;
        const int row = 16 * i + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * k + 4 * q + r;
          *((row < n && col < row) ? &stage[row * LS + col] : mine) = accU[O::ui(i, k)][r];      // (no branches: see cg_store)
        }
      });
    }
    if constexpr (O::owns_x(i)) {                        // (here i is the block COLUMN j of X)
      cb_for<i, kCbNB>([&](auto ri) {
        constexpr int ib = decltype(ri)::value;
        const int col = 16 * i + c;                      // T's column, the staging matrix's row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int trow = 16 * ib + 4 * q + r;          // T's row, the staging matrix's column
          *((trow < n && col < trow) ? &stage[col * LS + trow] : mine) = accX[O::xi(i, ib)][r];
        }
      });
    }
  });
  CB_STAMP(7, 1);
}

// All four waves (256 threads).  Returns with `fail` set for every thread (0: fine) and, on success, a barrier behind the results.
__device__ __forceinline__ void cb16_factor(float* __restrict__ stage, const int LS, double* __restrict__ sq, double* __restrict__ sd,
                                            const int n, const float eps, const int tid, int* __restrict__ flag, float* __restrict__ dump, int& fail) {
  const int wave = tid >> 6, lane = tid & 63;
  fail = 0;
  if (wave == 0) cb16_wave<0>(stage, LS, sq, sd, n, eps, lane, flag, dump, fail);
  else if (wave == 1) cb16_wave<1>(stage, LS, sq, sd, n, eps, lane, flag, dump, fail);
  else if (wave == 2) cb16_wave<2>(stage, LS, sq, sd, n, eps, lane, flag, dump, fail);
  else cb16_wave<3>(stage, LS, sq, sd, n, eps, lane, flag, dump, fail);
  __syncthreads();
}

}  // namespace vargp
