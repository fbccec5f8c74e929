// The last launch of the first-task backward (M <= 104, M % 4 == 0, D % 4 == 0, S <= kTailSMax = 16): the product P_uu = W_uu z of the
// kernel-matrix backward and the finalisation that consumes it, in one kernel -- P_uu never goes to memory.
// Reference: autograd of kernels.py:24-44 (the RBF kernel matrix of scaled inputs) w.r.t. z and the lengthscales; with
// W = gK o K, r = row sums of W, P = W y (see rbf.hip for the algebra):
//   gz[c,i,d]       = -sum_s w_sd ((r_uu z - P_uu) + (r_uf z - P_uf))                (W_uu = W + W^T already)
//   gtheta[s,d]    +=  w_sd sum_{c,i} z ((r_uu z - P_uu) + (r_uf z - 2 P_uf))      (+ the minibatch side: x^2 c_uf)
// Roles by block index:
//   [0, nz)       four waves, each with its own (class, 32-row block, 32-column block of D): the z fragments (the same for every
//                 hyper-sample) stay in registers, the W_uu fragments come straight from memory; f32 MFMA 32x32x2 with the
//                 k-pairing of gemm.hip; the finalisation runs on the accumulators.  No LDS operands, no barriers.
//   next nx       the minibatch side of gtheta (t0_final_x_body, elbo_shared.h)
//   rest          the gradient of the packed Cholesky vector of q(u) (t0_gvec_role), which only shares the launch
#pragma once
#include "elbo_shared.h"
#include "t0_bwd_common.h"

namespace vargp {

constexpr int kTailSMax = 16;
constexpr int kTailXRows = 64;            // minibatch rows per block of the x role
constexpr int kTailNG = kBmKP / 8;        // k-groups of 8

struct TailArgs {
  const float *z, *x, *Wuu, *Puf, *r_uu, *r_uf, *c_uf, *w;
  float *gz, *gtheta;
  int S, C, M, D, B;
  int64_t Dp;
  int nrb, ncb, nz, nx, gx;      // 32 x 32 blocks per class (rows, columns); z-role workgroups; x-role blocks and their grid width
};

#ifdef TAIL_STAMPS   // per-phase cycle accounting (workgroup 0, thread 0), tuning builds only: tests/native/bm_stamps.py tail
__device__ unsigned long long g_tail_stamps[16];
extern "C" void vargp_debug_tail_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), 128); }
#define TAIL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_tail_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void tail_load_a(const float* __restrict__ wp, int lh, int M, float4 (&af)[kTailNG]) {
#pragma unroll
  for (int g = 0; g < kTailNG; ++g) af[g] = *reinterpret_cast<const float4*>(wp + min(8 * g + 4 * lh, M - 4));
}

// (S is a template parameter: the sample loop is unrolled, so that the waits on the prefetched loads are exact -- around a
// runtime loop the compiler waits for every outstanding load before the first MFMA)
template <int S>
__global__ __launch_bounds__(256, 2) void t0_puu_final_kernel(const TailArgs a, const GvecArgs gv) {
  __shared__ __attribute__((aligned(16))) float rsl[4][kTailSMax][32];
  __shared__ float redx[2][4][64];
  int blk = blockIdx.x;
  if (blk >= a.nz) {      // (the short roles last: dispatched first they measured the same)
    blk -= a.nz;
    if (blk < a.nx) t0_final_x_body<kTailXRows>(a.x, a.c_uf, a.w, a.gtheta, (int64_t)a.B, a.D, a.Dp, a.S, blk % a.gx, blk / a.gx, redx);
    else t0_gvec_role(blk - a.nx, gv.vec, gv.Lu, gv.gSu, gv.gLu_acc, gv.seeds, gv.gvec, 1, gv.C, gv.M, gv.M, 0);
    return;
  }
  // one 32 x 32 block of one class per WAVE (flat index: C * ceil(M / 32) * ceil(D / 32) = 1000 blocks = 250 workgroups at the
  // BASELINE shape, i.e. one round even at one workgroup per CU); the waves of a workgroup share nothing
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = a.M, D = a.D;
  const int id = __builtin_amdgcn_readfirstlane(blk * 4 + wave);
  if (id >= a.C * a.nrb * a.ncb) return;
  const int cb = id % a.ncb, rb = (id / a.ncb) % a.nrb, c = id / (a.ncb * a.nrb);
  const int d0 = 32 * cb, r0 = 32 * rb;
  const int64_t zrows = (int64_t)a.C * M;
  const float* zc = a.z + (int64_t)c * M * D;
  TAIL_STAMP(0);
  // --- r_uu + r_uf of the block's 32 rows, every sample -> LDS (this wave's slice) --------------------------------------------
  for (int e = lane; e < 32 * (S == 0 ? a.S : S); e += 64) {
    const int s = e >> 5, row = min(r0 + (e & 31), M - 1);
    const int64_t sr = (int64_t)s * zrows + (int64_t)c * M + row;
    rsl[wave][s][e & 31] = a.r_uu[sr] + a.r_uf[sr];
  }
  // --- this lane's 16 output positions: rows rbase + 8 (r / 4) + r % 4, column d -------------------------------------------
  const int d = d0 + li;
  const bool dok = d < D;
  const int dc = dok ? d : D - 1;
  const int rbase = r0 + 4 * lh;
  auto rowof = [rbase](int r) { return rbase + 8 * (r >> 2) + (r & 3); };
  // B fragments (z[c][k][d], k = 8 g + 4 lh + j): the same for every hyper-sample, so they live in registers -- no LDS
  // panel, no LDS latency inside the MFMA loops (one wave per SIMD: nothing would hide it).  Rows k >= M are only ever
  // multiplied by the zeroed A fragments, columns d >= D are masked in the epilogue: clamped loads, no masks here.
  float4 bf[kTailNG];
#pragma unroll
  for (int g = 0; g < kTailNG; ++g) {
    const float* zp = zc + (int64_t)min(8 * g + 4 * lh, M - 4) * D + dc;
    bf[g] = make_float4(zp[0], zp[D], zp[2 * (int64_t)D], zp[3 * (int64_t)D]);
  }
  // A fragments (rows of W_uu[s, c], K-contiguous) straight from memory: the lane's float4 of k-group g
  const int arow = min(r0 + li, M - 1);
  const bool arow_ok = r0 + li < M;
  const float* wrow = a.Wuu + ((int64_t)c * M + arow) * M;      // + s * C * M * M
  const int64_t wstep = (int64_t)a.C * M * M;
  const float* pufp = a.Puf + ((int64_t)c * M) * D + dc;        // + (s * zrows + row) * D
  float4 af[kTailNG];
  float p2[16], ga[16];
  tail_load_a(wrow, lh, M, af);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t ro = (int64_t)min(rowof(r), M - 1) * D;
    p2[r] = pufp[ro];
    ga[r] = 0.f;
  }
  TAIL_STAMP(1);

  // S == 0: the sample count is a.S and the loop stays a loop (more than four samples: unrolled, the compiler hoists every
  // sample's addresses and spills)
  const int ns = S == 0 ? a.S : S;
  constexpr int kUnroll = S == 0 ? 1 : S;
#pragma unroll kUnroll
  for (int s = 0; s < ns; ++s) {
    const float wv = dok ? a.w[s * a.Dp + dc] : 0.f;
    bm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < kTailNG; ++g) {
      const bool ok = arow_ok && 8 * g + 4 * lh < M;
      const float4 av = ok ? af[g] : make_float4(0.f, 0.f, 0.f, 0.f);
      bm_mfma4(acc, av, bf[g]);
    }
    __builtin_amdgcn_sched_barrier(0);
    TAIL_STAMP(3 + 2 * s);
    if (s + 1 < ns) tail_load_a(wrow + (s + 1) * wstep, lh, M, af);       // next sample's fragments under the epilogue
    float th = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 rs4 = *reinterpret_cast<const float4*>(&rsl[wave][s][4 * lh + 8 * q]);
      const float rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
      // z[rowof(4 q + j)][d] is component j of the B fragment of k-group 4 rb + q (k = row): already in registers
      // (component-wise selects: a select between whole fragments takes the array's address and sends it to scratch)
      const float4 z0 = bf[q], z1 = bf[4 + q], z2 = bf[8 + q], z3 = bf[bm_min(12 + q, kTailNG - 1)];
      const float zq[4] = {rb == 0 ? z0.x : rb == 1 ? z1.x : rb == 2 ? z2.x : z3.x, rb == 0 ? z0.y : rb == 1 ? z1.y : rb == 2 ? z2.y : z3.y,
                           rb == 0 ? z0.z : rb == 1 ? z1.z : rb == 2 ? z2.z : z3.z, rb == 0 ? z0.w : rb == 1 ? z1.w : rb == 2 ? z2.w : z3.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        const bool okr = dok && rowof(r) < M;
        const float q1 = okr ? acc[r] : 0.f, q2 = okr ? p2[r] : 0.f, zr = okr ? zq[j] : 0.f;
        const float t = rs[j] * zr - q1 - q2;            // (r_uu z - P_uu) + (r_uf z - P_uf)
        ga[r] -= wv * t;
        th += zr * (t - q2);
      }
    }
    if (s + 1 < ns) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p2[r] = pufp[((int64_t)(s + 1) * zrows + min(rowof(r), M - 1)) * D];
    }
    th += __shfl_xor(th, 32);
    if (lh == 0 && dok) atomicAdd(&a.gtheta[(int64_t)s * (D + 1) + d], wv * th);
    __builtin_amdgcn_sched_barrier(0);
    TAIL_STAMP(4 + 2 * s);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (dok && rowof(r) < M) a.gz[((int64_t)c * M + rowof(r)) * D + d] = ga[r];
  }
  TAIL_STAMP(13);
}

}  // namespace vargp
