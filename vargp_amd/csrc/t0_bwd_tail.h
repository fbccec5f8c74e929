// The last launch of the first-task backward (M <= 104, M % 4 == 0, D % 4 == 0, S <= kTailSMax = 64): the product P_uu = W_uu z of the
// kernel-matrix backward and the finalisation that consumes it, in one kernel -- P_uu never goes to memory.
// Reference: autograd of kernels.py:24-44 (the RBF kernel matrix of scaled inputs) w.r.t. z and the lengthscales; with
// W = gK o K, r = row sums of W, P = W y (see rbf.hip for the algebra):
//   gz[c,i,d]       = -sum_s w_sd ((r_uu z - P_uu) + (r_uf z - P_uf))                (W_uu = W + W^T already)
//   gtheta[s,d]    +=  w_sd sum_{c,i} z ((r_uu z - P_uu) + (r_uf z - 2 P_uf))      (+ the minibatch side: x^2 c_uf)
// Roles by block index:
//   [0, nz)       four waves, each with its own (class, 32-row block, 32-column block of D): the z fragments (the same for every
//                 hyper-sample) stay in registers, the W_uu fragments come straight from memory; f32 MFMA 32x32x2 with the
//                 k-pairing of gemm.hip; the finalisation runs on the accumulators.  No LDS operands, no barriers.
//   next nrem     the last M % 32 <= 8 rows of every class on the vector units (tail_rem_rows): at M = 100 a quarter of the MFMA
//                 wave-blocks would hold 4 rows of 32
//                 The wave-blocks behind them: the gradient of the packed Cholesky vector of q(u), 2 gS_u Lu as lower 32 x 32
//                 blocks (tail_gvec_block), which only shares the launch.
//   next nx       the minibatch side of gtheta (t0_final_x_body, elbo_shared.h)
#pragma once
#include "elbo_shared.h"
#include "t0_bwd_common.h"

namespace vargp {

constexpr int kTailSMax = 64;
constexpr int kTailXRows = 64;            // minibatch rows per block of the x role
constexpr int kTailNG = kBmKP / 8;        // k-groups of 8
static_assert(kBmKP == 104, "t0_puu_final_lds_kernel pads its staged rows to 104 columns");

struct TailArgs {
  const float *z, *x, *Wuu, *Puf, *r_uu, *r_uf, *c_uf, *w;
  float *gz, *gtheta;
  int S, C, M, D, B;
  int64_t Dp;
  int nrb, ncb, nz, nx, gx;      // 32 x 32 blocks per class (rows, columns); z-role workgroups; x-role blocks and their grid width
  int nrem;                      // remainder-row blocks (rows 32 nrb .. M - 1 when M % 32 <= 8: C x gx x chunks of 4 rows)
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

// Gradient of the packed Cholesky vector of q(u) as 32 x 32 MFMA blocks (one lower block (rb >= cb) of one class per wave):
//   gLu = 2 gS_u Lu + sum_s gRK[Lu block] - seed_kl diag(1 / Lu_ii),   through vec2tril (softplus on the diagonal)
// -- the same arithmetic as t0_gvec_role (elbo_shared.h), whose 391 light workgroups of 100-long dot products per thread load the
// memory pipes the z role's waves wait on.  A = gS_u rows (symmetric, K-contiguous) and B = Lu ([k][column]) straight from
// memory, as in the z role.
__device__ __forceinline__ void tail_gvec_block(const GvecArgs& gv, int c, int rb, int cb, int li, int lh) {
  const int M = gv.M;
  const int64_t MM = (int64_t)M * M;
  const float* gs = gv.gSu + c * MM;
  const float* lu = gv.Lu + c * MM;
  const int arow = min(32 * rb + li, M - 1);
  const bool arow_ok = 32 * rb + li < M;
  const int colc = min(32 * cb + li, M - 1);
  float4 af[kTailNG], bf[kTailNG];
  tail_load_a(gs + (int64_t)arow * M, lh, M, af);
#pragma unroll
  for (int g = 0; g < kTailNG; ++g) {
    const float* p = lu + (int64_t)min(8 * g + 4 * lh, M - 4) * M + colc;
    bf[g] = make_float4(p[0], p[M], p[2 * M], p[3 * M]);
  }
  const int rbase = 32 * rb + 4 * lh, k = 32 * cb + li;
  float ga[16];          // sum over the samples' shares (t0_bwd_mat.h writes one lower triangle per (s, c))
#pragma unroll
  for (int r = 0; r < 16; ++r) ga[r] = 0.f;
  for (int sidx = 0; sidx < gv.S; ++sidx) {
    const float* gp = gv.gLu_part + ((int64_t)sidx * gv.C + c) * MM + colc;
    float gs_[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gs_[r] = gp[(int64_t)min(rbase + 8 * (r >> 2) + (r & 3), M - 1) * M];
#pragma unroll
    for (int r = 0; r < 16; ++r) ga[r] += gs_[r];
  }
  bm_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < kTailNG; ++g) {
    const bool ok = arow_ok && 8 * g + 4 * lh < M;         // (Lu[j][k] = 0 for j < k: the groups below 4 cb only add zeros)
    const float4 av = ok ? af[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    bm_mfma4(acc, av, bf[g]);
  }
  const int64_t vbase = (int64_t)c * ((int64_t)M * (M + 1) / 2);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = rbase + 8 * (r >> 2) + (r & 3);
    if (i < M && k <= i) {          // (k <= i < M)
      float g = 2.f * acc[r] + ga[r];
      const int64_t idx = vbase + (int64_t)i * (i + 1) / 2 + k;
      if (i == k) {
        g -= gv.seeds[1] / lu[(int64_t)i * M + i];
        const float x = gv.vec[idx];
        g *= (x > 20.f) ? 1.f : sigmoid_t0(x);
      }
      gv.gvec[idx] = g;
    }
  }
}

// The last M % 32 <= 8 rows of every class on the vector units, one wave per row and 64 columns of D per block: as a 32 x 32
// MFMA block they would be a quarter of the z role's wave-blocks at M = 100 (rows 96 .. 99 of 128) doing 1/8 of a block's work,
// and push it over one workgroup per CU.  The lane's column of z lives in registers (the same for every sample), the row of
// W_uu is loaded once per sample across the lanes and broadcast by v_readlane.  Same finalisation as the MFMA blocks.
__device__ __forceinline__ void tail_rem_rows(const TailArgs& a, int blk, float (*red)[4][64]) {
  const int M = a.M, D = a.D;
  const int tx = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nchunk = (M - 32 * a.nrb + 3) / 4;
  const int dt = blk % a.gx, ch = (blk / a.gx) % nchunk, c = blk / (a.gx * nchunk);
  const int row = 32 * a.nrb + 4 * ch + wave;
  const bool rok = row < M;
  const int rowc = min(row, M - 1);
  const int d = 64 * dt + tx;
  const bool ok = rok && d < D;
  const int dc = min(d, D - 1);
  const int64_t zrows = (int64_t)a.C * M;
  const float* zc = a.z + (int64_t)c * M * D + dc;
  float zk[kBmKP];
#pragma unroll
  for (int k = 0; k < kBmKP; ++k) zk[k] = zc[(int64_t)min(k, M - 1) * D];      // (k >= M only ever meets a zero of W_uu)
  const float zr = ok ? zc[(int64_t)rowc * D] : 0.f;
  float ga = 0.f;
  for (int s = 0; s < a.S; ++s) {
    const int64_t sr = (int64_t)s * zrows + (int64_t)c * M + rowc;
    const float* wr = a.Wuu + sr * M;
    const float w0 = tx < M ? wr[min(tx, M - 1)] : 0.f, w1 = 64 + tx < M ? wr[min(64 + tx, M - 1)] : 0.f;
    const float p2 = ok ? a.Puf[sr * D + dc] : 0.f, rs = a.r_uu[sr] + a.r_uf[sr], wv = d < D ? a.w[s * a.Dp + dc] : 0.f;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k += 2) {
      acc0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), k)), zk[k], acc0);
      acc1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), k + 1)), zk[k + 1], acc1);
    }
#pragma unroll
    for (int k = 64; k < kBmKP; k += 2) {
      acc0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), k - 64)), zk[k], acc0);
      acc1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), k - 63)), zk[k + 1], acc1);
    }
    const float q1 = ok ? acc0 + acc1 : 0.f;
    const float t = rs * zr - q1 - p2;              // (r_uu z - P_uu) + (r_uf z - P_uf)
    ga -= wv * t;
    red[s & 1][wave][tx] = zr * (t - p2);
    __syncthreads();
    if (wave == 0 && d < D)
      atomicAdd(&a.gtheta[(int64_t)s * (D + 1) + d], wv * (red[s & 1][0][tx] + red[s & 1][1][tx] + red[s & 1][2][tx] + red[s & 1][3][tx]));
  }
  if (ok) a.gz[((int64_t)c * M + row) * D + d] = ga;
}

// (S is a template parameter: the sample loop is unrolled, so that the waits on the prefetched loads are exact -- around a
// runtime loop the compiler waits for every outstanding load before the first MFMA)
template <int S>
__global__ __launch_bounds__(256, 2) void t0_puu_final_kernel(const TailArgs a, const GvecArgs gv) {
  __shared__ __attribute__((aligned(16))) float rsl[4][kTailSMax][32];
  __shared__ float redx[2][4][64];
  STEP_SPAN(t0, 6);
  int blk = blockIdx.x;
  if (blk >= a.nz) {      // (the short roles last: dispatched first they measured the same)
    blk -= a.nz;
    if (blk < a.nrem) { tail_rem_rows(a, blk, redx); return; }
    blk -= a.nrem;
    t0_final_x_body<kTailXRows>(a.x, a.c_uf, a.w, a.gtheta, (int64_t)a.B, a.D, a.Dp, a.S, blk % a.gx, blk / a.gx, redx);
    return;
  }
  // one 32 x 32 block of one class per WAVE (flat index: C * ceil(M / 32) * ceil(D / 32) = 1000 blocks = 250 workgroups at the
  // BASELINE shape, i.e. one round even at one workgroup per CU); the waves of a workgroup share nothing
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = a.M, D = a.D;
  const int id = __builtin_amdgcn_readfirstlane(blk * 4 + wave);
  const int nzb = a.C * a.nrb * a.ncb;
  if (id >= nzb) {      // the wave-blocks behind the z role's: lower blocks of the packed-vector gradient, class by class
    const int ngr = (gv.M + 31) / 32, nlow = ngr * (ngr + 1) / 2, e = id - nzb;
    if (e >= gv.C * nlow) return;
    int q = e % nlow, rbg = 0;
    while (q > rbg) { q -= rbg + 1; ++rbg; }             // q-th lower block: row block rbg, column block q
    tail_gvec_block(gv, e / nlow, rbg, q, li, lh);
    return;
  }
  const int cb = id % a.ncb, rb = (id / a.ncb) % a.nrb, c = id / (a.ncb * a.nrb);
  const int d0 = 32 * cb, r0 = 32 * rb;
  const int64_t zrows = (int64_t)a.C * M;
  const float* zc = a.z + (int64_t)c * M * D;
  TAIL_STAMP(0);
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
  // --- r_uu + r_uf of the block's 32 rows, every sample -> LDS (this wave's slice).  BEHIND the fragment loads: as the first
  // thing in the kernel its load -> add -> LDS-store iterations were two memory round trips before any other load was issued;
  // the sums are only needed by the first sample's epilogue, after the first MFMA group.
  for (int e = lane; e < 32 * (S == 0 ? a.S : S); e += 64) {
    const int s = e >> 5, row = min(r0 + (e & 31), M - 1);
    const int64_t sr = (int64_t)s * zrows + (int64_t)c * M + row;
    rsl[wave][s][e & 31] = a.r_uu[sr] + a.r_uf[sr];
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

// ---- more than four hyper-samples: the z role with its per-sample operands staged through LDS by loader waves ---------------------------
// In the kernel above every wave-block fetches its own fragments of W_uu[s, c] (13 float4 per lane and sample) and its 16 rows of the
// P_uf block (16 dwords per lane and sample) straight from memory, sample after sample: with many samples the launch is bound by those
// loads -- 29 vector-memory instructions per lane, sample and block, and the four waves of a workgroup (four neighbouring column
// blocks of the SAME 32 rows) each pull the same rows of W_uu.  Here a workgroup is (class, 32-row block, group of four 32-column
// blocks) and has EIGHT waves: four compute waves (one 32 x 32 block each: MFMAs + finalisation on operands in LDS) and four loader
// waves that stream the 32 x M rows of W_uu[s, c] (once for the four blocks), the four 32 x 32 blocks of P_uf (float4 rows) and the
// 128 values of 1/sigma_s^2 from memory into double-buffered LDS, two samples ahead in registers of their own.  Why separate waves:
// staged by the compute waves themselves the float4 of the next sample were spilled to scratch by the compiler (i.e. waited for on
// the spot), and vmcnt counts in order, so their one scalar load per sample waited for the whole prefetch.  One LDS-only barrier per
// sample joins the eight waves.  The sums r_uu + r_uf of the 32 rows are shared too.  Roles behind the z groups as in the kernel
// above, on the first four waves.
constexpr int kTailWS = 108;              // row stride of the staged W_uu rows (108 / 4 odd: conflict-free b128 fragments)
constexpr int kTailPS = 40;               // row stride of a staged 32 x 32 block of P_uf (4 rows = 160 words: the two half-waves on disjoint banks)
constexpr int kTailWBuf = 32 * kTailWS, kTailPBuf = 4 * 32 * kTailPS;
constexpr size_t kTailLdsBytes = sizeof(float) * (2 * kTailWBuf + 2 * kTailPBuf + 2 * 128 + kTailSMax * 32 + 2 * 4 * 64);

__global__ __launch_bounds__(512) void t0_puu_final_lds_kernel(const TailArgs a, const GvecArgs gv, const int nzg, const int ngvw) {
  extern __shared__ __attribute__((aligned(16))) float tail_lds[];
  float* wbuf = tail_lds;                               // [2][32][kTailWS]
  float* pbuf = wbuf + 2 * kTailWBuf;                   // [2][4][32][kTailPS]
  float* wtab = pbuf + 2 * kTailPBuf;                   // [2][128]   1/sigma_s^2 of the group's 128 columns
  float* rsl = wtab + 2 * 128;                          // [S][32]
  float (*redx)[4][64] = reinterpret_cast<float (*)[4][64]>(rsl + kTailSMax * 32);
  STEP_SPAN(t0, 6);
  int blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  if (blk >= nzg) {
    if (tid >= 256) return;                             // (a finished wave no longer counts at s_barrier)
    blk -= nzg;
    if (blk < ngvw) {          // lower blocks of the packed-vector gradient, one per wave
      const int ngr = (gv.M + 31) / 32, nlow = ngr * (ngr + 1) / 2, e = blk * 4 + wave;
      if (e >= gv.C * nlow) return;
      int q = e % nlow, rbg = 0;
      while (q > rbg) { q -= rbg + 1; ++rbg; }
      tail_gvec_block(gv, e / nlow, rbg, q, li, lh);
      return;
    }
    blk -= ngvw;
    if (blk < a.nrem) { tail_rem_rows(a, blk, redx); return; }
    blk -= a.nrem;
    t0_final_x_body<kTailXRows>(a.x, a.c_uf, a.w, a.gtheta, (int64_t)a.B, a.D, a.Dp, a.S, blk % a.gx, blk / a.gx, redx);
    return;
  }
  const int M = a.M, D = a.D, S = a.S;
  const int ncg = (a.ncb + 3) >> 2;
  const int cg = blk % ncg, rb = (blk / ncg) % a.nrb, c = blk / (ncg * a.nrb);
  const int r0 = 32 * rb;
  const int64_t zrows = (int64_t)a.C * M;
  if (tid >= 256) {
    // ---------------- loader waves: sample s + 1 -> the idle LDS buffer during sample s, from loads issued two samples earlier --------
    const int t = tid - 256;
    const int nq = M >> 2;
    int wrow[4], wq[4];
    bool wok[4];
    const float* wptr[4];
    const float* pptr[4];
    const float* wsrc = a.Wuu + ((int64_t)c * M) * M;
    const int prow = t >> 3, pq = (t & 7) << 2;
    const float* psrc = a.Puf + ((int64_t)c * M + min(r0 + prow, M - 1)) * D;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      const int row = e / nq;
      wq[u] = (e - row * nq) << 2;
      wok[u] = e < 32 * nq;
      wrow[u] = min(row, 31);
      wptr[u] = wsrc + (int64_t)min(r0 + wrow[u], M - 1) * M + wq[u];
      pptr[u] = psrc + min(32 * (4 * cg + u) + pq, D - 4);
    }
    const int64_t wstep = (int64_t)a.C * M * M, pstep = zrows * D;
    const int wcol = min(128 * cg + (t & 127), D - 1);
    float4 rwA[4], rpA[4], rwB[4], rpB[4];
    float wvA, wvB;
    auto load_sample = [&](int s, float4 (&rw)[4], float4 (&rp)[4], float& wv) {
      const int sc = min(s, S - 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) rw[u] = *reinterpret_cast<const float4*>(wptr[u] + sc * wstep);
#pragma unroll
      for (int u = 0; u < 4; ++u) rp[u] = *reinterpret_cast<const float4*>(pptr[u] + sc * pstep);
      wv = a.w[sc * a.Dp + wcol];
    };
    auto store_sample = [&](int buf, const float4 (&rw)[4], const float4 (&rp)[4], const float wv) {
      float* wb = wbuf + buf * kTailWBuf;
      float* pb = pbuf + buf * kTailPBuf;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = r0 + wrow[u] < M ? rw[u] : make_float4(0.f, 0.f, 0.f, 0.f);       // rows past M: zeros
        if (wok[u]) *reinterpret_cast<float4*>(&wb[wrow[u] * kTailWS + wq[u]]) = v;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(&pb[(u * 32 + prow) * kTailPS + pq]) = rp[u];
      if (t < 128) wtab[buf * 128 + t] = wv;
    };
    load_sample(0, rwA, rpA, wvA);
    load_sample(1, rwB, rpB, wvB);
    // columns M .. 107 of the staged rows are only ever multiplied: zeros, written once (the staging never touches them)
    {
      const int npad = kTailWS - M;
      for (int e = t; e < 2 * 32 * npad; e += 256) {
        const int b2 = e / (32 * npad), r = (e / npad) & 31, k = M + e % npad;
        wbuf[b2 * kTailWBuf + r * kTailWS + k] = 0.f;
      }
    }
    for (int e = t; e < 32 * S; e += 256) {
      const int s = e >> 5, row = min(r0 + (e & 31), M - 1);
      const int64_t sr = (int64_t)s * zrows + (int64_t)c * M + row;
      rsl[e] = a.r_uu[sr] + a.r_uf[sr];
    }
    store_sample(0, rwA, rpA, wvA);
    load_sample(2, rwA, rpA, wvA);
    bmm_lds_barrier();                                  // sample 0 is in buffer 0
    for (int s = 0; s < S; s += 2) {
      if (s + 1 < S) store_sample(1, rwB, rpB, wvB);    // sample s + 1 (set B, requested two samples ago)
      load_sample(s + 3, rwB, rpB, wvB);
      bmm_lds_barrier();                                // end of sample s
      if (s + 1 < S) {
        if (s + 2 < S) store_sample(0, rwA, rpA, wvA);  // sample s + 2
        load_sample(s + 4, rwA, rpA, wvA);
        bmm_lds_barrier();                              // end of sample s + 1
      }
    }
    return;
  }
  // ---------------- compute waves ------------------------------------------------------------------------------------------------
  const int cb = 4 * cg + wave;
  const bool active = cb < a.ncb;                       // (wave-uniform; an idle wave still meets the barriers)
  const int d0 = 32 * cb;
  const float* zc = a.z + (int64_t)c * M * D;
  const int d = d0 + li;
  const bool dok = active && d < D;
  const int dc = min(d, D - 1);
  const int rbase = 4 * lh;                             // row of register r inside the block: rbase + 8 (r / 4) + r % 4
  auto rowl = [rbase](int r) { return rbase + 8 * (r >> 2) + (r & 3); };
  // B fragments (z[c][k][d]): the same for every sample, in registers
  float4 bf[kTailNG];
#pragma unroll
  for (int g = 0; g < kTailNG; ++g) {
    const float* zp = zc + (int64_t)min(8 * g + 4 * lh, M - 4) * D + dc;
    bf[g] = make_float4(zp[0], zp[D], zp[2 * (int64_t)D], zp[3 * (int64_t)D]);
  }
  float ga[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ga[r] = 0.f;
  // z[row of register r][d] (component j of the B fragment of k-group 4 rb + q, r = 4 q + j; zero where the row or column does
  // not exist) and the mask itself: the same for every sample
  float zsel[16];
  bool zok[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 z0 = bf[q], z1 = bf[4 + q], z2 = bf[8 + q], z3 = bf[bm_min(12 + q, kTailNG - 1)];
    const float zq[4] = {rb == 0 ? z0.x : rb == 1 ? z1.x : rb == 2 ? z2.x : z3.x, rb == 0 ? z0.y : rb == 1 ? z1.y : rb == 2 ? z2.y : z3.y,
                         rb == 0 ? z0.z : rb == 1 ? z1.z : rb == 2 ? z2.z : z3.z, rb == 0 ? z0.w : rb == 1 ? z1.w : rb == 2 ? z2.w : z3.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      zok[4 * q + j] = dok && r0 + rowl(4 * q + j) < M;
      zsel[4 * q + j] = zok[4 * q + j] ? zq[j] : 0.f;
    }
  }
  bmm_lds_barrier();                                    // sample 0 is in buffer 0 (and the row sums, the zero padding)
  for (int s = 0; s < S; ++s) {
    const int bsel = s & 1;
    if (active) {
      const float* wb = wbuf + bsel * kTailWBuf + li * kTailWS + 4 * lh;
      const float* pb = pbuf + bsel * kTailPBuf + wave * 32 * kTailPS + li;
      const float wv = dok ? wtab[bsel * 128 + wave * 32 + li] : 0.f;
      bm_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float4 av = *reinterpret_cast<const float4*>(wb);
#pragma unroll
      for (int g = 0; g < kTailNG; ++g) {
        const float4 cur = av;
        if (g + 1 < kTailNG) av = *reinterpret_cast<const float4*>(wb + 8 * (g + 1));
        bm_mfma4(acc, cur, bf[g]);
      }
      float p2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) p2[r] = pb[rowl(r) * kTailPS];
      float th = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 rs4 = *reinterpret_cast<const float4*>(&rsl[s * 32 + 4 * lh + 8 * q]);
        const float rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * q + j;
          const bool okr = zok[r];
          const float q1 = okr ? acc[r] : 0.f, q2 = okr ? p2[r] : 0.f, zr = zsel[r];
          const float t = rs[j] * zr - q1 - q2;            // (r_uu z - P_uu) + (r_uf z - P_uf)
          ga[r] -= wv * t;
          th += zr * (t - q2);
        }
      }
      th += __shfl_xor(th, 32);
      if (lh == 0 && dok) atomicAdd(&a.gtheta[(int64_t)s * (D + 1) + d], wv * th);
    }
    bmm_lds_barrier();                                  // end of sample s: the loaders have filled the other buffer
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (dok && r0 + rowl(r) < M) a.gz[((int64_t)c * M + r0 + rowl(r)) * D + d] = ga[r];
    }
  }
}

}  // namespace vargp
