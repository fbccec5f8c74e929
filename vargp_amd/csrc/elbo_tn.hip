// The ELBO of a model WITH previous tasks (t > 0; also any first-task model, nblk = 1) as ONE native program:
// vargp_elbo_tn_fwd / vargp_elbo_tn_bwd (reference: VARGP.compute_q / compute_pf_diag / forward / loss,
// var_gp/vargp.py:35-194; linear_joint / linear_marginal_diag / gp_cond, var_gp/gp_utils.py:68-191).
//
// The reference folds the earlier tasks into q(u_<=t | theta) with a chain of linear_joint calls (one Cholesky, four
// triangular solves and three GEMMs per earlier task, on joint covariances that grow to Mt x Mt), then factorises
// K_uu + eps I and S_<=t + eps I again for the predictive moments, then the conditional prior for the KL.  All of it
// is a function of ONE kernel matrix.  With K' = K(z_<=t, z_<=t) + eps I, L = chol(K'), T = L^-1 and blocks of size M in
// task order (tests/block_algorithm.py pins every identity against the oracle in fp64):
//   * every Lz of the chain is a leading block of L, and A_i = K_{i,<i} (K_{<i,<i} + eps I)^-1 = L_{i,<i} T_{<i,<i};
//   * joint mean mu_<=t = L a with a = [T_ii m_i]_i, joint covariance S_<=t = (L H)(L H)^T with H = blockdiag(T_ii Lu_i);
//   * predictive moments: P = T K_uf, V2 = T^T P, W = H^T P (block-diagonal):
//         mu_b = sum_m P a,   var_b = gamma^2 - |P_b|^2 + |W_b|^2 + eps |V2_b|^2
//     (the eps term is the jitter of chol(S_<=t + eps I), gp_utils.py:182);
//   * p(u_t | u_<t) has covariance + jitter = L_tt L_tt^T (last diagonal block), so
//         KL[s,c] = sum log diag L_tt - sum log diag Lu_t + 0.5 (|H_t|_F^2 + |a_t|^2 - M)        (ep_var_mean = True).
// So a step is: two kernel-matrix GEMMs, ONE blocked factorisation of size Mt, four GEMMs, two reductions -- and the
// backward is GEMMs as well (the Cholesky adjoint needs no L at all: gL is diagonal here).
// Gradients reach theta (through both kernel matrices), the current z (last M rows of each class), u_mean, u_tril_vec.
#include "common.h"
#include "elbo_shared.h"

namespace vargp {

constexpr int kNmMaxV = 16;       // most u_<t samples (n_v) of the ep_var_mean = False KL

struct TnWs {
  float *theta, *eps_theta, *eps_f;          // first, in this order (vargp_amd/fused.py exposes them as views)
  float *g2, *kd, *w, *na, *nb, *xs, *mu, *var, *gmu, *gvar;   // gmu, gvar adjacent: one zero range (accumulated by the softmax kernel)
  int64_t Dp;
  float *Kall, *Kuf, *LL, *TT, *QPs, *P, *V2, *W;
  float *gQPs, *gP, *gT, *gKuf, *gK, *gRKt, *gkd, *gz_all;
  float *r_uf, *c_uf, *gtheta;               // adjacent: one zero range (accumulators of the kernel-matrix backward)
  float *r_uu, *Wuu, *Puu, *Puf;             // Wuu aliases gT (dead once the Cholesky backward has consumed it)
  float *nm_v, *nm_y1, *nm_y2, *nm_d, *nm_gy2, *nm_gy1;   // ep_var_mean = False (tn_nm_* kernels): [b][row][kNmMaxV]
  float* zs;                                 // aliases Puu
  void *chol, *rbf;
  size_t chol_bytes, rbf_bytes;
  int NRs, Mt;
  size_t bytes;
};

// fwd_only: the carve of a program that only ever evaluates predictive moments (VARGP.forward / predict, no backward):
// the gradient buffers are not carved at all (null), z_all o w gets a buffer of its own.
static TnWs carve_tn(void* ws, int S, int C, int M, int D, int B, int F, int nblk, bool fwd_only = false) {
  TnWs o{};
  o.Mt = M * nblk;
  o.NRs = (int)round_up(4 + M, 4);
  const int64_t SC = (int64_t)S * C, Mt = o.Mt, D1 = D + 1;
  float* p = reinterpret_cast<float*>(ws);
  auto take = [&](int64_t n) { float* q = p; p += round_up(n, 64); return q; };
  auto takeb = [&](int64_t n) { return fwd_only ? (float*)nullptr : take(n); };      // backward-only buffers
  o.theta = take(S * D1); o.eps_theta = take(S * D1); o.eps_f = takeb((int64_t)S * F * C * B);
  o.Dp = round_up(D, 4);
  o.g2 = take(S); o.kd = take(SC);
  o.w = take(S * o.Dp); o.na = take(SC * Mt); o.nb = take((int64_t)S * B);
  o.xs = take((int64_t)S * B * D);           // x o w per hyper-sample: the K_uf GEMM then needs no per-k scaling
  o.mu = take(SC * B); o.var = take(SC * B);
  o.gmu = take(SC * B); o.gvar = take(SC * B);
  o.Kall = take(SC * Mt * Mt); o.Kuf = take(SC * Mt * B);
  o.LL = take(SC * Mt * Mt); o.TT = take(SC * Mt * Mt);
  o.QPs = take(SC * Mt * o.NRs);
  o.P = take(SC * Mt * B); o.V2 = take(SC * Mt * B); o.W = take(SC * Mt * B);
  o.gQPs = takeb(SC * Mt * o.NRs); o.gP = takeb(SC * Mt * B);
  o.gT = takeb(SC * Mt * Mt); o.gKuf = takeb(SC * Mt * B); o.gK = takeb(SC * Mt * Mt);
  o.gRKt = takeb(SC * M * o.NRs); o.gkd = takeb(SC);
  o.gz_all = takeb((int64_t)C * Mt * D);
  o.r_uf = takeb(SC * Mt); o.c_uf = takeb((int64_t)S * B); o.gtheta = takeb(S * D1);
  o.r_uu = takeb(SC * Mt); o.Wuu = o.gT; o.Puu = take(SC * Mt * D); o.Puf = takeb(SC * Mt * D);
  o.zs = o.Puu;       // z_all o w per hyper-sample (forward only; P_uu is written by the backward)
  if (nblk > 1) {     // the u_<t samples of the ep_var_mean = False KL and what its backward keeps (small: rows x kNmMaxV)
    const int64_t Ml = Mt - M;
    o.nm_v = takeb(SC * Ml * kNmMaxV); o.nm_y1 = takeb(SC * Ml * kNmMaxV); o.nm_gy1 = takeb(SC * Ml * kNmMaxV);
    o.nm_y2 = takeb(SC * M * kNmMaxV); o.nm_d = takeb(SC * M * kNmMaxV); o.nm_gy2 = takeb(SC * M * kNmMaxV);
  }
  const size_t cf = vargp_chol_workspace_bytes((int)SC, o.Mt, 0), cb = fwd_only ? 0 : vargp_chol_workspace_bytes((int)SC, o.Mt, 1);
  o.chol_bytes = cf > cb ? cf : cb;
  o.chol = p;
  p += round_up((int64_t)(o.chol_bytes + 3) / 4, 64);
  o.rbf_bytes = 0;
  o.rbf = nullptr;
  o.bytes = (size_t)((char*)p - (char*)ws);
  return o;
}

// ---------------------------------------------------------------------------------------------------------------
// forward kernels
// ---------------------------------------------------------------------------------------------------------------
struct TnProArgs {
  const float *mean, *logvar, *pmean, *plogvar, *eps_theta, *vec, *u_mean, *z;
  float *theta, *g2, *kd, *scalars, *zero_begin, *bump, *rk_last, *z_all;
  int32_t* info;
  int64_t zero_count;
  int S, C, M, D, Mt, NRs, nblk, ninfo, map_est, nzero_blocks, npack_blocks;
  int native, nrng_blocks;
  uint64_t seed;
  const uint32_t* rng_counter;
  int64_t g0_theta, g0_f, n_f;
  float *eps_theta_out, *eps_f_out;
};

// Multi-role prologue, role by block index:
//   block 0            kl_hypers (kernels.py:70-77) -> scalars[0]; scalars[1..2] = 0; info = 0; *bump += 1
//   blocks 1..S        theta_s = mean + eps_s exp(logvar/2) (kernels.py:62-68), gamma_s^2 (kernels.py:58-60)
//   next nzero_blocks  zero-fill of the softmax-gradient accumulators
//   next nrng_blocks   (native noise only) the likelihood noise
//   next npack_blocks  the current task's block of the packed operand rk_all: [u_mean | 0 0 0 | Lu = vec2tril(vec) | 0..]
//                      (gp_utils.py:22-49: softplus on the diagonal)
//   rest               the current inducing points into the last M rows of every class of z_all
__global__ __launch_bounds__(256) void tn_prologue_kernel(const TnProArgs a) {
  __shared__ float red[4];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const int D1 = a.D + 1;
  if (blk == 0) {
    float acc = 0.f;
    if (!a.map_est)
      for (int d = tid; d < D1; d += 256) {
        const float dv = a.logvar[d] - a.plogvar[d], dm = a.mean[d] - a.pmean[d];
        acc += 0.5f * (expf(dv) + dm * dm * expf(-a.plogvar[d]) - 1.f - dv);
      }
    const float t = block_sum<256>(acc, red);
    if (tid == 0) {
      a.scalars[0] = t; a.scalars[1] = 0.f; a.scalars[2] = 0.f;
      if (a.bump) a.bump[0] += 1.f;
    }
    for (int i = tid; i < a.ninfo; i += 256) a.info[i] = 0;
    return;
  }
  if (blk <= a.S) {
    const int s = blk - 1;
    for (int d = tid; d < D1; d += 256) {
      float t;
      if (a.map_est) {
        t = a.mean[d];
      } else {
        float e;
        if (a.native) {
          e = normal1(a.seed, kStreamTheta, (uint64_t)(a.g0_theta + (int64_t)s * D1 + d), a.rng_counter[0]);
          a.eps_theta_out[s * D1 + d] = e;
        } else {
          e = a.eps_theta[s * D1 + d];
        }
        t = a.mean[d] + e * expf(0.5f * a.logvar[d]);
      }
      a.theta[s * D1 + d] = t;
      if (d == a.D) {
        const float g = expf(2.f * t);
        a.g2[s] = g;
        for (int c = 0; c < a.C; ++c) a.kd[s * a.C + c] = g;
      }
    }
    return;
  }
  int id = blk - 1 - a.S;
  if (id < a.nzero_blocks) {
    for (int64_t i = (int64_t)id * 256 + tid; i < a.zero_count; i += (int64_t)a.nzero_blocks * 256) a.zero_begin[i] = 0.f;
    return;
  }
  id -= a.nzero_blocks;
  if (id < a.nrng_blocks) {
    const uint32_t step = a.rng_counter[0];
    const int64_t gfirst = a.g0_f >> 2, glast = (a.g0_f + a.n_f + 3) >> 2;
    for (int64_t G = gfirst + (int64_t)id * 256 + tid; G < glast; G += (int64_t)a.nrng_blocks * 256) {
      float v[4];
      normal4(a.seed, kStreamF, (uint64_t)G, step, v);
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int64_t i = 4 * G + l - a.g0_f;
        if (i >= 0 && i < a.n_f) a.eps_f_out[i] = v[l];
      }
    }
    return;
  }
  id -= a.nrng_blocks;
  const int M = a.M;
  if (id < a.npack_blocks) {
    const int64_t e = (int64_t)id * 256 + tid;          // (c, i, col) of the last block, col < NRs
    if (e >= (int64_t)a.C * M * a.NRs) return;
    const int col = e % a.NRs, i = (e / a.NRs) % M;
    const int64_t c = e / ((int64_t)a.NRs * M);
    float v = 0.f;
    if (col == 0) v = a.u_mean[c * M + i];
    else if (col >= 4 && col < 4 + M) {
      const int j = col - 4;
      if (j <= i) {
        const float x = a.vec[c * ((int64_t)M * (M + 1) / 2) + (int64_t)i * (i + 1) / 2 + j];
        v = j == i ? softplus_t0(x) : x;
      }
    }
    a.rk_last[(c * a.nblk * M + i) * a.NRs + col] = v;      // rk_last = rk_all + (nblk - 1) * M * NRs; class stride nblk*M*NRs
    return;
  }
  id -= a.npack_blocks;
  const int64_t e = (int64_t)id * 256 + tid;              // (c, i, d) of the current inducing points
  if (e >= (int64_t)a.C * M * a.D) return;
  const int d = e % a.D, i = (e / a.D) % M;
  const int64_t c = e / ((int64_t)a.D * M);
  a.z_all[(c * a.Mt + (a.Mt - M) + i) * a.D + d] = a.z[e];
}

// Two roles.  Blocks < npd: predictive mean / variance, 64 minibatch columns x 4 row lanes over all Mt rows:
//   mu = sum_m P a,  var = kd - sum P^2 + sum W^2 + eps sum V2^2     (a = column 0 of QPs, one value per row of P).
// Blocks >= npd: the KL of q(u_t | u_<t) against p(u_t | u_<t) (vargp.py:182-190) from the last block of QPs:
//   kl[s,c] = sum log diag L_tt - sum log diag Lu_t + 0.5 (|H_t|_F^2 + |a_t|^2 - M),  kl_u = (1/S) sum kl[s,c]  (atomic)
// COLS = 64 or 32 columns per block ((256 / COLS) row lanes): 32 when 64 would give fewer than two blocks per CU -- every
// thread then walks half as many rows (Split-MNIST t = 1: 240 blocks of 50 dependent row steps took 22.9 us)

// ---------------------------------------------------------------------------------------------------------------
// ep_var_mean = False (reference var_gp/vargp.py:137-152): the KL of q(u_t) against p(u_t | u_<t) keeps the conditional
// prior's MEAN, evaluated at n_v samples u_<t ~ q(u_<t | theta) (eps_u (n_v, S, C, M<)):
//     kl[s,c] = ... + 0.5 mean_j |d_j|^2   instead of   0.5 |a_t|^2,        d_j = L_tt^-1 (u_mean - prior_mu_j).
// In the block form (M< = Mt - M rows of earlier tasks, t = the current block):
//     u_<t = L_<< v,  v = a_< + H_< eps   (L_<< H_< IS the Cholesky factor of S_<t: no second factorisation),
//     prior_mu = K_t< K'_<<^-1 u_<t = K_t< T_<<^T v,          d = a_t - T_tt (K_t< (T_<<^T v)).
// Forward: v, y1 = T_<<^T v, y2 = K_t< y1, d (+ the KL term); backward: the transposes, adding to ga / gH of the earlier blocks
// (gQPs), to the T_tt and T_<< blocks of gT, and -- symmetrically split -- to the (t, <) block of gK.  All of it is
// matrix-vector sized (n_v <= 16 columns): plain kernels, one thread or one wave per output row.
// ---------------------------------------------------------------------------------------------------------------
// v[b][k][j] = a[k] + sum_{l <= k % M} H[k][l] eps[j, s, c, (k / M) M + l]
__global__ __launch_bounds__(256) void tn_nm_v_kernel(const float* __restrict__ QPs, const float* __restrict__ eps_u,
                                                      float* __restrict__ v, int S, int C, int M, int Mt, int NRs, int NV) {
  const int Ml = Mt - M;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * C * Ml * NV) return;
  const int j = e % NV, k = (e / NV) % Ml;
  const int64_t b = e / ((int64_t)NV * Ml);
  const int kl = k % M, k0 = k - kl;
  const float* q = QPs + (b * Mt + k) * NRs;
  const float* ep = eps_u + ((int64_t)j * S * C + b) * Ml + k0;
  float acc = q[0];
  for (int l = 0; l <= kl; ++l) acc = fmaf(q[4 + l], ep[l], acc);
  v[(b * Ml + k) * kNmMaxV + j] = acc;
}
// y1[b][c][j] = sum_{k >= c} T[b][k][c] v[b][k][j]   (c < M<: the leading block of T, transposed)
__global__ __launch_bounds__(256) void tn_nm_y1_kernel(const float* __restrict__ T, const float* __restrict__ v,
                                                       float* __restrict__ y1, int64_t SC, int M, int Mt, int NV) {
  const int Ml = Mt - M;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= SC * Ml * NV) return;
  const int j = e % NV, c = (e / NV) % Ml;
  const int64_t b = e / ((int64_t)NV * Ml);
  const float* t = T + b * Mt * Mt + c;
  const float* vv = v + b * Ml * kNmMaxV + j;
  float a0 = 0.f, a1 = 0.f;
  int k = c;
  for (; k + 1 < Ml; k += 2) {
    a0 = fmaf(t[(int64_t)k * Mt], vv[(int64_t)k * kNmMaxV], a0);
    a1 = fmaf(t[(int64_t)(k + 1) * Mt], vv[(int64_t)(k + 1) * kNmMaxV], a1);
  }
  if (k < Ml) a0 = fmaf(t[(int64_t)k * Mt], vv[(int64_t)k * kNmMaxV], a0);
  y1[(b * Ml + c) * kNmMaxV + j] = a0 + a1;
}
// one wave per (b, i): out[b][i][:] = sum_c A[b][row0 + i][c] x[b][c][:]  over c < ncols (<= i + 1 if tri), A row-major (ld Mt)
__device__ __forceinline__ void nm_row_dot(const float* __restrict__ arow, const float* __restrict__ x, int ncols, int NV, int lane,
                                           float (&acc)[kNmMaxV]) {
#pragma unroll
  for (int j = 0; j < kNmMaxV; ++j) acc[j] = 0.f;
  for (int c = lane; c < ncols; c += 64) {
    const float a = arow[c];
    const float* xr = x + (int64_t)c * kNmMaxV;
#pragma unroll
    for (int j = 0; j < kNmMaxV; ++j) if (j < NV) acc[j] = fmaf(a, xr[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < kNmMaxV; ++j) if (j < NV) acc[j] = wave_sum(acc[j]);
}
// y2[b][i][:] = sum_{c < M<} K[b][r0 + i][c] y1[b][c][:]     (one wave per (b, i))
__global__ __launch_bounds__(256) void tn_nm_y2_kernel(const float* __restrict__ K, const float* __restrict__ y1,
                                                       float* __restrict__ y2, int64_t SC, int M, int Mt, int NV) {
  const int Ml = Mt - M, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= SC * M) return;
  const int i = w % M;
  const int64_t b = w / M;
  float acc[kNmMaxV];
  nm_row_dot(K + (b * Mt + Ml + i) * Mt, y1 + b * Ml * kNmMaxV, Ml, NV, lane, acc);
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < kNmMaxV; ++j) if (j < NV) y2[(b * M + i) * kNmMaxV + j] = acc[j];
}
// d[b][i][j] = a_t[i] - sum_{l <= i} T_tt[i][l] y2[b][l][j];  kl_u += 0.5 / (S NV) sum d^2     (one block per b)
__global__ __launch_bounds__(256) void tn_nm_d_kernel(const float* __restrict__ T, const float* __restrict__ QPs,
                                                      const float* __restrict__ y2, float* __restrict__ dd,
                                                      float* __restrict__ kl_u, int S, int M, int Mt, int NRs, int NV) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const int r0 = Mt - M;
  float acc2 = 0.f;
  for (int e = threadIdx.x; e < M * NV; e += 256) {
    const int j = e % NV, i = e / NV;
    const float* t = T + (b * Mt + r0 + i) * Mt + r0;
    const float* yy = y2 + b * M * kNmMaxV + j;
    float a0 = QPs[(b * Mt + r0 + i) * NRs], a1 = 0.f;
    int l = 0;
    for (; l + 1 <= i; l += 2) { a0 = fmaf(-t[l], yy[(int64_t)l * kNmMaxV], a0); a1 = fmaf(-t[l + 1], yy[(int64_t)(l + 1) * kNmMaxV], a1); }
    if (l <= i) a0 = fmaf(-t[l], yy[(int64_t)l * kNmMaxV], a0);
    const float dv = a0 + a1;
    dd[(b * M + i) * kNmMaxV + j] = dv;
    acc2 = fmaf(dv, dv, acc2);
  }
  const float tot = block_sum<256>(acc2, red);
  if (threadIdx.x == 0) atomicAdd(kl_u, 0.5f * tot / ((float)S * (float)NV));
}
// backward, step 1 (one block per b):  gd = g d / NV (g = seed_kl / S);  gy2[l][j] = - sum_{i >= l} T_tt[i][l] gd[i][j];
//     ga_t[i] += sum_j gd[i][j]  (column 0 of gQPs: the head kernel left the moments' share there)
__global__ __launch_bounds__(256) void tn_nm_bwd1_kernel(const float* __restrict__ T, const float* __restrict__ dd,
                                                         const float* __restrict__ seeds, float* __restrict__ gy2,
                                                         float* __restrict__ gQPs, int S, int M, int Mt, int NRs, int NV) {
  const int64_t b = blockIdx.x;
  const int r0 = Mt - M;
  const float gf = seeds[1] / ((float)S * (float)NV);
  for (int e = threadIdx.x; e < M * NV; e += 256) {
    const int j = e % NV, l = e / NV;
    const float* t = T + (b * Mt + r0) * Mt + r0 + l;
    const float* dv = dd + b * M * kNmMaxV + j;
    float a0 = 0.f;
    for (int i = l; i < M; ++i) a0 = fmaf(t[(int64_t)i * Mt], dv[(int64_t)i * kNmMaxV], a0);
    gy2[(b * M + l) * kNmMaxV + j] = -gf * a0;
  }
  for (int i = threadIdx.x; i < M; i += 256) {
    float a0 = 0.f;
    for (int j = 0; j < NV; ++j) a0 += dd[(b * M + i) * kNmMaxV + j];
    gQPs[(b * Mt + r0 + i) * NRs] += gf * a0;
  }
}
// backward, step 2:  gy1[b][c][j] = sum_i K[b][r0 + i][c] gy2[b][i][j]
__global__ __launch_bounds__(256) void tn_nm_bwd2_kernel(const float* __restrict__ K, const float* __restrict__ gy2,
                                                         float* __restrict__ gy1, int64_t SC, int M, int Mt, int NV) {
  const int Ml = Mt - M;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= SC * Ml * NV) return;
  const int j = e % NV, c = (e / NV) % Ml;
  const int64_t b = e / ((int64_t)NV * Ml);
  const float* kk = K + (b * Mt + Ml) * Mt + c;
  const float* g = gy2 + b * M * kNmMaxV + j;
  float a0 = 0.f, a1 = 0.f;
  int i = 0;
  for (; i + 1 < M; i += 2) { a0 = fmaf(kk[(int64_t)i * Mt], g[(int64_t)i * kNmMaxV], a0); a1 = fmaf(kk[(int64_t)(i + 1) * Mt], g[(int64_t)(i + 1) * kNmMaxV], a1); }
  if (i < M) a0 = fmaf(kk[(int64_t)i * Mt], g[(int64_t)i * kNmMaxV], a0);
  gy1[(b * Ml + c) * kNmMaxV + j] = a0 + a1;
}
// backward, step 3 (one wave per (b, k), k < M<):  gv[k][:] = sum_{c <= k} T[k][c] gy1[c][:];
//     ga_<[k] += sum_j gv[k][j];   gH[k][l] += sum_j gv[k][j] eps[j, s, c, (k / M) M + l]   (l <= k % M)
__global__ __launch_bounds__(256) void tn_nm_bwd3_kernel(const float* __restrict__ T, const float* __restrict__ gy1,
                                                         const float* __restrict__ eps_u, float* __restrict__ gQPs, int S, int C,
                                                         int M, int Mt, int NRs, int NV) {
  const int Ml = Mt - M, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)S * C * Ml) return;
  const int k = w % Ml;
  const int64_t b = w / Ml;
  float gv[kNmMaxV];
  nm_row_dot(T + (b * Mt + k) * Mt, gy1 + b * Ml * kNmMaxV, k + 1, NV, lane, gv);
  const int kl = k % M, k0 = k - kl;
  float* q = gQPs + (b * Mt + k) * NRs;
  if (lane == 0) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < kNmMaxV; ++j) if (j < NV) t += gv[j];
    q[0] += t;
  }
  for (int l = lane; l <= kl; l += 64) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < kNmMaxV; ++j) if (j < NV) t = fmaf(gv[j], eps_u[((int64_t)j * S * C + b) * Ml + k0 + l], t);
    q[4 + l] += t;
  }
}
// backward, step 4 (after gT is complete, before the Cholesky adjoint):
//     gT_tt[i][l] -= sum_j gd[i][j] y2[l][j]  (l <= i);      gT_<<[k][c] += sum_j v[k][j] gy1[c][j]  (c <= k)
__global__ __launch_bounds__(256) void tn_nm_bwd4_kernel(const float* __restrict__ dd, const float* __restrict__ y2,
                                                         const float* __restrict__ v, const float* __restrict__ gy1,
                                                         const float* __restrict__ seeds, float* __restrict__ gT, int64_t SC, int S,
                                                         int M, int Mt, int NV, int64_t n_tt) {
  const int Ml = Mt - M;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_tt) {                                   // (b, i, l) of the T_tt block
    const int l = e % M, i = (e / M) % M;
    const int64_t b = e / ((int64_t)M * M);
    if (l > i) return;
    const float gf = seeds[1] / ((float)S * (float)NV);
    const float* dv = dd + (b * M + i) * kNmMaxV;
    const float* yy = y2 + (b * M + l) * kNmMaxV;
    float t = 0.f;
    for (int j = 0; j < NV; ++j) t = fmaf(dv[j], yy[j], t);
    gT[(b * Mt + Ml + i) * Mt + Ml + l] -= gf * t;
    return;
  }
  const int64_t f = e - n_tt;                        // (b, k, c) of the leading M< x M< block
  if (f >= SC * Ml * Ml) return;
  const int c = f % Ml, k = (f / Ml) % Ml;
  const int64_t b = f / ((int64_t)Ml * Ml);
  if (c > k) return;
  const float* vv = v + (b * Ml + k) * kNmMaxV;
  const float* g = gy1 + (b * Ml + c) * kNmMaxV;
  float t = 0.f;
  for (int j = 0; j < NV; ++j) t = fmaf(vv[j], g[j], t);
  gT[(b * Mt + k) * Mt + c] += t;
}
// backward, step 5 (after the Cholesky adjoint has written the symmetric gK): the direct dependence on K_t<, split over the
// two mirrored blocks:  gK[r0 + i][c] += h,  gK[c][r0 + i] += h,   h = 0.5 sum_j gy2[i][j] y1[c][j]
__global__ __launch_bounds__(256) void tn_nm_bwd5_kernel(const float* __restrict__ gy2, const float* __restrict__ y1,
                                                         float* __restrict__ gK, int64_t SC, int M, int Mt, int NV) {
  const int Ml = Mt - M;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= SC * M * Ml) return;
  const int c = e % Ml, i = (e / Ml) % M;
  const int64_t b = e / ((int64_t)Ml * M);
  const float* g = gy2 + (b * M + i) * kNmMaxV;
  const float* yy = y1 + (b * Ml + c) * kNmMaxV;
  float t = 0.f;
  for (int j = 0; j < NV; ++j) t = fmaf(g[j], yy[j], t);
  t *= 0.5f;
  gK[(b * Mt + Ml + i) * Mt + c] += t;
  gK[(b * Mt + c) * Mt + Ml + i] += t;
}

constexpr int kTnKlRows = 8;
template <int COLS>
__global__ __launch_bounds__(256) void tn_pdiag_kl_kernel(const float* __restrict__ P, const float* __restrict__ W,
                                                          const float* __restrict__ V2, const float* __restrict__ QPs,
                                                          const float* __restrict__ kd, const float* __restrict__ L,
                                                          const float* __restrict__ rk_all, float* __restrict__ mu,
                                                          float* __restrict__ var, float* __restrict__ kl_u, float eps,
                                                          int S, int C, int M, int Mt, int nblk, int B, int NRs, int nbx,
                                                          int npd, int nkx, uint32_t* rng_counter, int a_term = 1) {
  constexpr int RL = 256 / COLS;       // row lanes
  __shared__ float red[4][RL][COLS];
  if (rng_counter && blockIdx.x == 0 && threadIdx.x == 0) rng_counter[0] += 1u;   // this step's noise has been drawn
  if ((int)blockIdx.x < npd) {
    const int cx = threadIdx.x % COLS, ry = threadIdx.x / COLS;
    const int col = ((int)blockIdx.x % nbx) * COLS + cx;
    const int64_t b = blockIdx.x / nbx;
    float m0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (col < B) {
      const float* p = P + b * Mt * B + col;
      const float* w = W + b * Mt * B + col;
      const float* v = V2 + b * Mt * B + col;
      const float* q = QPs + b * Mt * NRs;
      // eight rows per round, ALL their loads first (clamped rows, masked sums): as `#pragma unroll 8` over load-use iterations the
      // compiler kept every iteration's four loads next to their uses -- Mt / RL memory round trips in a row (Permuted-MNIST t = 1:
      // a hundred of them per thread, 85 us for a kernel that reads 120 MB)
      for (int m0r = ry; m0r < Mt; m0r += 8 * RL) {
        float pv[8], wv[8], vv[8], qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t m = min(m0r + u * RL, Mt - 1);
          pv[u] = p[m * B]; wv[u] = w[m * B]; vv[u] = v[m * B]; qv[u] = q[m * NRs];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = m0r + u * RL < Mt;
          const float pp = ok ? pv[u] : 0.f, ww = ok ? wv[u] : 0.f, v2 = ok ? vv[u] : 0.f;
          m0 = fmaf(pp, qv[u], m0);
          d1 = fmaf(pp, pp, d1);
          d2 = fmaf(ww, ww, d2);
          d3 = fmaf(v2, v2, d3);
        }
      }
    }
    red[0][ry][cx] = m0; red[1][ry][cx] = d1; red[2][ry][cx] = d2; red[3][ry][cx] = d3;
    __syncthreads();
    if (ry == 0 && col < B) {
      m0 = d1 = d2 = d3 = 0.f;
#pragma unroll
      for (int r = 0; r < RL; ++r) { m0 += red[0][r][cx]; d1 += red[1][r][cx]; d2 += red[2][r][cx]; d3 += red[3][r][cx]; }
      mu[b * B + col] = m0;
      var[b * B + col] = kd[b] - d1 + d2 + eps * d3;
    }
    return;
  }
  if (!kl_u) return;
  const int id = (int)blockIdx.x - npd;
  const int64_t b = id / nkx;          // s * C + c
  const int c = b % C;
  const int i0 = (id % nkx) * kTnKlRows, i1 = min(M, i0 + kTnKlRows);
  const int r0 = Mt - M;               // first row of the current task's block
  const float* q = QPs + (b * Mt + r0) * NRs;
  const float* rk = rk_all + ((int64_t)c * nblk + (nblk - 1)) * M * NRs;
  float acc = 0.f;
  for (int e = threadIdx.x; e < (i1 - i0) * M; e += 256) {
    const int i = i0 + e / M, j = e % M;
    if (j <= i) { const float v = q[(int64_t)i * NRs + 4 + j]; acc = fmaf(v, v, acc); }
  }
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    const float a = a_term ? q[(int64_t)i * NRs] : 0.f;      // (ep_var_mean = False: tn_nm_d_kernel adds mean_j |d_j|^2 instead)
    acc = fmaf(a, a, acc);
    acc += 2.f * (logf(L[(b * Mt + r0 + i) * Mt + r0 + i]) - logf(rk[(int64_t)i * NRs + 4 + i])) - 1.f;
  }
  const float t = block_sum<256>(acc, &red[0][0][0]);
  if (threadIdx.x == 0) atomicAdd(kl_u, 0.5f * t / (float)S);
}

// ---------------------------------------------------------------------------------------------------------------
// backward kernels
// ---------------------------------------------------------------------------------------------------------------
// First backward launch, two roles by block index.
//   blocks < npd (one per row (b, m), m < Mt): gP = a gmu - 2 P gvar; W <- gW = 2 W gvar and V2 <- gV2 = 2 eps V2 gvar
//       IN PLACE; ga = sum_col P gmu (+ g a on the current task's block: KL) -> column 0 of gQPs; row 0 of each b also
//       reduces gkd = sum_col gvar.  gscale (nullable) = seed multiplying the stored unscaled softmax gradients.
//   rest: the other columns of gQPs: 1..3 and the padding = 0; the H columns = g tril(H_t) on the current task's block
//       (KL), 0 elsewhere (the product P gW^T is accumulated on top by a GEMM).                       (g = seed_kl / S)
//   last: zero-fill of the accumulators of the kernel-matrix backward.
__global__ __launch_bounds__(256) void tn_bwd_head_kernel(const float* __restrict__ P, float* __restrict__ W,
                                                          float* __restrict__ V2, const float* __restrict__ QPs,
                                                          const float* __restrict__ gmu, const float* __restrict__ gvar,
                                                          const float* __restrict__ gscale, const float* __restrict__ seeds,
                                                          float* __restrict__ gP, float* __restrict__ gQPs,
                                                          float* __restrict__ gkd, float eps, int S, int M, int Mt, int B,
                                                          int NRs, int npd, int nrest, float* __restrict__ zero_begin,
                                                          int64_t zero_count, int accumulate, int a_term = 1) {
  __shared__ float red[4];
  const float g = seeds[1] / (float)S;
  if ((int)blockIdx.x >= npd + nrest) {       // zero-fill of the r / c / gtheta accumulators of the kernel-matrix backward
    const int nz = (int)gridDim.x - npd - nrest;
    for (int64_t i = (int64_t)((int)blockIdx.x - npd - nrest) * 256 + threadIdx.x; i < zero_count; i += (int64_t)nz * 256)
      zero_begin[i] = 0.f;
    return;
  }
  if ((int)blockIdx.x < npd) {
    const int m = (int)blockIdx.x % Mt;
    const int64_t b = blockIdx.x / Mt;
    const int64_t off = (b * Mt + m) * B;
    const float am = QPs[(b * Mt + m) * NRs];
    const float gs = gscale ? gscale[0] : 1.f;
    float acc = 0.f, accv = 0.f;
    for (int col = threadIdx.x; col < B; col += 256) {
      const float gm = gs * gmu[b * B + col], gv = gs * gvar[b * B + col];
      const float pv = P[off + col];
      gP[off + col] = am * gm - 2.f * pv * gv;
      W[off + col] = 2.f * W[off + col] * gv;
      V2[off + col] = 2.f * eps * V2[off + col] * gv;
      acc = fmaf(pv, gm, acc);
      accv += gv;
    }
    const float t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
      // accumulate (N-tiled ELBO): this minibatch tile's share on top of the earlier tiles'; the KL term comes at the end
      if (accumulate) gQPs[(b * Mt + m) * NRs] += t;
      else gQPs[(b * Mt + m) * NRs] = t + ((m >= Mt - M && a_term) ? g * am : 0.f);
    }
    if (m == 0) {
      const float tv = block_sum<256>(accv, red);
      if (threadIdx.x == 0) { if (accumulate) gkd[b] += tv; else gkd[b] = tv; }
    }
    return;
  }
  const int64_t e = (int64_t)((int)blockIdx.x - npd) * 256 + threadIdx.x;       // (row, col >= 1) of gQPs
  const int cols = NRs - 1;
  const int64_t row = e / cols;
  const int col = 1 + (int)(e % cols);
  if (row >= (int64_t)npd) return;             // npd = number of rows of gQPs
  const int m = row % Mt;
  float v = 0.f;
  if (m >= Mt - M && col >= 4 && col < 4 + M) {
    const int i = m - (Mt - M), j = col - 4;
    if (j <= i) v = g * QPs[row * NRs + col];
  }
  gQPs[row * NRs + col] = v;
}

// Parameter gradients of the current task from gRKt[s,c] = T_tt^T [ga_t | . | gH_t]:
//   g_u_mean[c,i] = sum_s gRKt[s,c,i,0]
//   gLu[c,i,k]    = sum_s gRKt[s,c,i,4+k]  (k <= i)  - seed_kl / Lu_ii on the diagonal, through vec2tril (softplus')
// and, in the remaining blocks, the gradient of the current inducing points = the last M rows of every class of gz_all.
__global__ __launch_bounds__(256) void tn_unpack_kernel(const float* __restrict__ gRKt, const float* __restrict__ vec,
                                                        const float* __restrict__ rk_last, const float* __restrict__ seeds,
                                                        const float* __restrict__ gz_all, float* __restrict__ g_u_mean,
                                                        float* __restrict__ gvec, float* __restrict__ g_z, int S, int C, int M,
                                                        int Mt, int D, int NRs, int nblk, int nun) {
  if ((int)blockIdx.x < nun) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (c, i, jj), jj = 0: mean, jj = 1 + k: Lu_ik
    if (e >= (int64_t)C * M * (M + 1)) return;
    const int jj = e % (M + 1), i = (e / (M + 1)) % M;
    const int64_t c = e / ((int64_t)(M + 1) * M);
    const int k = jj - 1;
    if (k > i) return;
    const int col = jj == 0 ? 0 : 3 + jj;
    float acc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {          // eight samples' loads in flight together
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = gRKt[(((int64_t)min(s0 + u, S - 1) * C + c) * M + i) * NRs + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (s0 + u < S) ? t[u] : 0.f;
    }
    if (jj == 0) { g_u_mean[c * M + i] = acc; return; }
    const int64_t idx = c * ((int64_t)M * (M + 1) / 2) + (int64_t)i * (i + 1) / 2 + k;
    if (i == k) {
      acc -= seeds[1] / rk_last[(c * nblk * M + i) * NRs + 4 + i];
      const float x = vec[idx];
      acc *= (x > 20.f) ? 1.f : sigmoid_t0(x);
    }
    gvec[idx] = acc;
    return;
  }
  const int64_t e = (int64_t)((int)blockIdx.x - nun) * 256 + threadIdx.x;
  if (e >= (int64_t)C * M * D) return;
  const int d = e % D, i = (e / D) % M;
  const int64_t c = e / ((int64_t)D * M);
  g_z[e] = gz_all[(c * Mt + (Mt - M) + i) * D + d];
}

// N-tiled ELBO, start of a minibatch tile: zero the per-tile accumulators (softmax gradients, column sums of W_uf) and,
// with native noise, draw this tile's likelihood noise (one generator step per tile).
__global__ __launch_bounds__(256) void tn_tile_prep_kernel(float* __restrict__ z0, int64_t n0, float* __restrict__ z1, int64_t n1,
                                                           int nzero, int native, uint64_t seed, const uint32_t* rng_counter,
                                                           int64_t g0_f, int64_t n_f, float* __restrict__ eps_f_out) {
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < nzero) {
    for (int64_t i = (int64_t)blk * 256 + tid; i < n0 + n1; i += (int64_t)nzero * 256) {
      if (i < n0) z0[i] = 0.f; else z1[i - n0] = 0.f;
    }
    return;
  }
  if (!native) return;
  const uint32_t step = rng_counter[0];
  const int nr = (int)gridDim.x - nzero;
  const int64_t gfirst = g0_f >> 2, glast = (g0_f + n_f + 3) >> 2;
  for (int64_t G = gfirst + (int64_t)(blk - nzero) * 256 + tid; G < glast; G += (int64_t)nr * 256) {
    float v[4];
    normal4(seed, kStreamF, (uint64_t)G, step, v);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int64_t i = 4 * G + l - g0_f;
      if (i >= 0 && i < n_f) eps_f_out[i] = v[l];
    }
  }
}

// N-tiled ELBO, end: the KL's share of the small-product gradients on the current task's block (what the head kernel adds
// in the one-minibatch program):  gQPs[.., 0] += g a_t,  gQPs[.., H columns] += g tril(H_t).     (g = seed_kl / S)
__global__ void tn_kl_bwd_kernel(const float* __restrict__ QPs, float* __restrict__ gQPs, const float* __restrict__ seeds,
                                 int S, int M, int Mt, int NRs, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (b, i, col) over the last block's rows
  if (e >= total) return;
  const int col = e % NRs, i = (e / NRs) % M;
  const int64_t b = e / ((int64_t)NRs * M);
  const float g = seeds[1] / (float)S;
  const int64_t off = (b * Mt + (Mt - M) + i) * NRs + col;
  if (col == 0 || (col >= 4 && col - 4 <= i && col < 4 + M)) gQPs[off] += g * QPs[off];
}

static int check_tn(const vargp_elbo_tn_desc* d, const char* who, bool tiled = false) {
  VARGP_REQUIRE(d, "%s: null descriptor", who);
  VARGP_REQUIRE(d->S > 0 && d->C > 0 && d->M > 0 && d->D > 0 && d->B > 0 && d->F > 0 && d->nblk > 0, "%s: bad dims", who);
  VARGP_REQUIRE(d->log_mean && d->z && d->u_mean && d->u_tril_vec && d->z_all && d->rk_all && d->x && d->scalars && d->info &&
                    d->ws, "%s: null pointer", who);
  if (tiled) {   // the tile calls bring x, y and the likelihood noise; begin only needs theta's noise (given or native)
    VARGP_REQUIRE(d->map_est ? d->S == 1 : (d->log_logvar && d->prior_log_mean && d->prior_log_logvar &&
                                            (d->eps_theta || (d->rng_counter && d->rng_sample_offset >= 0))),
                  "%s: hyper-parameter arguments inconsistent", who);
    VARGP_REQUIRE(d->ws_bytes >= carve_tn(nullptr, d->S, d->C, d->M, d->D, d->B, d->F, d->nblk, d->forward_only != 0).bytes,
                  "%s: workspace too small", who);
    return VARGP_OK;
  }
  const bool native = d->eps_f == nullptr && d->y != nullptr && !d->ext_lik;
  VARGP_REQUIRE(!native || (d->rng_counter && d->eps_theta == nullptr && d->rng_sample_offset >= 0),
                "%s: native noise needs rng_counter, eps_theta == eps_f == NULL and a sample offset >= 0", who);
  VARGP_REQUIRE(d->map_est ? d->S == 1
                           : (d->log_logvar && d->prior_log_mean && d->prior_log_logvar && (native || d->eps_theta)),
                "%s: hyper-parameter arguments inconsistent with map_est", who);
  VARGP_REQUIRE(!d->forward_only || d->y == nullptr, "%s: a forward_only program evaluates moments only (y must be NULL)", who);
  VARGP_REQUIRE(!d->no_var_mean || (d->nblk > 1 && d->eps_u && d->n_v >= 1 && d->n_v <= kNmMaxV),
                "%s: no_var_mean needs earlier tasks (nblk > 1), eps_u and 1 <= n_v <= %d", who, kNmMaxV);
  VARGP_REQUIRE(d->ws_bytes >= carve_tn(nullptr, d->S, d->C, d->M, d->D, d->B, d->F, d->nblk, d->forward_only != 0).bytes,
                "%s: workspace too small", who);
  return VARGP_OK;
}

// batched product over (s, c, block): strides of the three batch dims given per operand
static GemmParams blk_gemm(const float* A, int lda, const int64_t (&sA)[3], const float* B, int ldb, const int64_t (&sB)[3],
                           float* C, int ldc, const int64_t (&sC)[3], int M, int N, int K, int nC, int nblk) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.D = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldd = ldc;
  p.nb1 = nC; p.nb2 = nblk;
  for (int i = 0; i < 3; ++i) { p.sA[i] = sA[i]; p.sB[i] = sB[i]; p.sC[i] = sC[i]; p.sD[i] = sC[i]; }
  p.alpha = 1.f; p.beta = 0.f;
  return p;
}

}  // namespace vargp

using namespace vargp;

extern "C" size_t vargp_elbo_tn_workspace_bytes(int S, int C, int M, int D, int B, int F, int nblk) {
  return carve_tn(nullptr, S, C, M, D, B, F, nblk).bytes + 256;
}

extern "C" size_t vargp_elbo_tn_workspace_bytes_fwd(int S, int C, int M, int D, int B, int F, int nblk) {
  return carve_tn(nullptr, S, C, M, D, B, F, nblk, true).bytes + 256;
}


extern "C" int vargp_elbo_tn_moments(const vargp_elbo_tn_desc* d, float** mu, float** var) {
  VARGP_REQUIRE(d && d->ws && mu && var, "elbo_tn_moments: null pointer");
  const TnWs o = carve_tn(d->ws, d->S, d->C, d->M, d->D, d->B, d->F, d->nblk, d->forward_only != 0);
  *mu = o.mu; *var = o.var;
  return VARGP_OK;
}

extern "C" int vargp_elbo_tn_lik_buffers(const vargp_elbo_tn_desc* d, float** mu, float** var, float** gmu, float** gvar) {
  VARGP_REQUIRE(d && d->ws && !d->forward_only, "elbo_tn_lik_buffers: null pointer / forward_only program");
  const TnWs o = carve_tn(d->ws, d->S, d->C, d->M, d->D, d->B, d->F, d->nblk, false);
  if (mu) *mu = o.mu;
  if (var) *var = o.var;
  if (gmu) *gmu = o.gmu;
  if (gvar) *gvar = o.gvar;
  return VARGP_OK;
}

extern "C" int vargp_elbo_tn_fwd(const vargp_elbo_tn_desc* d, vargp_stream_t stream) {
  int rc = check_tn(d, "elbo_tn_fwd");
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, F = d->F, nblk = d->nblk, SC = S * C;
  const TnWs o = carve_tn(d->ws, S, C, M, D, B, F, nblk, d->forward_only != 0);
  const int Mt = o.Mt, NRs = o.NRs;
  const int64_t MtMt = (int64_t)Mt * Mt, MtB = (int64_t)Mt * B, MtN = (int64_t)Mt * NRs;
  const bool lik = d->y != nullptr;                 // y == NULL: predictive moments only (no likelihood, no KL)
  const bool native = lik && d->eps_f == nullptr && !d->ext_lik;
  const bool fused_softmax = C <= 16;
  const float* eps_f = native ? o.eps_f : d->eps_f;
  {
    ProfScope prof("tn_prologue", st);
    TnProArgs a{};
    a.mean = d->log_mean; a.logvar = d->log_logvar; a.pmean = d->prior_log_mean; a.plogvar = d->prior_log_logvar;
    a.eps_theta = d->eps_theta; a.vec = d->u_tril_vec; a.u_mean = d->u_mean; a.z = d->z;
    a.theta = o.theta; a.g2 = o.g2; a.kd = o.kd; a.scalars = d->scalars; a.bump = d->bump;
    a.rk_last = d->rk_all + (int64_t)(nblk - 1) * M * NRs; a.z_all = d->z_all;
    a.info = d->info; a.ninfo = SC;
    a.zero_begin = o.gmu; a.zero_count = o.Kall - o.gmu;
    a.S = S; a.C = C; a.M = M; a.D = D; a.Mt = Mt; a.NRs = NRs; a.nblk = nblk; a.map_est = d->map_est;
    a.nzero_blocks = (int)std::min<int64_t>(64, cdiv(a.zero_count, 1024));
    if (native) {
      const int64_t per_sample_f = (int64_t)F * C * B;
      a.native = 1; a.seed = d->rng_seed; a.rng_counter = d->rng_counter;
      a.g0_theta = (int64_t)d->rng_sample_offset * (D + 1); a.g0_f = (int64_t)d->rng_sample_offset * per_sample_f;
      a.n_f = S * per_sample_f;
      a.eps_theta_out = o.eps_theta; a.eps_f_out = o.eps_f;
      a.nrng_blocks = (int)std::min<int64_t>(512, cdiv(a.n_f + 7, 1024));
    }
    a.npack_blocks = cdiv((int64_t)C * M * NRs, 256);
    const int grid = 1 + S + a.nzero_blocks + a.nrng_blocks + a.npack_blocks + cdiv((int64_t)C * M * D, 256);
    hipLaunchKernelGGL(tn_prologue_kernel, dim3(grid), dim3(256), 0, st, a);
  }
  // kernel matrices over ALL inducing points (earlier tasks + current): K_all (S,C,Mt,Mt), K_uf (S,C,Mt,B)
  const int64_t zrows = (int64_t)C * Mt;
  const bool mfma = D > kRbfDirectD;
  bool kuf_done = false;
  GemmParams pf{};       // K_uf = rbf(z_all, x): the classes' inducing points are just more rows of one [C*Mt, D] x [D, B] product
  rc = rbf_prep_norm_launch(o.theta, d->z_all, zrows, d->x, B, o.w, o.g2, o.na, o.nb, S, D, o.Dp, st, o.xs, o.zs);
  if (rc) return rc;
  if (!mfma) {           // small input dimension: direct (cancellation-free) distances
    rc = rbf_direct_launch(d->z_all, nullptr, o.w, o.g2, o.Kall, Mt, S, C, Mt, Mt, D, o.Dp, 0, st);
    if (rc) return rc;
    rc = rbf_direct_launch(d->z_all, d->x, o.w, o.g2, o.Kuf, B, S, C, Mt, B, D, o.Dp, 1, st);
    if (rc) return rc;
    kuf_done = true;
  } else {
    GemmParams p0{};     // K_all: tiles touching the lower triangle, mirrored (the matrix is symmetric)
    p0.A = o.zs; p0.B = d->z_all; p0.C = o.Kall;        // A pre-scaled by the norm pass: no per-k scaling in the main loop
    p0.M = Mt; p0.N = Mt; p0.K = D; p0.lda = D; p0.ldb = D; p0.ldc = Mt;
    p0.nb1 = C; p0.nb2 = 1;
    p0.sA[0] = zrows * D; p0.sA[1] = (int64_t)Mt * D; p0.sB[1] = (int64_t)Mt * D;
    p0.sC[0] = C * MtMt; p0.sC[1] = MtMt;
    p0.alpha = 1.f;
    p0.kscale = nullptr; p0.ks_ld = o.Dp; p0.g2 = o.g2;
    p0.na = o.na; p0.sNa[0] = zrows; p0.sNa[1] = Mt;
    p0.nbv = o.na; p0.sNb[0] = zrows; p0.sNb[1] = Mt;
    static const int ksym = [] { const char* e = getenv("VARGP_TN_KSYM"); return e ? atoi(e) : 1; }();   // tuning aid
    p0.same_xy = 1;
    if (ksym) { p0.triC = 2; p0.symout = 1; }
    rc = launch_gemm(p0, 0, 1, SC, true, st, "rbf_kuu_gemm");
    if (rc) return rc;
    pf.A = d->z_all; pf.B = o.xs; pf.C = o.Kuf;
    pf.M = C * Mt; pf.N = B; pf.K = D; pf.lda = D; pf.ldb = D; pf.ldc = B;
    pf.nb1 = 1; pf.nb2 = 1;
    pf.sB[0] = (int64_t)B * D;
    pf.sC[0] = (int64_t)C * MtB;
    pf.alpha = 1.f;
    pf.kscale = nullptr; pf.ks_ld = o.Dp; pf.g2 = o.g2;      // pre-scaled B operand
    pf.na = o.na; pf.sNa[0] = zrows;
    pf.nbv = o.nb; pf.sNb[0] = B;
  }
  // L = chol(K_all + eps I), T = L^-1: every factor of the reference's chain is a leading block of these.  The K_uf
  // GEMM, which nothing needs before T exists, shares the launch of the first diagonal block's pivot chain.
  // With a blocked factorisation (Mt > 100) every diagonal block has a pivot chain of its own: the K_uf GEMM is cut into
  // row slices (whole 64-row tiles of the [C*Mt x B] product), one per chain.
  constexpr int kMaxSlices = 24;   // one slice per pivot chain up to Mt = 2400
  GemmParams slices[kMaxSlices];
  int nsl = 0, consumed = 0;
  if (!kuf_done) {
    const int npanel = Mt <= 100 ? 1 : cdiv(Mt, 100);
    nsl = std::min(std::min(npanel, kMaxSlices), std::max(1, (int)(zrows / 128)));
    const int64_t per = round_up(cdiv(zrows, nsl), 64);
    nsl = cdiv(zrows, per);
    for (int i = 0; i < nsl; ++i) {
      const int64_t r0 = i * per, rows = std::min<int64_t>(per, zrows - r0);
      slices[i] = pf;
      slices[i].A = pf.A + r0 * D; slices[i].C = pf.C + r0 * B; slices[i].M = (int)rows;
      slices[i].na = pf.na + r0;
    }
  }
  rc = chol_inv_fwd_impl(o.Kall, d->jitter, o.LL, o.TT, nullptr, d->info, SC, Mt, o.chol, o.chol_bytes, false, st,
                         nsl ? slices : nullptr, S, &consumed, nsl, /*chain_f32=*/!d->forward_only && lik);
  if (rc) return rc;
  // early hand-over of the Cholesky status (include/vargp_hip.h: info_host / info_event)
  if (d->info_host && d->info_event) {
    VARGP_REQUIRE(hipMemcpyAsync(d->info_host, d->info, sizeof(int32_t) * (size_t)SC, hipMemcpyDeviceToHost, st) == hipSuccess &&
                      hipEventRecord(reinterpret_cast<hipEvent_t>(d->info_event), st) == hipSuccess,
                  "elbo_tn_fwd: copy / event record of the early Cholesky status failed");
  }
  if (!kuf_done) {
    if (consumed == 0) {
      rc = launch_gemm(pf, 0, 1, S, true, st, "rbf_kuf_gemm");
      if (rc) return rc;
    } else {
      for (int i = consumed; i < nsl; ++i) {
        rc = launch_gemm(slices[i], 0, 1, S, true, st, "rbf_kuf_gemm");
        if (rc) return rc;
      }
    }
  }
  {  // [a_i | . | H_i] = T_ii [m_i | 0 | Lu_i] for every (s, c, block i)
    const int64_t sA[3] = {C * MtMt, MtMt, (int64_t)M * Mt + M}, sB[3] = {0, (int64_t)nblk * M * NRs, (int64_t)M * NRs},
                  sC[3] = {C * MtN, MtN, (int64_t)M * NRs};
    GemmParams p = blk_gemm(o.TT, Mt, sA, d->rk_all, NRs, sB, o.QPs, NRs, sC, M, NRs, M, C, nblk);
    p.triA = 1;
    // P = T K_uf: independent of the small products -- mid-size shapes share one launch
    GemmParams q = flat_gemm(o.TT, Mt, MtMt, o.Kuf, B, MtB, o.P, B, MtB, Mt, B, Mt);
    q.triA = 1;
    const int64_t wgs = (int64_t)SC * (nblk * cdiv(M, 64) * cdiv(NRs, 64) + cdiv(Mt, 64) * cdiv(B, 64));
    static const int pair_fwd = [] { const char* e = getenv("VARGP_TN_PAIRFWD"); return e ? atoi(e) : 1; }();   // tuning aid
    if (pair_fwd && wgs <= 4096) {
      rc = launch_gemm_pair2(p, 0, 0, SC * nblk, q, 0, 0, SC, st, "tn_p_gemm");
      if (rc) return rc;
    } else {
      rc = launch_gemm(p, 0, 0, SC * nblk, false, st, "tn_small_gemm");
      if (rc) return rc;
      rc = launch_gemm(q, 0, 0, SC, false, st, "tn_p_gemm");
      if (rc) return rc;
    }
  }
  {  // V2 = T^T P  (= K'^-1 K_uf: the eps term of the variance)  and  W_i = H_i^T P_i: both only need P
    GemmParams p = flat_gemm(o.TT, Mt, MtMt, o.P, B, MtB, o.V2, B, MtB, Mt, B, Mt);
    p.triA = 2;
    const int64_t sA[3] = {C * MtN, MtN, (int64_t)M * NRs}, sB[3] = {C * MtB, MtB, (int64_t)M * B};
    GemmParams q = blk_gemm(o.QPs + 4, NRs, sA, o.P, B, sB, o.W, B, sB, M, B, M, C, nblk);
    q.triA = 2;
    const int64_t wgs = (int64_t)SC * cdiv(B, 64) * (cdiv(Mt, 64) + nblk * cdiv(M, 64));
    static const int pair_fwd = [] { const char* e = getenv("VARGP_TN_PAIRFWD"); return e ? atoi(e) : 1; }();   // tuning aid
    if (pair_fwd && wgs <= 4096) {
      rc = launch_gemm_pair2(p, 1, 0, SC, q, 1, 0, SC * nblk, st, "tn_v2_gemm");
      if (rc) return rc;
    } else {
      rc = launch_gemm(p, 1, 0, SC, false, st, "tn_v2_gemm");
      if (rc) return rc;
      rc = launch_gemm(q, 1, 0, SC * nblk, false, st, "tn_w_gemm");
      if (rc) return rc;
    }
  }
  {
    const bool narrow = cdiv(B, 64) * SC < 512;
    const int nbx = cdiv(B, narrow ? 32 : 64), npd = nbx * SC, nkx = cdiv(M, kTnKlRows);
    const bool nomean = lik && d->no_var_mean && nblk > 1;
    hipLaunchKernelGGL(narrow ? tn_pdiag_kl_kernel<32> : tn_pdiag_kl_kernel<64>, dim3(npd + (lik ? nkx * SC : 0)), dim3(256), 0, st, o.P, o.W, o.V2, o.QPs, o.kd,
                       o.LL, d->rk_all, o.mu, o.var, lik ? d->scalars + 1 : nullptr, d->jitter, S, C, M, Mt, nblk, B, NRs,
                       nbx, npd, nkx, native ? d->rng_counter : nullptr, nomean ? 0 : 1);
    if (nomean) {   // the mean term of the KL at the n_v samples of u_<t (tn_nm_* above)
      const int NV = d->n_v, Ml = Mt - M;
      const int64_t nvl = (int64_t)SC * Ml * NV;
      hipLaunchKernelGGL(tn_nm_v_kernel, dim3(cdiv(nvl, 256)), dim3(256), 0, st, o.QPs, d->eps_u, o.nm_v, S, C, M, Mt, NRs, NV);
      hipLaunchKernelGGL(tn_nm_y1_kernel, dim3(cdiv(nvl, 256)), dim3(256), 0, st, o.TT, o.nm_v, o.nm_y1, (int64_t)SC, M, Mt, NV);
      hipLaunchKernelGGL(tn_nm_y2_kernel, dim3(cdiv((int64_t)SC * M, 4)), dim3(256), 0, st, o.Kall, o.nm_y1, o.nm_y2, (int64_t)SC, M, Mt, NV);
      hipLaunchKernelGGL(tn_nm_d_kernel, dim3(SC), dim3(256), 0, st, o.TT, o.QPs, o.nm_y2, o.nm_d, d->scalars + 1, S, M, Mt, NRs, NV);
    }
  }
  if (lik && !d->ext_lik) {       // (ext_lik: the caller evaluates the likelihood on the moments of ALL classes, include/vargp_hip.h)
    if (fused_softmax) {
      const int64_t total = (int64_t)S * F * B;
      hipLaunchKernelGGL(t0_softmax_kernel<16>, dim3(cdiv(total, 256)), dim3(256), 0, st, o.mu, o.var, eps_f, d->y,
                         d->scalars + 2, o.gmu, o.gvar, S, F, C, B);
    } else {
      rc = vargp_softmax_nll_fwd(o.mu, o.var, eps_f, d->y, d->scalars + 2, S, F, C, B, stream);
      if (rc) return rc;
    }
  }
  return check_launch("elbo_tn_fwd");
}

extern "C" int vargp_elbo_tn_bwd(const vargp_elbo_tn_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar,
                                 float* g_z, float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream) {
  int rc = check_tn(d, "elbo_tn_bwd");
  if (rc) return rc;
  VARGP_REQUIRE(seeds && g_z && g_u_mean && g_u_tril_vec && d->y && (d->defer_hyper || (g_log_mean && g_log_logvar)),
                "elbo_tn_bwd: null pointer");
  VARGP_REQUIRE(!d->forward_only, "elbo_tn_bwd: the program was carved forward_only");
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, F = d->F, nblk = d->nblk, SC = S * C;
  const TnWs o = carve_tn(d->ws, S, C, M, D, B, F, nblk);
  const int Mt = o.Mt, NRs = o.NRs;
  const int64_t MtMt = (int64_t)Mt * Mt, MtB = (int64_t)Mt * B, MtN = (int64_t)Mt * NRs;
  const bool native = d->eps_f == nullptr && !d->ext_lik;
  const bool fused_softmax = C <= 16 && !d->ext_lik;      // ext_lik: gmu / gvar arrive seeded, as from the generic kernel
  const float* eps_f = native ? o.eps_f : d->eps_f;
  const float* eps_theta = native ? o.eps_theta : d->eps_theta;

  const bool nomean = d->no_var_mean && nblk > 1;
  if (!fused_softmax && !d->ext_lik) {
    rc = vargp_softmax_nll_bwd(o.mu, o.var, eps_f, d->y, seeds + 2, o.gmu, o.gvar, S, F, C, B, stream);
    if (rc) return rc;
  }
  {
    const int npd = SC * Mt;
    const int nrest = cdiv((int64_t)npd * (NRs - 1), 256);
    const int64_t zc = o.r_uu - o.r_uf;
    const int nz = (int)std::min<int64_t>(64, cdiv(zc, 1024));
    hipLaunchKernelGGL(tn_bwd_head_kernel, dim3(npd + nrest + nz), dim3(256), 0, st, o.P, o.W, o.V2, o.QPs, o.gmu, o.gvar,
                       fused_softmax ? seeds + 2 : nullptr, seeds, o.gP, o.gQPs, o.gkd, d->jitter, S, M, Mt, B, NRs, npd,
                       nrest, o.r_uf, zc, 0, nomean ? 0 : 1);
  }
  if (nomean) {   // ep_var_mean = False: the KL's mean term back to ga / gH of every block (gQPs), keeping gy2 / gy1 for gT and gK
    const int NV = d->n_v, Ml = Mt - M;
    const int64_t nvl = (int64_t)SC * Ml * NV;
    hipLaunchKernelGGL(tn_nm_bwd1_kernel, dim3(SC), dim3(256), 0, st, o.TT, o.nm_d, seeds, o.nm_gy2, o.gQPs, S, M, Mt, NRs, NV);
    hipLaunchKernelGGL(tn_nm_bwd2_kernel, dim3(cdiv(nvl, 256)), dim3(256), 0, st, o.Kall, o.nm_gy2, o.nm_gy1, (int64_t)SC, M, Mt, NV);
    hipLaunchKernelGGL(tn_nm_bwd3_kernel, dim3(cdiv((int64_t)SC * Ml, 4)), dim3(256), 0, st, o.TT, o.nm_gy1, d->eps_u, o.gQPs, S, C, M, Mt,
                       NRs, NV);
  }
  float* gW = o.W;      // in place (tn_bwd_head_kernel)
  float* gV2 = o.V2;
  {  // W_i = H_i^T P_i:  gH_i += P_i gW_i^T (K-split, atomic accumulation on top of the KL term),  gP_i += H_i gW_i
    const int64_t sQ[3] = {C * MtN, MtN, (int64_t)M * NRs}, sP[3] = {C * MtB, MtB, (int64_t)M * B};
    GemmParams p = blk_gemm(o.P, B, sP, gW, B, sP, o.gQPs + 4, NRs, sQ, M, M, B, C, nblk);
    p.splitk = ksplit(B);
    if (p.splitk <= 1) { p.D = o.gQPs + 4; p.ldd = NRs; p.beta = 1.f; for (int i = 0; i < 3; ++i) p.sD[i] = sQ[i]; }
    GemmParams q = blk_gemm(o.QPs + 4, NRs, sQ, gW, B, sP, o.gP, B, sP, M, B, M, C, nblk);
    q.triA = 1; q.D = o.gP; q.ldd = B; q.beta = 1.f;
    const int64_t wgs = (int64_t)SC * nblk * (cdiv(M, 64) * cdiv(M, 64) * std::max(p.splitk, 1) + cdiv(M, 64) * cdiv(B, 64));
    if (wgs <= 4096) {     // mid-size: one launch for both
      rc = launch_gemm_pair2(p, 0, 1, SC * nblk, q, 0, 0, SC * nblk, st, "tn_gh_gp_gemm");
      if (rc) return rc;
    } else {               // each fills the chip by itself: own launches with the tile shape that suits them
      // (only tril(gH_i) is used: by tril(gH_i Lu_i^T) in gT and by the packed-vector gradient; the head kernel left
      //  zeros above the diagonal)
      p.splitk = 1; p.D = o.gQPs + 4; p.ldd = NRs; p.beta = 1.f; p.triC = 1;
      for (int i = 0; i < 3; ++i) p.sD[i] = sQ[i];
      rc = launch_gemm(p, 0, 1, SC * nblk, false, st, "tn_gh_gemm");
      if (rc) return rc;
      rc = launch_gemm(q, 0, 0, SC * nblk, false, st, "tn_gp_gemm");
      if (rc) return rc;
    }
  }
  {  // V2 = T^T P:  gP += T gV2
     // gT = tril(gP K_uf^T + P gV2^T)  (P = T K_uf and V2 = T^T P), then the diagonal blocks' share from the small products;
     // gK_uf = T^T gP and the parameter gradients of the current task, [g m_t | . | g Lu_t] = T_tt^T [ga_t | . | gH_t], do not
     // depend on gT.  Mid-size shapes (no product fills the chip alone) run the seven products as FOUR launches, ordered by what
     // each needs:   [gT = P gV2^T  ||  gP += T gV2]  ->  [gT += gP K_uf^T  ||  gK_uf = T^T gP]  ->  [gT_ii += gQP_i RK_i^T  ||  gRK_t]
     // (P gV2^T needs nothing of this segment, so it leads and the gP product is added on top of it; round 4 ran five launches)
    GemmParams v = flat_gemm(o.TT, Mt, MtMt, gV2, B, MtB, o.gP, B, MtB, Mt, B, Mt);
    v.triA = 1; v.D = o.gP; v.beta = 1.f;
    GemmParams p = flat_gemm(o.gP, B, MtB, o.Kuf, B, MtB, o.gT, Mt, MtMt, Mt, Mt, B);
    p.triC = 1;
    GemmParams q = flat_gemm(o.P, B, MtB, gV2, B, MtB, o.gT, Mt, MtMt, Mt, Mt, B);
    q.triC = 1;
    GemmParams ku = flat_gemm(o.TT, Mt, MtMt, o.gP, B, MtB, o.gKuf, B, MtB, Mt, B, Mt);
    ku.triA = 2;
    const int64_t off = (int64_t)(Mt - M) * Mt + (Mt - M);
    GemmParams rk = flat_gemm(o.TT + off, Mt, MtMt, o.gQPs + (int64_t)(Mt - M) * NRs, NRs, MtN, o.gRKt, NRs, (int64_t)M * NRs, M,
                              NRs, M);
    rk.triA = 2;
    const int64_t sQ[3] = {C * MtN, MtN, (int64_t)M * NRs}, sR[3] = {0, (int64_t)nblk * M * NRs, (int64_t)M * NRs},
                  sT[3] = {C * MtMt, MtMt, (int64_t)M * Mt + M};
    GemmParams r = blk_gemm(o.gQPs, NRs, sQ, d->rk_all, NRs, sR, o.gT, Mt, sT, M, M, NRs, C, nblk);
    r.triC = 1; r.D = o.gT; r.ldd = Mt; r.beta = 1.f;
    const int64_t wgs = (int64_t)SC * cdiv(Mt, 64) * (cdiv(Mt, 64) + cdiv(B, 64));
    static const int pair_bwd = [] { const char* e = getenv("VARGP_TN_PAIRBWD"); return e ? atoi(e) : 1; }();   // tuning aid
    if (pair_bwd && wgs <= 4096) {
      p.D = o.gT; p.beta = 1.f;           // on top of q, which leads
      rc = launch_gemm_pair2(q, 0, 1, SC, v, 0, 0, SC, st, "tn_gt_gemm");
      if (rc) return rc;
      rc = launch_gemm_pair2(p, 0, 1, SC, ku, 1, 0, SC, st, "tn_gt_gemm");
      if (rc) return rc;
      rc = launch_gemm_pair2(r, 0, 1, SC * nblk, rk, 1, 0, SC, st, "tn_gt_diag_gemm");
      if (rc) return rc;
    } else {
      q.D = o.gT; q.beta = 1.f;           // on top of p
      rc = launch_gemm(v, 0, 0, SC, false, st, "tn_gp_v2_gemm");
      if (rc) return rc;
      rc = launch_gemm(p, 0, 1, SC, false, st, "tn_gt_gemm");
      if (rc) return rc;
      rc = launch_gemm(q, 0, 1, SC, false, st, "tn_gt_gemm");
      if (rc) return rc;
      rc = launch_gemm(ku, 1, 0, SC, false, st, "tn_gkuf_gemm");
      if (rc) return rc;
      rc = launch_gemm(rk, 1, 0, SC, false, st, "tn_grk_gemm");
      if (rc) return rc;
      rc = launch_gemm(r, 0, 1, SC * nblk, false, st, "tn_gt_diag_gemm");
      if (rc) return rc;
    }
  }
  if (nomean) {   // ... and to the T_tt and T_<< blocks of gT (complete by now)
    const int NV = d->n_v, Ml = Mt - M;
    const int64_t n_tt = (int64_t)SC * M * M, n_ll = (int64_t)SC * Ml * Ml;
    hipLaunchKernelGGL(tn_nm_bwd4_kernel, dim3(cdiv(n_tt + n_ll, 256)), dim3(256), 0, st, o.nm_d, o.nm_y2, o.nm_v, o.nm_gy1, seeds, o.gT,
                       (int64_t)SC, S, M, Mt, NV, n_tt);
  }
  // Cholesky backward.  Only diag(L_tt) is used forward (log-determinant), so gL = diag(g / L_jj) on the current block:
  //   P_low = tril(L^T gL - gT T^T) = g I_t - tril(gT T^T)   (the lower triangle of L^T diag(.) is its diagonal)
  //   Smat  = (Phi(P_low) + Phi(P_low)^T) / 2 = -0.5 sym(tril(gT T^T)) + 0.5 g I_t,     gK = T^T Smat T
  float* Smat = reinterpret_cast<float*>(o.chol);
  float* tmp = Smat + SC * MtMt;
  {
    GemmParams p = flat_gemm(o.gT, Mt, MtMt, o.TT, Mt, MtMt, Smat, Mt, MtMt, Mt, Mt, Mt);
    p.alpha = -0.5f; p.triA = 1; p.triB = 2; p.triC = 2; p.symout = 1;
    p.diag_ptr = seeds + 1; p.diag_scale = 0.5f / (float)S; p.diag_from = Mt - M;     // + g / 2 on the current task's diagonal
    rc = launch_gemm(p, 0, 1, SC, false, st, "tn_chol_bwd1");
    if (rc) return rc;
    // gK = T^T (Smat T) is symmetric: its lower triangle needs tril(Smat T) only, and as T^T [.] the tiles of the lower
    // triangle are the ones with the SHORT K ranges (k >= row), 40 % of the work of the full product
    GemmParams q = flat_gemm(Smat, Mt, MtMt, o.TT, Mt, MtMt, tmp, Mt, MtMt, Mt, Mt, Mt);
    q.triB = 1; q.triC = 1;
    rc = launch_gemm(q, 0, 0, SC, false, st, "tn_chol_bwd2");
    if (rc) return rc;
    GemmParams r = flat_gemm(o.TT, Mt, MtMt, tmp, Mt, MtMt, o.gK, Mt, MtMt, Mt, Mt, Mt);
    r.triA = 2; r.triB = 1; r.triC = 2; r.symout = 1;
    rc = launch_gemm(r, 1, 0, SC, false, st, "tn_chol_bwd3");
    if (rc) return rc;
  }
  if (nomean) {   // ... and the direct dependence of prior_mu on K_t<
    const int NV = d->n_v, Ml = Mt - M;
    hipLaunchKernelGGL(tn_nm_bwd5_kernel, dim3(cdiv((int64_t)SC * M * Ml, 256)), dim3(256), 0, st, o.nm_gy2, o.nm_y1, o.gK, (int64_t)SC, M,
                       Mt, NV);
  }
  // kernel matrices -> theta, z  (the fused passes of the first-task program, elbo_shared.h):
  //   W = gK o K for both kernel matrices in one launch (K_uf in place on gK_uf; K_all: gK is symmetric, W + W^T = 2 W),
  //   both W.Y products, one finalisation for the inducing-point and the minibatch side
  const int64_t zrows = (int64_t)C * Mt;
  {
    const int gx = cdiv(B, 256), gy = cdiv(zrows, kWRows), nuf = gx * gy * S;
    const int nuu = SC * cdiv(Mt, kUuRows);
    hipLaunchKernelGGL(t0_w_kernel, dim3(nuf + nuu), dim3(256), 0, st, o.Kuf, o.gKuf, o.Kall, o.gK, o.Wuu, o.r_uu, o.r_uf,
                       o.c_uf, o.gtheta, S, C, Mt, B, D, 0, B, gx, gy, nuf, nuu, (const float*)nullptr, (const float*)nullptr,
                       seeds, (float*)nullptr, 1);
  }
  {
    GemmParams p0{}, p1{};
    p0.A = o.Wuu; p0.B = d->z_all; p0.C = o.Puu;
    p0.M = Mt; p0.N = D; p0.K = Mt; p0.lda = Mt; p0.ldb = D; p0.ldc = D;
    p0.nb1 = C; p0.nb2 = 1;
    p0.sA[0] = C * MtMt; p0.sA[1] = MtMt;
    p0.sB[1] = (int64_t)Mt * D;
    p0.sC[0] = zrows * D; p0.sC[1] = (int64_t)Mt * D;
    p0.alpha = 1.f;
    p1.A = o.gKuf; p1.B = d->x; p1.C = o.Puf;
    p1.M = C * Mt; p1.N = D; p1.K = B; p1.lda = B; p1.ldb = D; p1.ldc = D;
    p1.nb1 = 1; p1.nb2 = 1;
    p1.sA[0] = C * MtB;
    p1.sC[0] = zrows * D;
    p1.alpha = 1.f;
    const int64_t wgs = (int64_t)cdiv(Mt, 64) * cdiv(D, 64) * SC + (int64_t)cdiv(C * Mt, 64) * cdiv(D, 64) * S;
    if (wgs <= 4096) {     // mid-size: neither fills the chip alone, one launch (64^3 tiles)
      rc = launch_gemm_pair(p0, SC, p1, S, 0, 0, false, st, "rbf_kuu_bwd_gemm", "rbf_kuf_bwd_gemm");
      if (rc) return rc;
    } else {
      rc = launch_gemm(p0, 0, 0, SC, false, st, "rbf_kuu_bwd_gemm");
      if (rc) return rc;
      rc = launch_gemm(p1, 0, 0, S, false, st, "rbf_kuf_bwd_gemm");
      if (rc) return rc;
    }
  }
  {
    const int nzy = cdiv(zrows, kFinRows), nxy = cdiv(B, kFinRows);
    hipLaunchKernelGGL(t0_final_kernel, dim3(cdiv(D, 64), nzy + nxy), dim3(256), 0, st, d->z_all, d->x, o.r_uu, o.r_uf, o.c_uf,
                       o.Puu, o.Puf, o.w, o.gz_all, o.gtheta, zrows, (int64_t)B, D, o.Dp, S, nzy);
  }
  {
    const int nun = cdiv((int64_t)C * M * (M + 1), 256);
    hipLaunchKernelGGL(tn_unpack_kernel, dim3(nun + cdiv((int64_t)C * M * D, 256)), dim3(256), 0, st, o.gRKt, d->u_tril_vec,
                       d->rk_all + (int64_t)(nblk - 1) * M * NRs, seeds, o.gz_all, g_u_mean, g_u_tril_vec, g_z, S, C, M, Mt, D,
                       NRs, nblk, nun);
  }
  if (!d->defer_hyper)
    hipLaunchKernelGGL(t0_hyper_bwd_kernel, dim3(cdiv(D + 1, 256)), dim3(256), 0, st, d->log_mean, d->log_logvar,
                       d->prior_log_mean, d->prior_log_logvar, eps_theta, o.gtheta, o.g2, o.gkd, seeds, g_log_mean,
                       g_log_logvar, S, C, D + 1, d->map_est);
  return check_launch("elbo_tn_bwd");
}

extern "C" int vargp_elbo_tn_hyper_desc(const vargp_elbo_tn_desc* d, const float* seeds, vargp_hyper_grad_desc* out) {
  int rc = check_tn(d, "elbo_tn_hyper_desc");
  if (rc) return rc;
  VARGP_REQUIRE(seeds && out && !d->forward_only, "elbo_tn_hyper_desc: bad arguments");
  const TnWs o = carve_tn(d->ws, d->S, d->C, d->M, d->D, d->B, d->F, d->nblk);
  out->log_mean = d->log_mean; out->log_logvar = d->log_logvar;
  out->prior_log_mean = d->prior_log_mean; out->prior_log_logvar = d->prior_log_logvar;
  out->eps_theta = (d->eps_f == nullptr && !d->ext_lik) ? o.eps_theta : d->eps_theta;
  out->gtheta = o.gtheta; out->g2 = o.g2; out->gkd = o.gkd; out->seeds = seeds;
  out->S = d->S; out->C = d->C; out->D1 = d->D + 1; out->map_est = d->map_est;
  return VARGP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// N-tiled ELBO (BASELINE config 5: N = 1e6, M = 2048): loss AND gradient over a data set swept in minibatch tiles, with
// everything that does not depend on the data -- kernel matrix of the inducing points, its factorisation, the small
// products, the KL -- computed ONCE (begin), a forward + partial backward per tile that only accumulates (tile), and the
// Cholesky / kernel-matrix backward of the accumulated gradients at the end (end).  The nll seed multiplies every tile, so
// the seeds are needed from the first tile on.  Same kernels as the one-minibatch program.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int vargp_elbo_tn_begin(const vargp_elbo_tn_desc* d, vargp_stream_t stream) {
  int rc = check_tn(d, "elbo_tn_begin", true);
  if (rc) return rc;
  VARGP_REQUIRE(d->D > kRbfDirectD, "elbo_tn_begin: the tiled ELBO needs D > %d (MFMA distance path)", kRbfDirectD);
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, B = d->B, F = d->F, nblk = d->nblk, SC = S * C;
  const bool fwd_only = d->forward_only != 0;      // predictive sweep (VARGP.predict(x, tile=)): no accumulators
  const TnWs o = carve_tn(d->ws, S, C, M, D, B, F, nblk, fwd_only);
  const int Mt = o.Mt, NRs = o.NRs;
  const int64_t MtMt = (int64_t)Mt * Mt, MtN = (int64_t)Mt * NRs, zrows = (int64_t)C * Mt;
  {
    TnProArgs a{};
    a.mean = d->log_mean; a.logvar = d->log_logvar; a.pmean = d->prior_log_mean; a.plogvar = d->prior_log_logvar;
    a.eps_theta = d->eps_theta; a.vec = d->u_tril_vec; a.u_mean = d->u_mean; a.z = d->z;
    a.theta = o.theta; a.g2 = o.g2; a.kd = o.kd; a.scalars = d->scalars; a.bump = d->bump;
    a.rk_last = d->rk_all + (int64_t)(nblk - 1) * M * NRs; a.z_all = d->z_all;
    a.info = d->info; a.ninfo = SC;
    a.zero_begin = o.gmu; a.zero_count = 0;
    a.S = S; a.C = C; a.M = M; a.D = D; a.Mt = Mt; a.NRs = NRs; a.nblk = nblk; a.map_est = d->map_est;
    a.nzero_blocks = 0;
    const bool native_theta = d->eps_theta == nullptr && !d->map_est;
    if (native_theta) {   // theta noise from the generator (the per-tile likelihood noise is drawn by the tile calls)
      VARGP_REQUIRE(d->rng_counter, "elbo_tn_begin: native noise needs rng_counter");
      a.native = 1; a.seed = d->rng_seed; a.rng_counter = d->rng_counter;
      a.g0_theta = (int64_t)d->rng_sample_offset * (D + 1);
      a.eps_theta_out = o.eps_theta; a.nrng_blocks = 0; a.n_f = 0;
    }
    a.npack_blocks = cdiv((int64_t)C * M * NRs, 256);
    const int grid = 1 + S + a.npack_blocks + cdiv((int64_t)C * M * D, 256);
    hipLaunchKernelGGL(tn_prologue_kernel, dim3(grid), dim3(256), 0, st, a);
  }
  // accumulators of the sweep
  if (!fwd_only) {
    zero_async(o.gT, sizeof(float) * SC * MtMt, st);
    zero_async(o.gQPs, sizeof(float) * SC * MtN, st);
    zero_async(o.gkd, sizeof(float) * SC, st);
    zero_async(o.r_uf, sizeof(float) * (size_t)(o.r_uu - o.r_uf), st);
    zero_async(o.Puf, sizeof(float) * SC * Mt * D, st);
  }
  rc = rbf_prep_norm_launch(o.theta, d->z_all, zrows, nullptr, 0, o.w, o.g2, o.na, o.nb, S, D, o.Dp, st, nullptr, o.zs);
  if (rc) return rc;
  {
    GemmParams p0{};
    p0.A = o.zs; p0.B = d->z_all; p0.C = o.Kall;        // A pre-scaled by the norm pass: no per-k scaling in the main loop
    p0.M = Mt; p0.N = Mt; p0.K = D; p0.lda = D; p0.ldb = D; p0.ldc = Mt;
    p0.nb1 = C; p0.nb2 = 1;
    p0.sA[0] = zrows * D; p0.sA[1] = (int64_t)Mt * D; p0.sB[1] = (int64_t)Mt * D;
    p0.sC[0] = C * MtMt; p0.sC[1] = MtMt;
    p0.alpha = 1.f;
    p0.kscale = nullptr; p0.ks_ld = o.Dp; p0.g2 = o.g2;
    p0.na = o.na; p0.sNa[0] = zrows; p0.sNa[1] = Mt;
    p0.nbv = o.na; p0.sNb[0] = zrows; p0.sNb[1] = Mt;
    p0.same_xy = 1; p0.triC = 2; p0.symout = 1;
    rc = launch_gemm(p0, 0, 1, SC, true, st, "rbf_kuu_gemm");
    if (rc) return rc;
  }
  rc = chol_inv_fwd_impl(o.Kall, d->jitter, o.LL, o.TT, nullptr, d->info, SC, Mt, o.chol, o.chol_bytes, false, st);
  if (rc) return rc;
  {
    const int64_t sA[3] = {C * MtMt, MtMt, (int64_t)M * Mt + M}, sB[3] = {0, (int64_t)nblk * M * NRs, (int64_t)M * NRs},
                  sC[3] = {C * MtN, MtN, (int64_t)M * NRs};
    GemmParams p = blk_gemm(o.TT, Mt, sA, d->rk_all, NRs, sB, o.QPs, NRs, sC, M, NRs, M, C, nblk);
    p.triA = 1;
    rc = launch_gemm(p, 0, 0, SC * nblk, false, st, "tn_small_gemm");
    if (rc) return rc;
  }
  {   // the KL (data-independent): the KL role of the moments kernel alone
    const int nkx = cdiv(M, kTnKlRows);
    hipLaunchKernelGGL(tn_pdiag_kl_kernel<64>, dim3(nkx * SC), dim3(256), 0, st, o.P, o.W, o.V2, o.QPs, o.kd, o.LL, d->rk_all, o.mu,
                       o.var, d->scalars + 1, d->jitter, S, C, M, Mt, nblk, B, NRs, 1, 0, nkx, (uint32_t*)nullptr, 1);
  }
  return check_launch("elbo_tn_begin");
}

// One minibatch tile: x (Bt, D), y (Bt), Bt <= d->B; eps_f (S, F, C, Bt) or NULL (native noise, one generator step per tile).
// Adds the tile's nll to scalars[2] and its share of every gradient to the accumulators.
extern "C" int vargp_elbo_tn_tile(const vargp_elbo_tn_desc* d, const float* seeds, const float* x, const int64_t* y,
                                  const float* eps_f_in, int Bt, vargp_stream_t stream) {
  VARGP_REQUIRE(d && x && d->ws && Bt > 0 && Bt <= d->B, "elbo_tn_tile: bad arguments");
  const bool moments_only = y == nullptr;          // predictive sweep: mu, var (S, C, Bt) of this tile, nothing else
  VARGP_REQUIRE(moments_only || seeds, "elbo_tn_tile: seeds missing");
  VARGP_REQUIRE(moments_only || !d->forward_only, "elbo_tn_tile: a forward_only program takes y == NULL tiles only");
  VARGP_REQUIRE(moments_only || eps_f_in || d->rng_counter, "elbo_tn_tile: native noise needs rng_counter");
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, F = d->F, nblk = d->nblk, SC = S * C, B = Bt;
  const TnWs o = carve_tn(d->ws, S, C, M, D, d->B, F, nblk, d->forward_only != 0);
  const int Mt = o.Mt, NRs = o.NRs;
  const int64_t MtMt = (int64_t)Mt * Mt, MtB = (int64_t)Mt * B, MtN = (int64_t)Mt * NRs, zrows = (int64_t)C * Mt;
  const bool native = eps_f_in == nullptr;
  const bool fused_softmax = C <= 16;
  const float* eps_f = native ? o.eps_f : eps_f_in;
  int rc;
  if (!moments_only) {
    const int64_t n0 = o.Kall - o.gmu, n1 = (int64_t)S * d->B;          // gmu | gvar, c_uf
    const int64_t n_f = (int64_t)S * F * C * B;
    const int nzero = (int)std::min<int64_t>(64, cdiv(n0 + n1, 1024));
    const int nrng = native ? (int)std::min<int64_t>(512, cdiv(n_f + 7, 1024)) : 0;
    hipLaunchKernelGGL(tn_tile_prep_kernel, dim3(nzero + nrng), dim3(256), 0, st, o.gmu, n0, o.c_uf, n1, nzero, native ? 1 : 0,
                       d->rng_seed, d->rng_counter, (int64_t)d->rng_sample_offset * F * C * B, n_f, o.eps_f);
  }
  rc = rbf_prep_norm_launch(o.theta, nullptr, 0, x, B, o.w, o.g2, o.na, o.nb, S, D, o.Dp, st, o.xs);
  if (rc) return rc;
  {
    GemmParams pf{};
    pf.A = d->z_all; pf.B = o.xs; pf.C = o.Kuf;
    pf.M = C * Mt; pf.N = B; pf.K = D; pf.lda = D; pf.ldb = D; pf.ldc = B;
    pf.nb1 = 1; pf.nb2 = 1;
    pf.sB[0] = (int64_t)B * D;
    pf.sC[0] = (int64_t)C * MtB;
    pf.alpha = 1.f;
    pf.kscale = nullptr; pf.ks_ld = o.Dp; pf.g2 = o.g2;      // pre-scaled B operand
    pf.na = o.na; pf.sNa[0] = zrows;
    pf.nbv = o.nb; pf.sNb[0] = B;
    rc = launch_gemm(pf, 0, 1, S, true, st, "rbf_kuf_gemm");
    if (rc) return rc;
  }
  {
    GemmParams p = flat_gemm(o.TT, Mt, MtMt, o.Kuf, B, MtB, o.P, B, MtB, Mt, B, Mt);
    p.triA = 1;
    rc = launch_gemm(p, 0, 0, SC, false, st, "tn_p_gemm");
    if (rc) return rc;
    GemmParams q = flat_gemm(o.TT, Mt, MtMt, o.P, B, MtB, o.V2, B, MtB, Mt, B, Mt);
    q.triA = 2;
    rc = launch_gemm(q, 1, 0, SC, false, st, "tn_v2_gemm");
    if (rc) return rc;
    const int64_t sA[3] = {C * MtN, MtN, (int64_t)M * NRs}, sB[3] = {C * MtB, MtB, (int64_t)M * B};
    GemmParams r = blk_gemm(o.QPs + 4, NRs, sA, o.P, B, sB, o.W, B, sB, M, B, M, C, nblk);
    r.triA = 2;
    rc = launch_gemm(r, 1, 0, SC * nblk, false, st, "tn_w_gemm");
    if (rc) return rc;
  }
  {
    const bool narrow = cdiv(B, 64) * SC < 512;
    const int nbx = cdiv(B, narrow ? 32 : 64), npd = nbx * SC;
    hipLaunchKernelGGL(narrow ? tn_pdiag_kl_kernel<32> : tn_pdiag_kl_kernel<64>, dim3(npd), dim3(256), 0, st, o.P, o.W, o.V2, o.QPs, o.kd, o.LL, d->rk_all, o.mu, o.var,
                       (float*)nullptr, d->jitter, S, C, M, Mt, nblk, B, NRs, nbx, npd, 1,
                       (native && !moments_only) ? d->rng_counter : nullptr, 1);
  }
  if (moments_only) return check_launch("elbo_tn_tile");
  if (fused_softmax) {
    const int64_t total = (int64_t)S * F * B;
    hipLaunchKernelGGL(t0_softmax_kernel<16>, dim3(cdiv(total, 256)), dim3(256), 0, st, o.mu, o.var, eps_f, y, d->scalars + 2,
                       o.gmu, o.gvar, S, F, C, B);
  } else {
    // the generic kernel overwrites nll: accumulate through a scratch scalar is not worth a kernel -- C <= 16 covers the configs
    VARGP_REQUIRE(false, "elbo_tn_tile: more than 16 classes is not supported by the tiled ELBO");
  }
  // ---- the tile's share of the backward -------------------------------------------------------------------------------
  {
    const int npd = SC * Mt;
    hipLaunchKernelGGL(tn_bwd_head_kernel, dim3(npd), dim3(256), 0, st, o.P, o.W, o.V2, o.QPs, o.gmu, o.gvar, seeds + 2, seeds, o.gP,
                       o.gQPs, o.gkd, d->jitter, S, M, Mt, B, NRs, npd, 0, (float*)nullptr, (int64_t)0, 1, 1);
  }
  float* gW = o.W;
  float* gV2 = o.V2;
  {
    const int64_t sQ[3] = {C * MtN, MtN, (int64_t)M * NRs}, sP[3] = {C * MtB, MtB, (int64_t)M * B};
    GemmParams p = blk_gemm(o.P, B, sP, gW, B, sP, o.gQPs + 4, NRs, sQ, M, M, B, C, nblk);
    p.D = o.gQPs + 4; p.ldd = NRs; p.beta = 1.f;
    for (int i = 0; i < 3; ++i) p.sD[i] = sQ[i];
    rc = launch_gemm(p, 0, 1, SC * nblk, false, st, "tn_gh_gemm");
    if (rc) return rc;
    GemmParams q = blk_gemm(o.QPs + 4, NRs, sQ, gW, B, sP, o.gP, B, sP, M, B, M, C, nblk);
    q.triA = 1; q.D = o.gP; q.ldd = B; q.beta = 1.f;
    rc = launch_gemm(q, 0, 0, SC * nblk, false, st, "tn_gp_gemm");
    if (rc) return rc;
  }
  {
    GemmParams p = flat_gemm(o.TT, Mt, MtMt, gV2, B, MtB, o.gP, B, MtB, Mt, B, Mt);
    p.triA = 1; p.D = o.gP; p.beta = 1.f;
    rc = launch_gemm(p, 0, 0, SC, false, st, "tn_gp_v2_gemm");
    if (rc) return rc;
    GemmParams q = flat_gemm(o.gP, B, MtB, o.Kuf, B, MtB, o.gT, Mt, MtMt, Mt, Mt, B);
    q.triC = 1; q.D = o.gT; q.beta = 1.f;
    rc = launch_gemm(q, 0, 1, SC, false, st, "tn_gt_gemm");
    if (rc) return rc;
    GemmParams r = flat_gemm(o.P, B, MtB, gV2, B, MtB, o.gT, Mt, MtMt, Mt, Mt, B);
    r.triC = 1; r.D = o.gT; r.beta = 1.f;
    rc = launch_gemm(r, 0, 1, SC, false, st, "tn_gt_gemm");
    if (rc) return rc;
    GemmParams u = flat_gemm(o.TT, Mt, MtMt, o.gP, B, MtB, o.gKuf, B, MtB, Mt, B, Mt);
    u.triA = 2;
    rc = launch_gemm(u, 1, 0, SC, false, st, "tn_gkuf_gemm");
    if (rc) return rc;
  }
  {   // W_uf = gK_uf o K_uf in place, row sums (accumulating over the tiles), column sums (this tile's)
    const int gx = cdiv(B, 256), gy = cdiv(zrows, kWRows), nuf = gx * gy * S;
    hipLaunchKernelGGL(t0_w_kernel, dim3(nuf), dim3(256), 0, st, o.Kuf, o.gKuf, o.Kall, o.gK, o.Wuu, o.r_uu, o.r_uf, o.c_uf, o.gtheta,
                       S, C, Mt, B, D, 0, B, gx, gy, nuf, 0, (const float*)nullptr, (const float*)nullptr, seeds, (float*)nullptr, 1);
    GemmParams p1{};
    p1.A = o.gKuf; p1.B = x; p1.C = o.Puf; p1.D = o.Puf;
    p1.M = C * Mt; p1.N = D; p1.K = B; p1.lda = B; p1.ldb = D; p1.ldc = D; p1.ldd = D;
    p1.nb1 = 1; p1.nb2 = 1;
    p1.sA[0] = C * MtB;
    p1.sC[0] = zrows * D; p1.sD[0] = zrows * D;
    p1.alpha = 1.f; p1.beta = 1.f;
    rc = launch_gemm(p1, 0, 0, S, false, st, "rbf_kuf_bwd_gemm");
    if (rc) return rc;
    const int nxy = cdiv(B, kFinRows);      // minibatch side only: gtheta += w sum_n c_uf x^2
    hipLaunchKernelGGL(t0_final_kernel, dim3(cdiv(D, 64), nxy), dim3(256), 0, st, d->z_all, x, o.r_uu, o.r_uf, o.c_uf, o.Puu,
                       o.Puf, o.w, o.gz_all, o.gtheta, zrows, (int64_t)B, D, o.Dp, S, 0);
  }
  return check_launch("elbo_tn_tile");
}

extern "C" int vargp_elbo_tn_end(const vargp_elbo_tn_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar,
                                 float* g_z, float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream) {
  VARGP_REQUIRE(d && d->ws && seeds && g_log_mean && g_log_logvar && g_z && g_u_mean && g_u_tril_vec, "elbo_tn_end: null pointer");
  hipStream_t st = as_stream(stream);
  const int S = d->S, C = d->C, M = d->M, D = d->D, F = d->F, nblk = d->nblk, SC = S * C;
  const TnWs o = carve_tn(d->ws, S, C, M, D, d->B, F, nblk);
  const int Mt = o.Mt, NRs = o.NRs;
  const int64_t MtMt = (int64_t)Mt * Mt, MtN = (int64_t)Mt * NRs, zrows = (int64_t)C * Mt;
  const float* eps_theta = d->eps_theta ? d->eps_theta : o.eps_theta;
  int rc;
  {
    const int64_t total = (int64_t)SC * M * NRs;
    hipLaunchKernelGGL(tn_kl_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, o.QPs, o.gQPs, seeds, S, M, Mt, NRs, total);
  }
  {
    const int64_t sQ[3] = {C * MtN, MtN, (int64_t)M * NRs}, sR[3] = {0, (int64_t)nblk * M * NRs, (int64_t)M * NRs},
                  sT[3] = {C * MtMt, MtMt, (int64_t)M * Mt + M};
    GemmParams r = blk_gemm(o.gQPs, NRs, sQ, d->rk_all, NRs, sR, o.gT, Mt, sT, M, M, NRs, C, nblk);
    r.triC = 1; r.D = o.gT; r.ldd = Mt; r.beta = 1.f;
    rc = launch_gemm(r, 0, 1, SC * nblk, false, st, "tn_gt_diag_gemm");
    if (rc) return rc;
    const int64_t off = (int64_t)(Mt - M) * Mt + (Mt - M);
    GemmParams p = flat_gemm(o.TT + off, Mt, MtMt, o.gQPs + (int64_t)(Mt - M) * NRs, NRs, MtN, o.gRKt, NRs, (int64_t)M * NRs, M,
                             NRs, M);
    p.triA = 2;
    rc = launch_gemm(p, 1, 0, SC, false, st, "tn_grk_gemm");
    if (rc) return rc;
  }
  float* Smat = reinterpret_cast<float*>(o.chol);
  float* tmp = Smat + SC * MtMt;
  {
    GemmParams p = flat_gemm(o.gT, Mt, MtMt, o.TT, Mt, MtMt, Smat, Mt, MtMt, Mt, Mt, Mt);
    p.alpha = -0.5f; p.triA = 1; p.triB = 2; p.triC = 2; p.symout = 1;
    p.diag_ptr = seeds + 1; p.diag_scale = 0.5f / (float)S; p.diag_from = Mt - M;     // + g / 2 on the current task's diagonal
    rc = launch_gemm(p, 0, 1, SC, false, st, "tn_chol_bwd1");
    if (rc) return rc;
    // gK = T^T (Smat T) is symmetric: its lower triangle needs tril(Smat T) only, and as T^T [.] the tiles of the lower
    // triangle are the ones with the SHORT K ranges (k >= row), 40 % of the work of the full product
    GemmParams q = flat_gemm(Smat, Mt, MtMt, o.TT, Mt, MtMt, tmp, Mt, MtMt, Mt, Mt, Mt);
    q.triB = 1; q.triC = 1;
    rc = launch_gemm(q, 0, 0, SC, false, st, "tn_chol_bwd2");
    if (rc) return rc;
    GemmParams r = flat_gemm(o.TT, Mt, MtMt, tmp, Mt, MtMt, o.gK, Mt, MtMt, Mt, Mt, Mt);
    r.triA = 2; r.triB = 1; r.triC = 2; r.symout = 1;
    rc = launch_gemm(r, 1, 0, SC, false, st, "tn_chol_bwd3");
    if (rc) return rc;
  }
  {   // K_all: W + W^T = 2 gK o K, its row sums; the W.z product; the inducing-point side of the finalisation
    const int nuu = SC * cdiv(Mt, kUuRows);
    hipLaunchKernelGGL(t0_w_kernel, dim3(nuu), dim3(256), 0, st, o.Kuf, o.gKuf, o.Kall, o.gK, o.Wuu, o.r_uu, o.r_uf, o.c_uf, o.gtheta,
                       S, C, Mt, d->B, D, 0, d->B, 1, 1, 0, nuu, (const float*)nullptr, (const float*)nullptr, seeds,
                       (float*)nullptr, 1);
    GemmParams p0{};
    p0.A = o.Wuu; p0.B = d->z_all; p0.C = o.Puu;
    p0.M = Mt; p0.N = D; p0.K = Mt; p0.lda = Mt; p0.ldb = D; p0.ldc = D;
    p0.nb1 = C; p0.nb2 = 1;
    p0.sA[0] = C * MtMt; p0.sA[1] = MtMt;
    p0.sB[1] = (int64_t)Mt * D;
    p0.sC[0] = zrows * D; p0.sC[1] = (int64_t)Mt * D;
    p0.alpha = 1.f;
    rc = launch_gemm(p0, 0, 0, SC, false, st, "rbf_kuu_bwd_gemm");
    if (rc) return rc;
    const int nzy = cdiv(zrows, kFinRows);
    hipLaunchKernelGGL(t0_final_kernel, dim3(cdiv(D, 64), nzy), dim3(256), 0, st, d->z_all, (const float*)nullptr, o.r_uu, o.r_uf,
                       o.c_uf, o.Puu, o.Puf, o.w, o.gz_all, o.gtheta, zrows, (int64_t)0, D, o.Dp, S, nzy);
  }
  {
    const int nun = cdiv((int64_t)C * M * (M + 1), 256);
    hipLaunchKernelGGL(tn_unpack_kernel, dim3(nun + cdiv((int64_t)C * M * D, 256)), dim3(256), 0, st, o.gRKt, d->u_tril_vec,
                       d->rk_all + (int64_t)(nblk - 1) * M * NRs, seeds, o.gz_all, g_u_mean, g_u_tril_vec, g_z, S, C, M, Mt, D,
                       NRs, nblk, nun);
  }
  hipLaunchKernelGGL(t0_hyper_bwd_kernel, dim3(cdiv(D + 1, 256)), dim3(256), 0, st, d->log_mean, d->log_logvar,
                     d->prior_log_mean, d->prior_log_logvar, eps_theta, o.gtheta, o.g2, o.gkd, seeds, g_log_mean,
                     g_log_logvar, S, C, D + 1, d->map_est);
  return check_launch("elbo_tn_end");
}
