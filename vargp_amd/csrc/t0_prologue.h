// Forward prologue of the first-task program (elbo_t0.hip) as device functions, so that its roles can share a launch with
// the K-split K_uu product (gemm.hip: t0_pro_kuu_kernel) as well as run alone (elbo_t0.hip: t0_prologue_kernel).
#pragma once
#include "common.h"
#include "elbo_rng.h"

namespace vargp {

struct ProArgs {
  const float *mean, *logvar, *pmean, *plogvar, *eps_theta, *vec;
  float *theta, *w, *g2, *kd, *Lu, *Su, *scalars, *zero_begin, *bump;
  // when the factorisation writes L_S into RK itself, the prologue writes the other small columns (no pack launch)
  float* RK;
  const float* u_mean;
  int NR, LD;
  int32_t* info;
  int64_t zero_count, Dp;
  int S, C, M, D, ninfo, map_est, nzero_blocks;
  // native noise (eps_theta == eps_f == NULL in the descriptor): the prologue draws it
  int native, nrng_blocks;
  uint64_t seed;
  const uint32_t* rng_counter;
  int64_t g0_theta, g0_f, n_f;
  float *eps_theta_out, *eps_f_out;
  // accumulators of the BACKWARD that need no seed to be cleared (elbo_t0.hip: the LDS-resident backward then starts without a
  // head launch of its own)
  ZeroJobs zero;
  int su_in_chain;      // S_u = Lu Lu^T is formed on the matrix core by its factorising workgroup (CholExtra::su_Lu): the Lu role
                        // here writes Lu / RK only
};

// theta[s, d] = mean_d + eps[s, d] exp(logvar_d / 2) (kernels.py:62-68; MAP: mean_d): a pure function of the parameters and
// the noise, so that any workgroup that needs 1/sigma^2 can evaluate it itself instead of waiting for a launch that does.
// keep: store a natively drawn eps for the backward (one caller per element does).
__device__ __forceinline__ float t0_theta_at(const ProArgs& a, int s, int d, bool keep) {
  if (a.map_est) return a.mean[d];
  const int D1 = a.D + 1;
  float e;
  if (a.native) {
    e = normal1(a.seed, kStreamTheta, (uint64_t)(a.g0_theta + (int64_t)s * D1 + d), a.rng_counter[0]);
    if (keep) a.eps_theta_out[s * D1 + d] = e;
  } else {
    e = a.eps_theta[s * D1 + d];
  }
  return a.mean[d] + e * expf(0.5f * a.logvar[d]);
}

// The same for U entries d0, d0 + 256, ... of one thread, every load of the batch issued BEFORE the first use (entries past D are
// clamped and come out as theta[D]; the caller masks).  t0_theta_at in a loop is one memory round trip per iteration -- the
// loads sit behind the map_est / native branches -- and every workgroup of the front launch that needs 1 / sigma^2 paid two to
// four of them before its own loads could start (t0_pro_kuu_kernel: 20 us of which the product is 7).
template <int U>
__device__ __forceinline__ void t0_theta_batch(const ProArgs& a, int s, int d0, bool keep, float (&out)[U]) {
  const int D1 = a.D + 1;
  float mv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) mv[u] = a.mean[min(d0 + 256 * u, a.D)];
  if (a.map_est) {
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = mv[u];
    return;
  }
  float lv[U], ev[U];
#pragma unroll
  for (int u = 0; u < U; ++u) lv[u] = a.logvar[min(d0 + 256 * u, a.D)];
  if (a.native) {
    const uint32_t step = a.rng_counter[0];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = min(d0 + 256 * u, a.D);
      ev[u] = normal1(a.seed, kStreamTheta, (uint64_t)(a.g0_theta + (int64_t)s * D1 + d), step);
      if (keep && d0 + 256 * u < D1) a.eps_theta_out[s * D1 + d] = ev[u];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) ev[u] = a.eps_theta[s * D1 + min(d0 + 256 * u, a.D)];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) out[u] = mv[u] + ev[u] * expf(0.5f * lv[u]);
}

// Multi-role prologue, role by block index:
//   block 0            kl_hypers (kernels.py:70-77) -> scalars[0]; scalars[1..2] = 0 (kl_u, nll accumulate); info = 0;
//                      *bump += 1 if the caller asked for it
//   blocks 1..S        theta_s = mean + eps_s exp(logvar/2) (kernels.py:62-68); w_s = exp(-2 theta), g2_s = exp(2 theta_D)
//   next nzero_blocks  zero-fill of the softmax-gradient accumulators
//   next nrng_blocks   (native noise only) the likelihood noise eps_f; eps_theta is drawn inline by blocks 1..S
//   rest               Lu = vec2tril(u_tril_vec) (gp_utils.py:22-49) and S_u = Lu Lu^T straight from the packed vector
__device__ __forceinline__ void t0_prologue_body(const ProArgs& a, const int blk, float* __restrict__ red /* 4 floats of LDS */) {
  const int tid = threadIdx.x;
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
    for (int d0 = tid; d0 < D1 || d0 < a.Dp; d0 += 1024) {      // four entries per thread and round trip
      float tb[4];
      t0_theta_batch<4>(a, s, d0, true, tb);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = d0 + 256 * u;
        const float t = d < D1 ? tb[u] : 0.f;
        if (d < D1) a.theta[s * D1 + d] = t;
        if (d < a.Dp) a.w[s * a.Dp + d] = d < a.D ? expf(-2.f * t) : 0.f;
        if (d == a.D) {
          const float g = expf(2.f * t);
          a.g2[s] = g;
          for (int c = 0; c < a.C; ++c) a.kd[s * a.C + c] = g;
        }
      }
    }
    return;
  }
  if (blk <= a.S + a.nzero_blocks) {
    for (int64_t i = (int64_t)(blk - a.S - 1) * 256 + tid; i < a.zero_count; i += (int64_t)a.nzero_blocks * 256)
      a.zero_begin[i] = 0.f;
    zero_jobs_role(a.zero, blk - a.S - 1, a.nzero_blocks);
    return;
  }
  if (blk <= a.S + a.nzero_blocks + a.nrng_blocks) {
    // likelihood noise: one Philox group (4 normals) per thread, groups aligned to the GLOBAL element index
    const uint32_t step = a.rng_counter[0];
    const int64_t gfirst = a.g0_f >> 2, glast = (a.g0_f + a.n_f + 3) >> 2;
    for (int64_t G = gfirst + (int64_t)(blk - a.S - a.nzero_blocks - 1) * 256 + tid; G < glast;
         G += (int64_t)a.nrng_blocks * 256) {
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
  const int64_t e = (int64_t)(blk - 1 - a.S - a.nzero_blocks - a.nrng_blocks) * 256 + tid;
  const int M = a.M;
  if (e >= (int64_t)a.C * M * M) return;
  const int j = e % M, i = (e / M) % M;
  const int64_t c = e / ((int64_t)M * M);
  const float* v = a.vec + c * ((int64_t)M * (M + 1) / 2);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const float* rh = v + (int64_t)hi * (hi + 1) / 2;
  const float* rl = v + (int64_t)lo * (lo + 1) / 2;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;   // independent chains: the loads of a group are in flight together
  int k = a.su_in_chain ? lo : 0;                          // (su_in_chain: no dot product, no S_u from here)
  for (; k + 4 <= lo; k += 4) {
    acc0 = fmaf(rh[k], rl[k], acc0); acc1 = fmaf(rh[k + 1], rl[k + 1], acc1);
    acc2 = fmaf(rh[k + 2], rl[k + 2], acc2); acc3 = fmaf(rh[k + 3], rl[k + 3], acc3);
  }
  for (; k < lo; ++k) acc0 = fmaf(rh[k], rl[k], acc0);
  const float dl = softplus_t0(rl[lo]);
  const float acc = fmaf(hi == lo ? dl : rh[lo], dl, (acc0 + acc1) + (acc2 + acc3));
  if (!a.su_in_chain) a.Su[e] = acc;
  const float lu = j < i ? v[(int64_t)i * (i + 1) / 2 + j] : (j == i ? dl : 0.f);
  a.Lu[e] = lu;
  if (a.RK) {   // RK[s, c, i, :] = [ m | 0 0 0 | (L_S: by the factorisation) | Lu | 0.. ]
    for (int s = 0; s < a.S; ++s) {
      float* r = a.RK + (((int64_t)s * a.C + c) * M + i) * a.LD;
      r[4 + M + j] = lu;
      if (j == 0) {
        r[0] = a.u_mean[c * M + i];
        r[1] = 0.f; r[2] = 0.f; r[3] = 0.f;
        for (int col = 4 + 2 * M; col < a.NR; ++col) r[col] = 0.f;
      }
    }
  }
}

// sum_d x_d^2 w_d of one row by one wave.  Four 64-wide chunks of loads are issued before the first use (a plain loop
// keeps one chunk in flight: D / 64 memory round trips in a row), on clamped indices with the overhang masked.
__device__ __forceinline__ float row_norm_wave(const float* __restrict__ xr, const float* __restrict__ ws, int D, int lane) {
  float acc0 = 0.f, acc1 = 0.f;
  for (int d0 = 0; d0 < D; d0 += 256) {
    float xv[4], wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int d = min(d0 + 64 * q + lane, D - 1);
      xv[q] = xr[d]; wv[q] = ws[d];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = (d0 + 64 * q + lane < D) ? xv[q] : 0.f;
      if (q & 1) acc1 = fmaf(v * v, wv[q], acc1); else acc0 = fmaf(v * v, wv[q], acc0);
    }
  }
  return wave_sum(acc0 + acc1);
}

// Row-norm role next to a product that is in the same launch as the hyper-parameter draw: the block evaluates 1/sigma_s^2
// itself (into `wl`, >= D floats of LDS), then `rows_per_block` rows of [z; x], one wave per row at a time.
struct NormArgs {
  const float *z, *x;
  float *na, *nb;
  int64_t zrows, xrows;
  int rows_per_block, nrow_blocks;       // blocks per hyper-sample
  float* xs;                             // (S, xrows, D) or NULL: x o 1/sigma_s^2, the pre-scaled operand of the K_uf distance product
};
__device__ __forceinline__ void t0_norm_body(const ProArgs& a, const NormArgs& n, const int id, float* __restrict__ wl) {
  const int s = id / n.nrow_blocks, rb = id - s * n.nrow_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n.rows_per_block <= 16 && (a.D & 3) == 0 && a.D <= 1024 &&
      ((reinterpret_cast<uintptr_t>(n.z) | reinterpret_cast<uintptr_t>(n.x) | reinterpret_cast<uintptr_t>(n.xs)) & 15) == 0) {
    // Short rows (the MNIST shapes): the wave's four rows are requested whole, as float4, BEFORE the theta phase -- one memory
    // round trip beside that phase's own instead of one per 256-column chunk behind it (the role was the long pole of its launch:
    // 19 us per workgroup against 16.6 for the K-split Gram tiles beside it).
    const int64_t nrows = n.zrows + n.xrows, last = nrows - 1;
    const int D4 = a.D >> 2;
    const float4* xr[4];
    float4* xo[4];
    int64_t row[4];
    bool live[4], xout[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      row[q] = (int64_t)rb * n.rows_per_block + wave + 4 * q;
      live[q] = wave + 4 * q < n.rows_per_block && row[q] < nrows;
      const int64_t rc = row[q] < last ? row[q] : last;
      xr[q] = reinterpret_cast<const float4*>(rc < n.zrows ? n.z + rc * a.D : n.x + (rc - n.zrows) * a.D);
      xout[q] = n.xs != nullptr && live[q] && row[q] >= n.zrows;
      xo[q] = reinterpret_cast<float4*>(n.xs + ((int64_t)s * n.xrows + (xout[q] ? row[q] - n.zrows : 0)) * a.D);
    }
    float4 xv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[q][c] = xr[q][min(64 * c + lane, D4 - 1)];
    for (int d0 = tid; d0 < a.D; d0 += 1024) {
      float tb[4];
      t0_theta_batch<4>(a, s, d0, false, tb);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (d0 + 256 * u < a.D) wl[d0 + 256 * u] = expf(-2.f * tb[u]);
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i4 = 64 * c + lane;
      const bool ok = i4 < D4;
      const float4 w4 = ok ? reinterpret_cast<const float4*>(wl)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = xv[q][c];
        const float4 sv = make_float4(v.x * w4.x, v.y * w4.y, v.z * w4.z, v.w * w4.w);
        acc[q] = fmaf(v.x, sv.x, fmaf(v.y, sv.y, fmaf(v.z, sv.z, fmaf(v.w, sv.w, acc[q]))));
        if (xout[q] && ok) xo[q][i4] = sv;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float t = wave_sum(acc[q]);
      if (lane == 0 && live[q]) {
        if (row[q] < n.zrows) n.na[(int64_t)s * n.zrows + row[q]] = t; else n.nb[(int64_t)s * n.xrows + (row[q] - n.zrows)] = t;
      }
    }
    return;
  }
  for (int d0 = tid; d0 < a.D; d0 += 1024) {
    float tb[4];
    t0_theta_batch<4>(a, s, d0, false, tb);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (d0 + 256 * u < a.D) wl[d0 + 256 * u] = expf(-2.f * tb[u]);
  }
  __syncthreads();
  // the wave's rows (wave, wave + 4, ...) four at a time, all their loads in flight together
  const int64_t nrows = n.zrows + n.xrows, last = nrows - 1;
  for (int r0 = wave; r0 < n.rows_per_block; r0 += 16) {
    const float* xr[4];
    int64_t row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      row[q] = (int64_t)rb * n.rows_per_block + r0 + 4 * q;
      const int64_t rc = row[q] < last ? row[q] : last;
      xr[q] = rc < n.zrows ? n.z + rc * a.D : n.x + (rc - n.zrows) * a.D;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    bool xout[4];                                     // (uniform) the row is a row of x whose scaled copy this wave writes
    float* xo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xout[q] = n.xs != nullptr && row[q] >= n.zrows && row[q] < nrows && r0 + 4 * q < n.rows_per_block;
      xo[q] = n.xs + ((int64_t)s * n.xrows + (xout[q] ? row[q] - n.zrows : 0)) * a.D;
    }
    // (the next chunk's loads are issued BEFORE this chunk's stores: a load behind a store waits for the store's
    // acknowledgement as well -- vmcnt counts both in order)
    float xv[4][4], xn[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[q][c] = xr[q][min(64 * c + lane, a.D - 1)];
    for (int d0 = 0; d0 < a.D; d0 += 256) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) xn[q][c] = xr[q][min(d0 + 256 + 64 * c + lane, a.D - 1)];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = d0 + 64 * c + lane;
        const float wv = dd < a.D ? wl[dd] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q] = fmaf(xv[q][c] * xv[q][c], wv, acc[q]);
          if (xout[q] && dd < a.D) xo[q][dd] = xv[q][c] * wv;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[q][c] = xn[q][c];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float t = wave_sum(acc[q]);
      if (lane == 0 && r0 + 4 * q < n.rows_per_block && row[q] < nrows) {
        if (row[q] < n.zrows) n.na[(int64_t)s * n.zrows + row[q]] = t; else n.nb[(int64_t)s * n.xrows + (row[q] - n.zrows)] = t;
      }
    }
  }
}

// prologue roles + row norms + the K-split K_uu inner products (ps: the split product, partials to ps.C + split * ps.sSplit)
// in ONE launch (gemm.hip); npro = number of prologue blocks
int launch_pro_kuu(const ProArgs& a, int npro, const NormArgs& n, const GemmParams& ps, int nbatch, hipStream_t st);

}  // namespace vargp
