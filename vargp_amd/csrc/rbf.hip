// RBF / ARD kernel-matrix construction and its backward (reference: RBFKernel.compute,
// var_gp/kernels.py:24-56).  The inner products run on the f32 MFMA through gemm.hip; this file has
// the pre-pass (1/sigma^2, gamma^2, weighted squared row norms), the backward pre-pass
// (W = gK o K and its row/column sums) and the backward finalisation (gX, gY, gtheta from W.Y).
//
//   d2_ij  = sum_d w_d (x_id - y_jd)^2 = na_i + nb_j - 2 sum_d w_d x_id y_jd,   w = exp(-2 theta_d)
//   K_ij   = g2 exp(-d2_ij / 2),  g2 = exp(2 theta_D)
//   with W = gK o K, r = rowsum W, c = colsum W, P = W Y, Q = W^T X:
//   gX_i   = -sum_s w_s o (r_i x_i - P_i)          gY_j = -sum_s w_s o (c_j y_j - Q_j)
//   gth_sd = w_sd [ sum_i x_id (r_i x_id - 2 P_id) + sum_j c_j y_jd^2 ]      gth_sD = 2 sum W
#include "common.h"

namespace vargp {

struct RbfWs {
  float *w, *g2, *na, *nb, *part, *ys, *Wm, *r, *c, *P, *Q;
  int64_t Dp;
  size_t bytes;
};

static RbfWs carve(void* ws, int S, int C, int M, int N, int D, bool backward) {
  RbfWs o{};
  o.Dp = round_up(D, 4);
  float* p = reinterpret_cast<float*>(ws);
  auto take = [&](int64_t n) { float* q = p; p += round_up(n, 64); return q; };
  o.w = take((int64_t)S * o.Dp);
  o.g2 = take(S);
  if (!backward) {
    o.na = take((int64_t)S * C * M);
    o.nb = take((int64_t)S * C * N);
    o.part = take((int64_t)2 * S * C * M * N);      // split-K partial products (at most 2 splits)
    o.ys = take((int64_t)S * N * D);                // y o w of a shared y (one copy per hyper-sample)
  } else {
    o.Wm = take((int64_t)S * C * M * N);
    o.r = take((int64_t)S * C * M);
    o.c = take((int64_t)S * C * N);
    o.P = take((int64_t)S * C * M * D);
    o.Q = take((int64_t)S * C * N * D);
  }
  o.bytes = (size_t)((char*)p - (char*)ws);
  return o;
}

__global__ void rbf_prep_kernel(const float* __restrict__ theta, float* __restrict__ w, float* __restrict__ g2,
                                int D, int64_t Dp) {
  const int s = blockIdx.x;
  const float* th = theta + (int64_t)s * (D + 1);
  for (int d = threadIdx.x; d < Dp; d += blockDim.x) w[s * Dp + d] = d < D ? expf(-2.f * th[d]) : 0.f;
  if (threadIdx.x == 0) g2[s] = expf(2.f * th[D]);
}

// Small input dimension (D <= kDirectD, e.g. the 2-D toy problem): form the squared distance directly as
// sum_d w_d (x_d - y_d)^2 — no cancellation, no GEMM.  One thread per kernel-matrix entry.
constexpr int kDirectD = kRbfDirectD;
__global__ __launch_bounds__(256) void rbf_direct_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         const float* __restrict__ w, const float* __restrict__ g2,
                                                         float* __restrict__ K, int64_t ldk, int C, int M, int N,
                                                         int D, int64_t Dp, int y_shared, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int n = e % N, m = (e / N) % M, c = (e / ((int64_t)N * M)) % C;
  const int s = e / ((int64_t)N * M * C);
  const float* xr = X + ((int64_t)c * M + m) * D;
  const float* yr = Y ? (y_shared ? Y + (int64_t)n * D : Y + ((int64_t)c * N + n) * D) : X + ((int64_t)c * M + n) * D;
  const float* ws = w + s * Dp;
  float d2 = 0.f;
  for (int d = 0; d < D; ++d) { const float t = xr[d] - yr[d]; d2 = fmaf(ws[d] * t, t, d2); }
  K[(((int64_t)s * C + c) * M + m) * ldk + n] = g2[s] * expf(-0.5f * d2);
}

// K = g2 exp(-0.5 (na + nb - 2 (ab_0 + ab_1 ...))) from split-K partial inner products; same arithmetic as the fused
// GEMM epilogue.  One thread per entry of the flattened [nb0][rows][N] result.
__global__ __launch_bounds__(256) void rbf_combine_kernel(const float* __restrict__ part, int nsplit, int64_t sSplit,
                                                          const float* __restrict__ na, const float* __restrict__ nbv,
                                                          const float* __restrict__ g2, float* __restrict__ K,
                                                          int64_t rows_per_s, int N, int64_t nb_stride_s,
                                                          int64_t nb_stride_c, int Mb, int same_xy, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int col = e % N;
  const int64_t row = e / N;                 // s * rows_per_s + (c * Mb + m)
  const int64_t s = row / rows_per_s, rc = row % rows_per_s;
  const int64_t c = rc / Mb;
  const int m = rc % Mb;
  float ab = 0.f;
  for (int k = 0; k < nsplit; ++k) ab += part[k * sSplit + e];
  const float d2 = na[row] + nbv[s * nb_stride_s + c * nb_stride_c + col] - 2.f * ab;
  K[e] = (same_xy && m == col) ? g2[s] : g2[s] * expf(-0.5f * d2);
}

// prep + both norm passes in one launch: nrm_x[s][row] = sum_d w_sd x[row][d]^2 (likewise y), w_sd = exp(-2 theta_sd)
// evaluated on the fly; the blocks with blockIdx.x == 0 also store w (zero-padded to Dp) and g2 = exp(2 theta_sD) for the
// GEMM that follows.  One wave per row; grid (ceil((xrows + yrows) / 4), S).
// ys / xs (nullable): the scaled copies y o w, [S][yrows][D], and x o w, [S][xrows][D] -- with a pre-scaled operand the
// distance GEMM needs no per-k scaling (GemmParams.kscale = NULL).
__global__ __launch_bounds__(256) void rbf_prep_norm_kernel(const float* __restrict__ theta, const float* __restrict__ x,
                                                            const float* __restrict__ y, float* __restrict__ w,
                                                            float* __restrict__ g2, float* __restrict__ na,
                                                            float* __restrict__ nb, int64_t xrows, int64_t yrows, int D,
                                                            int64_t Dp, float* __restrict__ ys, float* __restrict__ xs) {
  const int s = blockIdx.y, lane = threadIdx.x & 63;
  const float* th = theta + (int64_t)s * (D + 1);
  if (blockIdx.x == 0) {
    for (int d = threadIdx.x; d < Dp; d += 256) w[s * Dp + d] = d < D ? expf(-2.f * th[d]) : 0.f;
    if (threadIdx.x == 0) g2[s] = expf(2.f * th[D]);
  }
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= xrows + yrows) return;
  const bool isx = row < xrows;
  const float* xr = isx ? x + row * D : y + (row - xrows) * D;
  float* yo = isx ? (xs ? xs + ((int64_t)s * xrows + row) * D : nullptr)
                  : (ys ? ys + ((int64_t)s * yrows + (row - xrows)) * D : nullptr);
  // sixteen 64-wide chunks per pass, ALL their loads first (clamped index), then the arithmetic and the stores: D = 784 is one
  // pass -- one memory round trip for the row instead of D / 256 with a store's acknowledgement in front of every next load
  // (vmcnt retires in order)
  float acc0 = 0.f, acc1 = 0.f;
  for (int d0 = 0; d0 < D; d0 += 1024) {
    float xv[16], tv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int d = min(d0 + 64 * q + lane, D - 1);
      xv[q] = xr[d]; tv[q] = th[d];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int d = d0 + 64 * q + lane;
      const float wv = expf(-2.f * tv[q]);
      const float v = d < D ? xv[q] : 0.f;
      if (q & 1) acc1 = fmaf(v * v, wv, acc1); else acc0 = fmaf(v * v, wv, acc0);
      if (yo && d < D) yo[d] = v * wv;
    }
  }
  const float acc = wave_sum(acc0 + acc1);
  if (lane == 0) {
    if (isx) na[(int64_t)s * xrows + row] = acc; else nb[(int64_t)s * yrows + (row - xrows)] = acc;
  }
}

// W = gK o K with its row sums r, column sums c and total (dlog gamma = 2 sum W) in one pass.
// grid (ceil(N/256), ceil(Mb/32), nb): a thread owns one column of a 32-row strip; r and c are
// accumulated with float atomics (pre-zeroed by the caller), the total with one atomic per block.
constexpr int WROWS = 8;
__global__ __launch_bounds__(256) void rbf_w_kernel(const float* __restrict__ K, const float* __restrict__ gK,
                                                    float* __restrict__ W, float* __restrict__ r,
                                                    float* __restrict__ c, float* __restrict__ gtheta, int Mb, int N,
                                                    int Cb, int D) {
  __shared__ float red[4];
  const int col = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int row0 = blockIdx.y * WROWS;
  const int64_t b = blockIdx.z;
  const bool cok = col < N;
  const int64_t base = b * Mb * N;
  float csum = 0.f;
  const int rend = min(WROWS, Mb - row0);
  for (int rr = 0; rr < rend; ++rr) {
    const int64_t off = base + (int64_t)(row0 + rr) * N + col;
    float v = 0.f;
    if (cok) { v = K[off] * gK[off]; W[off] = v; }
    csum += v;
    const float rs = wave_sum(v);
    if (lane == 0 && rs != 0.f) atomicAdd(&r[b * Mb + row0 + rr], rs);
  }
  if (cok) atomicAdd(&c[b * N + col], csum);
  const float tot = block_sum<256>(csum, red);
  if (threadIdx.x == 0) atomicAdd(&gtheta[(b / Cb) * (D + 1) + D], 2.f * tot);
}

// Square case (Y = X) in one pass: Ws = W + W^T with W = gK o K, r = its row sums (= row + column sums of W), and
// sum Ws (= 2 sum W) into gtheta[s, D].  kSelfRows consecutive rows of one matrix per block, a wave takes every 4th.
constexpr int kSelfRows = 16;
__global__ __launch_bounds__(256) void rbf_w_self_kernel(const float* __restrict__ K, const float* __restrict__ gK,
                                                         float* __restrict__ Ws, float* __restrict__ r,
                                                         float* __restrict__ gtheta, int M, int Cb, int D, int nchunk,
                                                         int sym) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const int64_t b = blockIdx.x / nchunk;
  const int i0 = ((int)blockIdx.x % nchunk) * kSelfRows, i1 = min(M, i0 + kSelfRows);
  const float* Kb = K + b * M * M;
  const float* gKb = gK + b * M * M;
  float tot = 0.f, dsum = 0.f;
  for (int i = i0 + (threadIdx.x >> 6); i < i1; i += 4) {
    float acc = 0.f;
    for (int j = lane; j < M; j += 64) {
      // sym: gK is symmetric (K always is), so W = gK o K is too and W + W^T = 2 W: no transposed (uncoalesced) reads
      const float v = sym ? 2.f * Kb[(int64_t)i * M + j] * gKb[(int64_t)i * M + j]
                          : Kb[(int64_t)i * M + j] * gKb[(int64_t)i * M + j] + Kb[(int64_t)j * M + i] * gKb[(int64_t)j * M + i];
      // The diagonal: K_ii = gamma^2 does not depend on x_i or the lengthscales (the reference's autograd cancels it
      // exactly: -2 g + g + g on the entry (i, i) of its Gram, kernels.py:44-54), so it must not reach P = Ws x and r, whose
      // difference would otherwise leave rounding noise of order eps W_ii x_i where the reference has an exact zero (with
      // underflowing off-diagonals -- MNIST pixels at the initial lengthscale -- the whole gradient).  It only counts for gamma.
      const bool dg = i == j;
      Ws[b * M * M + (int64_t)i * M + j] = dg ? 0.f : v;
      acc += dg ? 0.f : v;
      dsum += dg ? v : 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0) r[b * M + i] = acc;
    tot += acc;
  }
  tot += wave_sum(dsum);
  const float t = block_sum<256>(lane == 0 ? tot : 0.f, red);   // every lane of a wave holds the wave's total
  if (threadIdx.x == 0) atomicAdd(&gtheta[(b / Cb) * (D + 1) + D], t);
}

// Finalise one side.  rows = points of this side (flattened over classes), S samples.
//   g[row][d]  (+)= -sum_s w_sd (R_s,row x_row,d - P_s,row,d)            (if g != null)
//   gtheta[s][d] += w_sd sum_row x (R x - kappa P)                         (P may be null -> 0)
// block = 64 d-columns x 4 row lanes, RPB rows per block.
constexpr int RPB = 32;
__global__ __launch_bounds__(256) void rbf_final_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                        const float* __restrict__ P, const float* __restrict__ theta,
                                                        float* __restrict__ g, float* __restrict__ gtheta,
                                                        int64_t rows, int D, int64_t Dp, int S, float kappa,
                                                        int accumulate) {
  __shared__ float red[4][64];
  const int dx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dx;
  const int64_t row0 = (int64_t)blockIdx.y * RPB;
  const bool dok = d < D;
  float xa[RPB / 4], ga[RPB / 4];
#pragma unroll
  for (int j = 0; j < RPB / 4; ++j) {
    const int64_t row = row0 + ry + 4 * j;
    xa[j] = (dok && row < rows) ? x[row * D + d] : 0.f;
    ga[j] = 0.f;
  }
  for (int s = 0; s < S; ++s) {
    const float wv = dok ? expf(-2.f * theta[(int64_t)s * (D + 1) + d]) : 0.f;   // 1/sigma_d^2
    float th = 0.f;
#pragma unroll
    for (int j = 0; j < RPB / 4; ++j) {
      const int64_t row = row0 + ry + 4 * j;
      if (row < rows) {
        const float rr = R[(int64_t)s * rows + row];
        const float pv = (P && dok) ? P[((int64_t)s * rows + row) * D + d] : 0.f;
        const float rx = rr * xa[j];
        ga[j] -= wv * (rx - pv);
        th += xa[j] * (rx - kappa * pv);
      }
    }
    __syncthreads();
    red[ry][dx] = th;
    __syncthreads();
    if (ry == 0 && dok) {
      const float t = red[0][dx] + red[1][dx] + red[2][dx] + red[3][dx];
      atomicAdd(&gtheta[(int64_t)s * (D + 1) + d], wv * t);
    }
  }
  if (g && dok) {
#pragma unroll
    for (int j = 0; j < RPB / 4; ++j) {
      const int64_t row = row0 + ry + 4 * j;
      if (row < rows) {
        if (accumulate) g[row * D + d] += ga[j]; else g[row * D + d] = ga[j];
      }
    }
  }
}

int rbf_prep_norm_launch(const float* theta, const float* x, int64_t xrows, const float* y, int64_t yrows, float* w,
                         float* g2, float* na, float* nb, int S, int D, int64_t Dp, hipStream_t st, float* ys, float* xs) {
  hipLaunchKernelGGL(rbf_prep_norm_kernel, dim3(cdiv(xrows + yrows, 4), S), dim3(256), 0, st, theta, x, y, w, g2, na, nb,
                     xrows, yrows, D, Dp, ys, xs);
  return check_launch("rbf_prep_norm");
}

int rbf_direct_launch(const float* X, const float* Y, const float* w, const float* g2, float* K, int64_t ldk, int S,
                      int C, int M, int N, int D, int64_t Dp, int y_shared, hipStream_t st) {
  const int64_t total = (int64_t)S * C * M * N;
  hipLaunchKernelGGL(rbf_direct_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, X, Y, w, g2, K, ldk, C, M, N, D, Dp,
                     y_shared, total);
  return check_launch("rbf_gram_fwd(direct)");
}

}  // namespace vargp

using namespace vargp;

extern "C" size_t vargp_rbf_workspace_bytes(int S, int C, int M, int N, int D, int backward) {
  return carve(nullptr, S, C, M, N, D, backward != 0).bytes + 256;
}

extern "C" int vargp_rbf_gram_fwd(const float* theta, const float* X, const float* Y, float* K, int S, int C, int M,
                                  int N, int D, int y_shared, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  return vargp::rbf_gram_fwd_impl(theta, X, Y, K, S, C, M, N, D, y_shared, ws, ws_bytes, 0, as_stream(stream));
}

// sym_out (Y = NULL only): compute the tiles that touch the lower triangle and mirror them (K is symmetric)
int vargp::rbf_gram_fwd_impl(const float* theta, const float* X, const float* Y, float* K, int S, int C, int M, int N, int D,
                             int y_shared, void* ws, size_t ws_bytes, int sym_out, hipStream_t st) {
  VARGP_REQUIRE(theta && X && K && ws, "rbf_gram_fwd: null pointer");
  VARGP_REQUIRE(S > 0 && C > 0 && M > 0 && D > 0, "rbf_gram_fwd: bad dims");
  const bool self = (Y == nullptr);
  if (self) { N = M; y_shared = 0; }
  VARGP_REQUIRE(N > 0, "rbf_gram_fwd: bad N");
  VARGP_REQUIRE(ws_bytes >= vargp_rbf_workspace_bytes(S, C, M, N, D, 0), "rbf_gram_fwd: workspace too small");
  RbfWs o = carve(ws, S, C, M, N, D, false);
  if (D <= kDirectD) {
    hipLaunchKernelGGL(rbf_prep_kernel, dim3(S), dim3(256), 0, st, theta, o.w, o.g2, D, o.Dp);
    return rbf_direct_launch(X, Y, o.w, o.g2, K, N, S, C, M, N, D, o.Dp, y_shared, st);
  }
  const int64_t xrows = (int64_t)C * M, yrows = y_shared ? N : (int64_t)C * N;
  // shared y (the minibatch): pre-scaled once per hyper-sample by the norm pass, so that the GEMM's main loop carries no
  // scale loads / multiplies
  const bool prescale = y_shared && !self;
  hipLaunchKernelGGL(rbf_prep_norm_kernel, dim3(cdiv(xrows + (self ? 0 : yrows), 4), S), dim3(256), 0, st, theta, X, Y, o.w,
                     o.g2, o.na, o.nb, xrows, self ? (int64_t)0 : yrows, D, o.Dp, prescale ? o.ys : (float*)nullptr, (float*)nullptr);
  // shared Y: the classes' inducing points are just more rows of one [C*M, D] x [D, N] product
  const int Cb = y_shared ? 1 : C, Mb = y_shared ? C * M : M;
  GemmParams p{};
  p.A = X; p.B = self ? X : Y; p.C = K; p.D = nullptr;
  p.M = Mb; p.N = N; p.K = D; p.lda = D; p.ldb = D; p.ldc = N; p.ldd = 0;
  p.nb1 = Cb; p.nb2 = 1;
  p.sA[0] = 0; p.sA[1] = (int64_t)Mb * D;
  p.sB[0] = 0; p.sB[1] = y_shared ? 0 : (int64_t)N * D;
  p.sC[0] = (int64_t)Cb * Mb * N; p.sC[1] = (int64_t)Mb * N;
  p.alpha = 1.f; p.beta = 0.f;
  p.kscale = o.w; p.ks_ld = o.Dp; p.g2 = o.g2;
  p.na = o.na; p.sNa[0] = xrows; p.sNa[1] = Mb;
  p.nbv = self ? o.na : o.nb; p.sNb[0] = self ? xrows : yrows; p.sNb[1] = (self || !y_shared) ? N : 0;
  p.same_xy = self ? 1 : 0;
  if (prescale) { p.B = o.ys; p.sB[0] = (int64_t)N * D; p.kscale = nullptr; }
  const int nsplit = (self && sym_out) ? 1 : rbf_splitk(Mb, N, D, S * Cb);
  if (self && sym_out) { p.triC = 2; p.symout = 1; }
  int rc;
  {
    ProfScope whole(self ? "rbf_kuu" : "rbf_kuf", st);    // distance GEMM (+ combine pass if K was split)
    if (nsplit > 1) {
      p.splitk = nsplit;
      p.sSplit = (int64_t)S * Cb * Mb * N;
      p.C = o.part;
      rc = launch_gemm(p, 0, 1, S * Cb, true, st, self ? "rbf_kuu_gemm" : "rbf_kuf_gemm");
      if (rc) return rc;
      const int64_t total = (int64_t)S * Cb * Mb * N;
      hipLaunchKernelGGL(rbf_combine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, o.part, nsplit, p.sSplit, o.na,
                         self ? o.na : o.nb, o.g2, K, (int64_t)Cb * Mb, N, self ? xrows : yrows,
                         (self || !y_shared) ? (int64_t)N : 0, Mb, self ? 1 : 0, total);
    } else {
      rc = launch_gemm(p, 0, 1, S * Cb, true, st, self ? "rbf_kuu_gemm" : "rbf_kuf_gemm");
      if (rc) return rc;
    }
  }
  return check_launch("rbf_gram_fwd");
}

extern "C" int vargp_rbf_gram_bwd(const float* theta, const float* X, const float* Y, const float* K, const float* gK,
                                  float* gX, float* gY, float* gtheta, int S, int C, int M, int N, int D,
                                  int y_shared, int accumulate, void* ws, size_t ws_bytes, vargp_stream_t stream) {
  return vargp::rbf_gram_bwd_impl(theta, X, Y, K, gK, gX, gY, gtheta, S, C, M, N, D, y_shared, accumulate, ws, ws_bytes, 0,
                                  as_stream(stream));
}

// sym_gk (Y = NULL only): the caller guarantees a symmetric gK (e.g. the output of the Cholesky backward)
int vargp::rbf_gram_bwd_impl(const float* theta, const float* X, const float* Y, const float* K, const float* gK, float* gX,
                             float* gY, float* gtheta, int S, int C, int M, int N, int D, int y_shared, int accumulate,
                             void* ws, size_t ws_bytes, int sym_gk, hipStream_t st) {
  VARGP_REQUIRE(theta && X && K && gK && gtheta && ws, "rbf_gram_bwd: null pointer");
  const bool self = (Y == nullptr);
  if (self) { N = M; y_shared = 0; gY = nullptr; }
  VARGP_REQUIRE(ws_bytes >= vargp_rbf_workspace_bytes(S, C, M, N, D, 1), "rbf_gram_bwd: workspace too small");
  RbfWs o = carve(ws, S, C, M, N, D, true);
  const int Cb = y_shared ? 1 : C, Mb = y_shared ? C * M : M;
  const int64_t xrows = (int64_t)C * M, yrows = y_shared ? N : (int64_t)C * N;
  const int nb = S * Cb;

  if (!accumulate) zero_async(gtheta, sizeof(float) * (size_t)S * (D + 1), st);
  if (self) {
    const int nchunk = cdiv(M, kSelfRows);
    hipLaunchKernelGGL(rbf_w_self_kernel, dim3(nb * nchunk), dim3(256), 0, st, K, gK, o.Wm, o.r, gtheta, M, Cb, D, nchunk,
                       sym_gk);
  } else {
    zero_async(o.r, sizeof(float) * (size_t)(o.P - o.r), st);   // r and c are adjacent
    hipLaunchKernelGGL(rbf_w_kernel, dim3(cdiv(N, 256), cdiv(Mb, WROWS), nb), dim3(256), 0, st, K, gK, o.Wm, o.r, o.c,
                       gtheta, Mb, N, Cb, D);
  }
  // P = W . Y   ([Mb, N] x [N, D]) per (s, class-batch)
  GemmParams p{};
  p.A = o.Wm; p.B = self ? X : Y; p.C = o.P;
  p.M = Mb; p.N = D; p.K = N; p.lda = N; p.ldb = D; p.ldc = D;
  p.nb1 = Cb; p.nb2 = 1;
  p.sA[0] = (int64_t)Cb * Mb * N; p.sA[1] = (int64_t)Mb * N;
  p.sB[0] = 0; p.sB[1] = y_shared ? 0 : (int64_t)N * D;
  p.sC[0] = (int64_t)Cb * Mb * D; p.sC[1] = (int64_t)Mb * D;
  p.alpha = 1.f;
  int rc = launch_gemm(p, 0, 0, nb, false, st, self ? "rbf_kuu_bwd_gemm" : "rbf_kuf_bwd_gemm");
  if (rc) return rc;
  const dim3 gx(cdiv(D, 64), cdiv(xrows, RPB));
  hipLaunchKernelGGL(rbf_final_kernel, gx, dim3(256), 0, st, X, o.r, o.P, theta, gX, gtheta, xrows, D, o.Dp, S,
                     self ? 1.f : 2.f, accumulate);
  if (!self) {
    const float* Qp = nullptr;
    if (gY) {  // Q = W^T . X  ([N, Mb] x [Mb, D])
      GemmParams q{};
      q.A = o.Wm; q.B = X; q.C = o.Q;
      q.M = N; q.N = D; q.K = Mb; q.lda = N; q.ldb = D; q.ldc = D;
      q.nb1 = Cb; q.nb2 = 1;
      q.sA[0] = (int64_t)Cb * Mb * N; q.sA[1] = (int64_t)Mb * N;
      q.sB[0] = 0; q.sB[1] = (int64_t)Mb * D;
      q.sC[0] = (int64_t)Cb * N * D; q.sC[1] = (int64_t)N * D;
      q.alpha = 1.f;
      rc = launch_gemm(q, 1, 0, nb, false, st);
      if (rc) return rc;
      Qp = o.Q;
    }
    const dim3 gy(cdiv(D, 64), cdiv(yrows, RPB));
    hipLaunchKernelGGL(rbf_final_kernel, gy, dim3(256), 0, st, Y, o.c, Qp, theta, gY, gtheta, yrows, D, o.Dp, S, 0.f,
                       accumulate);
  }
  return check_launch("rbf_gram_bwd");
}
