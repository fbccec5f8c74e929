/*
 * libvargp_hip — C ABI of the MI355X (gfx950) kernels behind the VAR-GP ELBO hot path.
 *
 * The reference (uber-research/vargp) has no FFI: its hot path is Python on top of ATen call sites
 * (SURVEY.md §2.3).  Each entry point below replaces one group of those call sites; the reference
 * lines are cited per function.  Conventions:
 *   - all tensors are contiguous row-major fp32 DEVICE pointers unless a leading dimension /
 *     stride argument says otherwise; labels are int64; the caller owns every buffer, including
 *     the workspace (size from the matching *_workspace_bytes query);
 *   - every function only enqueues work on `stream` and never synchronises or allocates, so the
 *     caller may capture a sequence of calls into a hipGraph;
 *   - return value: 0 on success, negative VARGP_E* on a bad argument / launch failure
 *     (vargp_last_error() gives the text).  Numerical failure of a Cholesky is reported through the
 *     device-side `info` array (0 = ok, j+1 = first non-positive pivot, LAPACK style); the factor
 *     of a failed matrix is filled with NaN so the failure cannot go unnoticed downstream.
 */
#ifndef VARGP_HIP_H
#define VARGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vargp_stream_t; /* a hipStream_t */

#define VARGP_OK 0
#define VARGP_EINVAL (-1)
#define VARGP_ELAUNCH (-2)
#define VARGP_EWORKSPACE (-3)

int vargp_version(void);
const char* vargp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Batched strided fp32 GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32):
 *     C[b] = alpha * op(A[b]) * op(B[b]) + beta * D[b]          op(A): M x K, op(B): K x N
 * Replaces the einsum/bmm call sites of var_gp/gp_utils.py:18,94,96,127,131,136,178,184 and the
 * triangular_solve call sites :89,92,124-134,175-182 (solves are products with the explicit
 * inverse factor T = L^-1 produced by vargp_chol_inv_fwd).
 * transX = 0: X stored row-major as op(X) (ldx = row stride); transX = 1: stored as op(X)^T.
 * Batch index b = (i0*nb[1] + i1)*nb[2] + i2, element strides per batch dim (0 = broadcast).
 * triA/triB: structure hint for op(A)/op(B) (0 none, 1 lower-triangular, 2 upper-triangular); the
 * stored zeros must really be zero, the hint only clips the K range.
 * triC: 0 full; 1 compute the lower triangle and write zeros above it; 2 compute and write only
 * tiles that touch the lower triangle (entries above the diagonal are left unspecified / untouched).
 * D may be NULL (beta ignored) and may alias C.
 */
typedef struct vargp_gemm_desc {
  int32_t M, N, K;
  int32_t transA, transB;
  const float* A;
  const float* B;
  float* C;
  const float* D;
  int32_t lda, ldb, ldc, ldd;
  int32_t nb[3];
  int64_t sA[3], sB[3], sC[3], sD[3];
  float alpha, beta;
  int32_t triA, triB, triC;
} vargp_gemm_desc;

int vargp_bgemm(const vargp_gemm_desc* d, vargp_stream_t stream);

/* out[i] = sum_r in[r*inner + i], r < outer   (reduction of broadcast batch dims in backward) */
int vargp_sum_outer(const float* in, float* out, int64_t outer, int64_t inner, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RBF / ARD kernel matrix (reference: RBFKernel.compute, var_gp/kernels.py:24-56).
 *   theta[S, D+1] = [log lengthscale_1..D, log gamma];  X[C, M, D];
 *   Y: NULL (Y = X, symmetric K_uu: the diagonal is exactly gamma^2, kernels.py:47-48,54),
 *      or [C, N, D] (y_shared = 0), or [N, D] shared by every class (y_shared = 1; the reference
 *      expands the minibatch over classes, var_gp/vargp.py:106);
 *   K[S, C, M, N] = gamma_s^2 * exp(-0.5 * (|x/sig|^2 + |y/sig|^2 - 2 (x/sig).(y/sig))), no clamp.
 * Backward (autograd of the same lines): gK[S,C,M,N] -> gX[C,M,D], gY (NULL if not wanted; for
 * Y = NULL both sides are accumulated into gX), gtheta[S, D+1].
 */
size_t vargp_rbf_workspace_bytes(int S, int C, int M, int N, int D, int backward);
int vargp_rbf_gram_fwd(const float* theta, const float* X, const float* Y, float* K, int S, int C, int M,
                       int N, int D, int y_shared, void* ws, size_t ws_bytes, vargp_stream_t stream);
int vargp_rbf_gram_bwd(const float* theta, const float* X, const float* Y, const float* K, const float* gK,
                       float* gX, float* gY, float* gtheta, int S, int C, int M, int N, int D, int y_shared,
                       int accumulate /* add into gX / gY / gtheta instead of overwriting */, void* ws, size_t ws_bytes,
                       vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Batched Cholesky with jitter + explicit inverse factor (reference: gp_utils.cholesky,
 * var_gp/gp_utils.py:5-11, and every torch.triangular_solve(., Lz) that consumes it).
 *   A[nbatch, n, n] symmetric (lower triangle read);  L = chol(A + eps I) lower, zeros above;
 *   T = L^-1 (lower; NULL to skip);  logdet[nbatch] = sum_i log L_ii (NULL to skip);
 *   info[nbatch] as described at the top.
 * Backward: gA = d/dA of <gL, L> + <gT, T> (either may be NULL), symmetric like torch's
 * cholesky_backward.
 */
size_t vargp_chol_workspace_bytes(int nbatch, int n, int backward);
int vargp_chol_inv_fwd(const float* A, float eps, float* L, float* T, float* logdet, int32_t* info, int nbatch,
                       int n, void* ws, size_t ws_bytes, vargp_stream_t stream);
int vargp_chol_inv_bwd(const float* L, const float* T, const float* gL, const float* gT, float* gA, int nbatch,
                       int n, void* ws, size_t ws_bytes, vargp_stream_t stream);

/* Triangular solve against a factor L whose inverse T = L^-1 came with it from vargp_chol_inv_fwd (reference:
 * torch.triangular_solve(B, L, upper=False), var_gp/gp_utils.py:89,92,124-134,175-182).  By design a solve is a product
 * with T on the MFMA; these entries exist so that the op list of SURVEY.md §8(b) is complete.
 *   fwd: X[nb, n, nrhs] = T B.     bwd: gB = T^T gX,  gL = -tril(gB X^T)  (the adjoint w.r.t. L; either may be NULL;
 *   gB == NULL needs a workspace of nb*n*nrhs floats).
 */
int vargp_trsm_lower_fwd(const float* T, const float* B, float* X, int nbatch, int n, int nrhs, vargp_stream_t stream);
int vargp_trsm_lower_bwd(const float* T, const float* X, const float* gX, float* gB, float* gL, int nbatch, int n, int nrhs,
                         void* ws, size_t ws_bytes, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Packed lower triangle <-> matrix with softplus on the diagonal (reference: vec2tril /
 * mat2trilvec, var_gp/gp_utils.py:22-65; packing order = torch.tril_indices, row-major).
 */
int vargp_vec2tril_fwd(const float* vec, float* tril, int nbatch, int m, vargp_stream_t stream);
int vargp_vec2tril_bwd(const float* vec, const float* gtril, float* gvec, int nbatch, int m,
                       vargp_stream_t stream);
int vargp_mat2trilvec(const float* mat, float* vec, int nbatch, int m, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Predictive marginal mean / variance reductions (reference: linear_marginal_diag,
 * var_gp/gp_utils.py:178-186):  with P = Lz^-1 Kzx [nb, M, B], W = (Lz^-1 L_S)^T P [nb, M, B],
 * a = Lz^-1 m [nb, M]:   mu_b = sum_m P_mb a_m;  var_b = kdiag - sum_m P_mb^2 + sum_m W_mb^2,
 * kdiag[nb] (= gamma^2 of the sample, kernels.py:58-60).
 * a_stride: element stride between consecutive a_m (1 = contiguous [nb, M]; the fused path reads a as a
 * column of a wider matrix), a_bstride: stride between batches.  ga is always written contiguous [nb, M].
 */
int vargp_predictive_diag_fwd(const float* P, const float* W, const float* a, int64_t a_stride, int64_t a_bstride,
                              const float* kdiag, float* mu, float* var, int nbatch, int M, int B,
                              vargp_stream_t stream);
int vargp_predictive_diag_bwd(const float* P, const float* W, const float* a, int64_t a_stride, int64_t a_bstride,
                              const float* gmu, const float* gvar, float* gP, float* gW, float* ga, float* gkdiag,
                              int nbatch, int M, int B, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * KL(N(mu_q, Lq Lq^T) || N(mu_p, Lp Lp^T)) from its triangular ingredients (reference:
 * torch.distributions kl_divergence(MVN, MVN) as called from var_gp/vargp.py:182-190):
 *   kl = logdet_p - logdet_q + 0.5 * (|G|_F^2 + |d|^2 - M),  G = Lp^-1 Lq [nb, M, M],
 *   d = Lp^-1 (mu_q - mu_p) [nb, M];  logdet_* = sum log diag.
 * Forward reduces G and d; backward is gG = gkl * G, gd = gkl * d.
 */
int vargp_mvn_kl_fwd(const float* G, const float* d, const float* logdet_p, const float* logdet_q, float* kl,
                     int nbatch, int M, vargp_stream_t stream);
int vargp_mvn_kl_bwd(const float* G, const float* d, const float* gkl, float* gG, float* gd, int nbatch, int M,
                     vargp_stream_t stream);
/* logdet[b] = sum_i log L[b,i,i]; backward gL[b,i,i] = g[b] / L[b,i,i] (zeros elsewhere) */
int vargp_logdet_tril_fwd(const float* L, float* logdet, int nbatch, int n, vargp_stream_t stream);
int vargp_logdet_tril_bwd(const float* L, const float* g, float* gL, int nbatch, int n, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Monte-Carlo softmax likelihood (reference: MulticlassSoftmax, var_gp/likelihoods.py:13-63).
 *   mu, var [S, C, B]; eps [S, F, C, B]; y int64 [B].
 *   nll   = sum_b mean_{s,f} -log_softmax_c(mu + sqrt(var) eps)[y_b]           (likelihoods.py:33-47)
 *   probs[B, C] = mean_{s,f} softmax_c(...)                                       (likelihoods.py:49-63)
 * nll is accumulated with atomics into *nll, which the call zeroes first.
 */
int vargp_softmax_nll_fwd(const float* mu, const float* var, const float* eps, const int64_t* y, float* nll,
                          int S, int F, int C, int B, vargp_stream_t stream);
int vargp_softmax_nll_bwd(const float* mu, const float* var, const float* eps, const int64_t* y,
                          const float* gnll, float* gmu, float* gvar, int S, int F, int C, int B,
                          vargp_stream_t stream);
int vargp_softmax_predict(const float* mu, const float* var, const float* eps, float* probs, int S, int F,
                          int C, int B, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Yogi optimiser step, fused over one flat parameter buffer (reference call site:
 * experiments/vargp.py:23,37 -> torch_optimizer.Yogi; algorithm from Zaheer et al. 2018).
 * bias1/bias2 = 1 - beta^t; if `step` (device pointer to the step count t as a float) is not NULL
 * the corrections are computed on the device from it instead, which keeps a captured graph valid.
 */
int vargp_yogi_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float bias1, float bias2, const float* step, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The first-task ELBO as one native program (vargp_amd/csrc/elbo_t0.hip).
 * Replaces, for a model without previous tasks, the whole of VARGP.loss (var_gp/vargp.py:156-194:
 * RBFKernel.sample_hypers / kl_hypers kernels.py:62-77, RBFKernel.compute kernels.py:24-56, cholesky and
 * linear_marginal_diag gp_utils.py:5-11,150-191, the MVN KL vargp.py:182-190, MulticlassSoftmax.loss
 * likelihoods.py:13-45) and its autograd backward (loss.backward(), experiments/vargp.py:35).
 *
 * Shapes: log_mean, log_logvar, prior_* (D+1); eps_theta (S, D+1); z (C, M, D); u_mean (C, M); u_tril_vec
 * (C, M(M+1)/2); x (B, D); y (B) int64; eps_f (S, F, C, B).  map_est != 0: theta = log_mean, S must be 1, kl_hypers = 0
 * (log_logvar / prior_* / eps_theta may be NULL).  eps_theta = eps_f = NULL: native noise, see rng_* below.
 * fwd writes scalars[0..2] = (kl_hypers, kl_u, nll) and info[0 .. S*C + C) (Cholesky status of K_uu[s,c] + eps I, then of
 * S_u[c] + eps I; 0 = ok, k = leading minor k not positive, results NaN-filled).
 * bwd needs the workspace exactly as fwd left it; seeds (device, 3 floats) = d total / d (kl_hypers, kl_u, nll).  It
 * OVERWRITES the five gradient buffers (shapes of the parameters).  Neither call allocates or synchronises.
 */
typedef struct vargp_elbo_t0_desc {
  int32_t S, C, M, D, B, F;
  int32_t map_est;
  float jitter;
  const float *log_mean, *log_logvar, *prior_log_mean, *prior_log_logvar;
  const float *z, *u_mean, *u_tril_vec;
  const float* x;
  const int64_t* y;
  const float *eps_theta, *eps_f;
  float* scalars;
  int32_t* info;
  void* ws;
  size_t ws_bytes;
  float* bump; /* optional: fwd adds 1.0f to *bump (a caller's device-side step counter rides along for free) */
  /* Native noise: with eps_f == NULL (and eps_theta == NULL) fwd draws both noise tensors itself from a counter-based
   * generator (Philox4x32-10 + Box-Muller) keyed by rng_seed; *rng_counter (device) is the step number, read by fwd's
   * first kernel and incremented by a later one.  Element (s, ...) is a function of the GLOBAL sample index
   * rng_sample_offset + s, so sample-parallel ranks (offset = rank * S, same seed and counter) see slices of one global
   * draw.  The drawn tensors stay in the workspace (bwd reads them there). */
  uint64_t rng_seed;
  uint32_t* rng_counter;
  int32_t rng_sample_offset;
  /* 1: bwd leaves the gradients of log_mean / log_logvar to vargp_yogi_step_multi_hyper (see vargp_hyper_grad_desc): it does
   * not touch g_log_mean / g_log_logvar (they may be NULL) and launches one kernel less */
  int32_t defer_hyper;
  /* 1 (only when a vargp_elbo_t0_bwd on this workspace follows before scalars[2] is read): for the shapes of the LDS-resident
   * backward with C <= 16 and F <= 16 the forward does NOT launch the Monte-Carlo softmax likelihood; the backward's tile kernel
   * evaluates it (value and gradient) and adds nll into scalars[2], which is therefore valid only after bwd.  Ignored (the
   * forward evaluates the likelihood as usual) for every other shape, and for a caller-supplied eps_f that does not sit on a
   * 16-byte boundary (the tile kernel reads it as float4). */
  int32_t defer_softmax;
  /* 1: the Monte-Carlo likelihood is the CALLER's -- a rank of a class-sharded step holds only some of the classes the
   * softmax couples (likelihoods.py:26-29).  fwd stops at the predictive moments and the KL (scalars[2] stays 0); y and eps_f
   * are not read (may be NULL), eps_theta must be given unless map_est.  Between fwd and bwd the caller stores
   * d total / d mu and d total / d var -- ALREADY multiplied by their seed -- into the workspace's gmu, gvar (S, C, B)
   * (vargp_elbo_t0_lik_buffers); bwd takes them as they are and ignores seeds[2].  defer_softmax is ignored. */
  int32_t ext_lik;
  /* Optional early hand-over of the Cholesky status (both NULL: none).  Right behind the launch that writes info -- the second of
   * the forward's four -- fwd enqueues a copy of info[0 .. S*C + C) to info_host (host memory, pinned for the copy to be
   * asynchronous) and records info_event (a hipEvent_t of the caller's) on the stream.  A caller that has to raise on a failed
   * factorisation before it returns (the reference raises inside torch.cholesky, gp_utils.py:5-11) waits for that event instead
   * of the whole forward: the rest of the forward runs while the host carries on.  fwd itself still does not synchronise. */
  int32_t* info_host;
  void* info_event;
} vargp_elbo_t0_desc;
/* Streams: the work is issued to `stream` in order.  With many hyper-samples (S C + C per-matrix chains > a third of the CUs) fwd and
 * bwd fork a library-owned side stream for the pivot / adjoint chains and join it before they return (events; graph dependencies
 * under hipGraph capture): to the caller the call is still one unit of work on `stream`.  VARGP_T0_SIDE=0 disables it. */
size_t vargp_elbo_t0_workspace_bytes(int S, int C, int M, int D, int B, int F);
int vargp_elbo_t0_fwd(const vargp_elbo_t0_desc* d, vargp_stream_t stream);
/* Where the program keeps the predictive moments mu, var (S, C, B) of its last fwd and the likelihood gradients gmu, gvar
 * (S, C, B) its bwd reads (pointers into d->ws; any of the four outputs may be NULL). */
int vargp_elbo_t0_lik_buffers(const vargp_elbo_t0_desc* d, float** mu, float** var, float** gmu, float** gvar);
/* ONE vargp_elbo_t0_bwd per vargp_elbo_t0_fwd: for the shapes of the LDS-resident backward (M <= 104, M % 4 == 0, B % 4 == 0,
 * D % 4 == 0, S <= 64) the forward clears the accumulators the backward adds into (and the tile counters of the backward's
 * persistent product workgroups) -- there is no clearing launch in bwd.
 * Enforced by the library (host-side state per workspace, checked when the call is issued): a second bwd on one fwd, or a bwd
 * whose z / x alignment differs from its forward's, returns VARGP_EINVAL with a message instead of accumulating into stale sums. */
int vargp_elbo_t0_bwd(const vargp_elbo_t0_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar, float* g_z,
                      float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The ELBO of a model WITH previous tasks as one native program (vargp_amd/csrc/elbo_tn.hip); nblk = 1 is a first-task
 * model of any M.  Replaces VARGP.compute_q / compute_pf_diag / forward / loss for ep_var_mean = True
 * (var_gp/vargp.py:35-194: the linear_joint chain gp_utils.py:101-147 over the earlier tasks, linear_marginal_diag
 * gp_utils.py:150-191, the conditional prior gp_cond gp_utils.py:68-98 + MVN KL vargp.py:146-190, RBFKernel kernels.py:24-77,
 * MulticlassSoftmax.loss likelihoods.py:13-45) and its autograd backward.  The chain is evaluated in its block form: ONE
 * kernel matrix over all Mt = nblk * M inducing points, ONE factorisation L = chol(K + eps I) with T = L^-1 (every
 * factor of the chain is a leading block of L), block-diagonal small products with T_ii, three Mt x Mt x B GEMMs
 * (tests/block_algorithm.py pins the identities against the chain in fp64).
 *
 * Caller-maintained packed operands (the program writes only the CURRENT task's part, on every fwd):
 *   z_all  (C, Mt, D):        rows [i*M, (i+1)*M) of class c = inducing points of task i; the last M rows are scratch
 *   rk_all (C, nblk, M, NR):  NR = 4 + M rounded up to 4;  row j of block i = [ u_mean_i[j] | 0 0 0 | Lu_i[j, :] | 0.. ],
 *                             Lu_i = vec2tril(u_tril_vec_i) (gp_utils.py:22-49); the last block is scratch
 * Other shapes as vargp_elbo_t0_desc.  y == NULL: predictive moments only (no likelihood / KL; eps_theta must be given
 * unless map_est) -- VARGP.forward / predict.  vargp_elbo_tn_moments returns where fwd left mu, var (S, C, B).
 * info[0 .. S*C): Cholesky status of K(z_<=t) + eps I per (s, c).
 */
typedef struct vargp_elbo_tn_desc {
  int32_t S, C, M, D, B, F, nblk;
  int32_t map_est;
  float jitter;
  const float *log_mean, *log_logvar, *prior_log_mean, *prior_log_logvar;
  const float *z, *u_mean, *u_tril_vec; /* current task */
  float* z_all;
  float* rk_all;
  const float* x;
  const int64_t* y;
  const float *eps_theta, *eps_f;
  float* scalars;
  int32_t* info;
  void* ws;
  size_t ws_bytes;
  float* bump;
  uint64_t rng_seed;
  uint32_t* rng_counter;
  int32_t rng_sample_offset;
  int32_t forward_only; /* 1: workspace sized by vargp_elbo_tn_workspace_bytes_fwd; moments only (y == NULL), no bwd / end */
  int32_t defer_hyper;  /* as vargp_elbo_t0_desc.defer_hyper (vargp_elbo_tn_bwd only) */
  int32_t ext_lik;      /* as vargp_elbo_t0_desc.ext_lik (vargp_elbo_tn_fwd / _bwd; buffers: vargp_elbo_tn_lik_buffers); y must still be non-NULL
                         * for fwd to evaluate the KL (it is not dereferenced) */
  /* ep_var_mean = False (reference var_gp/vargp.py:137-152, VARGP(..., ep_var_mean=False)): the variational mean of the
   * current task is NOT offset by the conditional prior's mean, so the KL keeps that mean, evaluated at n_v samples
   * u_<t ~ q(u_<t | theta): eps_u (n_v, S, C, (nblk - 1) M) standard normal, 1 <= n_v <= 16.  no_var_mean = 1 selects it
   * (nblk > 1, vargp_elbo_tn_fwd / _bwd only; ignored by the tiled calls); 0: ep_var_mean = True, eps_u is not read. */
  const float* eps_u;
  int32_t n_v;
  int32_t no_var_mean;
  /* as vargp_elbo_t0_desc.info_host / info_event (vargp_elbo_tn_fwd only): info[0 .. S*C) is copied out and the event recorded
   * right behind the blocked factorisation, before the products that consume it */
  int32_t* info_host;
  void* info_event;
} vargp_elbo_tn_desc;
size_t vargp_elbo_tn_workspace_bytes(int S, int C, int M, int D, int B, int F, int nblk);
/* Workspace of a program that only ever evaluates predictive moments (VARGP.forward / predict, var_gp/vargp.py:115-131,
 * 196-198: no likelihood, no KL, no backward): none of the gradient buffers.  Set d->forward_only = 1. */
size_t vargp_elbo_tn_workspace_bytes_fwd(int S, int C, int M, int D, int B, int F, int nblk);
int vargp_elbo_tn_fwd(const vargp_elbo_tn_desc* d, vargp_stream_t stream);
int vargp_elbo_tn_bwd(const vargp_elbo_tn_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar, float* g_z,
                      float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream);
int vargp_elbo_tn_moments(const vargp_elbo_tn_desc* d, float** mu, float** var);
int vargp_elbo_tn_lik_buffers(const vargp_elbo_tn_desc* d, float** mu, float** var, float** gmu, float** gvar);
/* The same ELBO over a data set swept in minibatch tiles (BASELINE config 5: N = 1e6, M = 2048, K_uf tiled in HBM): loss
 * AND gradient, with everything that does not depend on the data (kernel matrix of the inducing points, factorisation,
 * small products, KL) computed once.  Workspace / descriptor as for fwd with d->B = the widest tile; d->y must be non-NULL
 * (any label pointer), d->x is not read.
 *   begin: theta, K(z_<=t), L, T, the small products, kl_hypers and kl_u into scalars[0..1], scalars[2] = 0, accumulators = 0
 *   tile : x (Bt, D), y (Bt), Bt <= d->B, eps_f (S, F, C, Bt) or NULL (native noise, one generator step per tile):
 *          scalars[2] += the tile's nll; the tile's share of every gradient is accumulated.
 *          y == NULL (seeds, eps_f ignored): the predictive moments mu, var (S, C, Bt) of the tile only, left where
 *          vargp_elbo_tn_moments says (row stride Bt) -- the predictive sweep VARGP.predict(x, tile=) of a model with or
 *          without previous tasks (var_gp/vargp.py:196-198 called per batch by var_gp/train_utils.py:21-35), with the
 *          x-independent part (begin) done once; the only tile mode of a forward_only program
 *   end  : Cholesky / kernel-matrix backward of the accumulated gradients; OVERWRITES the five gradient buffers with the
 *          gradient of seeds . (kl_hypers, kl_u, sum of the tiles' nll)
 * seeds (device, 3 floats) must be the same pointer contents for every tile and for end. */
int vargp_elbo_tn_begin(const vargp_elbo_tn_desc* d, vargp_stream_t stream);
int vargp_elbo_tn_tile(const vargp_elbo_tn_desc* d, const float* seeds, const float* x, const int64_t* y, const float* eps_f,
                       int Bt, vargp_stream_t stream);
int vargp_elbo_tn_end(const vargp_elbo_tn_desc* d, const float* seeds, float* g_log_mean, float* g_log_logvar, float* g_z,
                      float* g_u_mean, float* g_u_tril_vec, vargp_stream_t stream);

/* The last step of either program's backward -- theta-gradient -> variational hyper-parameters (RBFKernel.sample_hypers /
 * kl_hypers, var_gp/kernels.py:62-77: theta = mean + eps exp(logvar / 2), plus the gamma^2 of the predictive variance and the
 * gradient of kl_hypers scaled by its seed) -- as data: what the deferred form needs to finish it inside the optimiser's
 * launch.  Filled by vargp_elbo_t0_hyper_desc / vargp_elbo_tn_hyper_desc after a bwd with defer_hyper = 1 (pointers into
 * the program's workspace: valid until its next fwd). */
typedef struct vargp_hyper_grad_desc {
  const float *log_mean, *log_logvar, *prior_log_mean, *prior_log_logvar;
  const float *eps_theta, *gtheta, *g2, *gkd, *seeds;
  int32_t S, C, D1, map_est;
} vargp_hyper_grad_desc;
int vargp_elbo_t0_hyper_desc(const vargp_elbo_t0_desc* d, const float* seeds, vargp_hyper_grad_desc* out);
int vargp_elbo_tn_hyper_desc(const vargp_elbo_tn_desc* d, const float* seeds, vargp_hyper_grad_desc* out);
/* vargp_yogi_step_multi with the gradients of tensors idx_mean (log_mean) and idx_logvar (log_logvar; -1 under map_est)
 * computed on the fly from h (and stored to g[idx_*] as well): the step of experiments/vargp.py:35-37 with one launch less. */
int vargp_yogi_step_multi_hyper(int ntensors, float* const* p, float* const* g, float* const* m, float* const* v,
                                const int64_t* n, float lr, float beta1, float beta2, float eps, const float* step,
                                int step_mode, const vargp_hyper_grad_desc* h, int idx_mean, int idx_logvar,
                                vargp_stream_t stream);

/* Minibatch i of an epoch, gathered on the device: x[r, :] = data[perm[i B + r], :], y[r] = targets[perm[i B + r]], r < B, with
 * i = (int)(*step_now - *step_base) read from DEVICE memory -- step_now is a step counter some kernel of the step advances (the
 * `bump` of the ELBO programs' descriptors), step_base its value at the start of the epoch.  Replaces the two index_select
 * launches per step of the training loop (reference: DataLoader(train_set, batch_size, shuffle=True), experiments/vargp.py:26)
 * by a launch that can sit INSIDE a captured graph of K steps: the minibatch index advances on the device, the host launches
 * graphs.  perm: n int64 indices (one permutation per epoch); rows past n clamp to the last index. */
int vargp_gather_minibatch(const float* data, const int64_t* targets, const int64_t* perm, const float* step_now,
                           const float* step_base, int64_t n, int B, int D, float* x, int64_t* y, vargp_stream_t stream);

/* Same update for up to 8 tensors in one launch.  `step` (device float) = the step count t.
 * step_mode 0: use t as is.  1: use t + 1 (the caller advances the stored count elsewhere, e.g. through the `bump`
 * pointer of vargp_elbo_t0_desc, so that the optimiser needs no "t += 1" launch of its own). */
int vargp_yogi_step_multi(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                          const int64_t* n, float lr, float beta1, float beta2, float eps, const float* step,
                          int step_mode, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Variational kernel hyper-parameters (reference: RBFKernel.sample_hypers / kl_hypers,
 * var_gp/kernels.py:62-77).  D1 = D + 1.
 *   sample: theta[S, D1] = mean + eps * exp(logvar / 2)           (Normal.rsample)
 *   kl:     sum_d KL(N(mean_d, e^logvar_d) || N(prior_mean_d, e^prior_logvar_d))
 */
int vargp_hyper_sample_fwd(const float* mean, const float* logvar, const float* eps, float* theta, int S, int D1,
                           vargp_stream_t stream);
int vargp_hyper_sample_bwd(const float* logvar, const float* eps, const float* gtheta, float* gmean,
                           float* glogvar, int S, int D1, vargp_stream_t stream);
int vargp_hyper_kl_fwd(const float* mean, const float* logvar, const float* prior_mean, const float* prior_logvar,
                       float* kl, int D1, vargp_stream_t stream);
int vargp_hyper_kl_bwd(const float* mean, const float* logvar, const float* prior_mean, const float* prior_logvar,
                       const float* gkl, float* gmean, float* glogvar, int D1, vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Deep-kernel feature map (reference: DeepRBFKernel.phi, var_gp/kernels.py:80-96 = Linear(D,256) - ReLU - Linear(256,256)
 * - ReLU - Linear(256,64) in front of the RBF kernel).  The three matrix products are vargp_bgemm; these two fuse the
 * bias and the activation around them:  y[r,c] = act(x[r,c] + bias[c]),  act = ReLU if relu else identity;
 * backward: gx = gy * (y > 0) (or gy), gbias[c] = sum_r gx[r,c]  (gbias is zeroed by the call).
 */
int vargp_bias_act_fwd(const float* x, const float* bias, float* y, int64_t rows, int cols, int relu, vargp_stream_t stream);
int vargp_bias_act_bwd(const float* y, const float* gy, float* gx, float* gbias, int64_t rows, int cols, int relu,
                       vargp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart): when enabled, the heavy launches are bracketed by
 * hipEvents on their own stream, tagged "rbf_kuf" / "rbf_kuu" (distance GEMM incl. the split-K combine pass),
 * "rbf_kuf_gemm", "rbf_kuu_gemm", "rbf_kuf_bwd_gemm", "rbf_kuu_bwd_gemm", "chol_inv_small", "bgemm".
 * vargp_prof_read sums and clears one tag.
 */
int vargp_prof_enable(int on);
int vargp_prof_read(const char* tag, double* total_ms, int64_t* launches);
/* Launch replay, for timing one kernel of a step that otherwise runs inside a captured graph (hipEvents cannot
 * bracket a graph node, and an event pair around a single launch also times the dispatch gap):
 * vargp_prof_remember(1) makes the tagged launches ("rbf_kuf_gemm", "rbf_kuu_gemm", "rbf_kuu_bwd_gemm" = the pair
 * launch of the two W.Y products, "chol_rbf_gemm" = factorisations + K_uf GEMM, "bgemm", ...) keep a copy of their
 * arguments; vargp_prof_replay re-launches the most recent one with that tag `iters` times back to back between ONE
 * pair of hipEvents on `stream` and returns the average time per launch in microseconds.  The buffers of the
 * remembered launch must still be alive.  Synchronises. */
int vargp_prof_remember(int on);
/* Tuning aid: force the tile shape of subsequent vargp_bgemm / internal GEMM launches (0 = automatic choice,
 * 1 = 128x128x16, 2 = 128x64x32, 3 = 64x64x64).  Results do not depend on it beyond summation order. */
int vargp_tune_gemm_tile(int tile);
int vargp_prof_replay(const char* tag, int iters, double* avg_us, vargp_stream_t stream);
/* Time line of the first-task step as the GPU sees it -- also inside a replayed hipGraph, where hipEvents cannot bracket a
 * node: while switched on, every kernel of the step stamps the device's constant 100 MHz wall clock at its first
 * workgroup's start and (atomic max) at every workgroup's end into a small device table.
 *   mode 1: clear the table and switch the stamps on;  mode 2: switch them off;
 *   mode 0: copy the table to out[12][2] = (start, end) ticks of 10 ns per slot, 0 = the slot's kernel did not run;
 *   mode 3: out[12][2] = (shader-clock ticks (s_memtime), wall-clock ticks of 10 ns) that went by between the start and the end
 *           of WORKGROUP 0 of each slot's last launch: 100 MHz x the ratio = the clock the chip held under that kernel.
 * Slots: 0 t0_pro_kuu, 1 chol_rbf_gemm (9: end of its last factorisation), 2 gemm_kernel (any plain product: the last one
 * launched), 3 t0_fwd_fused, 4 t0_bwd_mid, 5 t0_bwdmat_gemm (8: end of its last matrix chain), 6 t0_puu_final,
 * 7 yogi_multi; 10 / 11: end of the Gram / row-norm role of t0_pro_kuu.  Synchronous (hipMemcpy to / from device symbols): call between steps, not inside a capture. */
int vargp_prof_spans(int mode, unsigned long long* out);

#ifdef __cplusplus
}
#endif
#endif /* VARGP_HIP_H */
