// Shared helpers for the libvargp_hip kernels (gfx950 / CDNA4 only: wave64, f32 MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include "../../include/vargp_hip.h"

// Step time line (include/vargp_hip.h: vargp_prof_spans): wall-clock (100 MHz) of the first workgroup's start and the last
// workgroup's end of each kernel of the first-task step -- the step as the GPU sees it, also inside a replayed hipGraph, where
// hipEvents cannot bracket a node.  One table per translation unit, switched on at run time (g_spans_on_<tu>: one scalar load
// per workgroup while off).
#if defined(__HIPCC__)
namespace vargp {
constexpr int kSpanEnds = 4096;      // per-workgroup end stamps of one kernel (plain stores: an atomic max on ONE address from
                                     // thousands of workgroups serialises in the memory-side cache and stretches the kernel it times)
struct SpanGuard {
  unsigned long long* p;             // start slot; the kernel's end stamps follow the table at p + kSpanEndsOffset
  unsigned long long* e;
  __device__ __forceinline__ SpanGuard(unsigned long long* q, unsigned long long* ends, int on) : p(on ? q : nullptr), e(ends) {
    // workgroup 0 also reads the SHADER clock (s_memtime) beside the 100 MHz wall clock at its start and at its end: the
    // ratio of the two differences is the clock the chip held while this workgroup ran (vargp_prof_spans mode 3)
    if (p && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      p[0] = wall_clock64();
      p[2] = __builtin_amdgcn_s_memtime();
    }
  }
  __device__ __forceinline__ ~SpanGuard() {
    if (p && threadIdx.x == 0) {
      const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      if (lin == 0) p[3] = __builtin_amdgcn_s_memtime();
      e[lin & (kSpanEnds - 1)] = wall_clock64();
    }
  }
};
}  // namespace vargp
// vargp_debug_spans_<tu>(out, mode): mode 0 = out[12][2] = (start, max of the end stamps) per slot; 1 = clear the tables and
// switch the stamps on; 2 = switch them off; 3 = out[12][2] = (shader-clock, wall-clock) ticks of each slot's workgroup 0
#define STEP_SPAN_TABLE(tu)                                                                                      \
  __device__ unsigned long long g_spans_##tu[12][4];                                                              \
  __device__ unsigned long long g_span_ends_##tu[12][vargp::kSpanEnds];                                           \
  __device__ int g_spans_on_##tu;                                                                                 \
  extern "C" void vargp_debug_spans_##tu(unsigned long long* out, int mode) {                                    \
    if (mode == 3) {         /* out[12][2] = (shader-clock ticks, wall-clock ticks) of workgroup 0 of each slot */ \
      unsigned long long st[12][4];                                                                              \
      static unsigned long long ends0[12][vargp::kSpanEnds];                                                     \
      (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_spans_##tu), sizeof(st));                                       \
      (void)hipMemcpyFromSymbol(ends0, HIP_SYMBOL(g_span_ends_##tu), sizeof(ends0));                             \
      for (int i = 0; i < 12; ++i) {                                                                             \
        const bool ok = st[i][3] > st[i][2] && ends0[i][0] > st[i][0] && st[i][0] != 0;                          \
        out[2 * i] = ok ? st[i][3] - st[i][2] : 0; out[2 * i + 1] = ok ? ends0[i][0] - st[i][0] : 0;            \
      }                                                                                                          \
    } else if (mode) {                                                                                           \
      const int on = mode == 1;                                                                                  \
      if (on) {                                                                                                  \
        void* sp = nullptr; void* ep = nullptr;                                                                  \
        (void)hipGetSymbolAddress(&sp, HIP_SYMBOL(g_spans_##tu));                                                \
        (void)hipGetSymbolAddress(&ep, HIP_SYMBOL(g_span_ends_##tu));                                            \
        if (sp) (void)hipMemset(sp, 0, sizeof(unsigned long long) * 12 * 4);                                     \
        if (ep) (void)hipMemset(ep, 0, sizeof(unsigned long long) * 12 * vargp::kSpanEnds);                      \
      }                                                                                                          \
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spans_on_##tu), &on, sizeof(on));                                     \
    } else {                                                                                                     \
      static unsigned long long ends[12][vargp::kSpanEnds];                                                      \
      unsigned long long st[12][4];                                                                              \
      (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_spans_##tu), sizeof(st));                                       \
      (void)hipMemcpyFromSymbol(ends, HIP_SYMBOL(g_span_ends_##tu), sizeof(ends));                               \
      for (int i = 0; i < 12; ++i) {                                                                             \
        unsigned long long mx = st[i][1];                                                                        \
        for (int k = 0; k < vargp::kSpanEnds; ++k) mx = ends[i][k] > mx ? ends[i][k] : mx;                       \
        out[2 * i] = st[i][0]; out[2 * i + 1] = mx;                                                              \
      }                                                                                                          \
    }                                                                                                            \
  }
#define STEP_SPAN(tu, i) vargp::SpanGuard step_span_guard_(g_spans_##tu[i], g_span_ends_##tu[i], g_spans_on_##tu)
// (the end of one role of a multi-role kernel -- a few dozen workgroups: slot i holds the last one's end)
#define STEP_SPAN_MARK(tu, i) do { if (step_span_guard_.p && threadIdx.x == 0) atomicMax(&g_spans_##tu[i][1], wall_clock64()); } while (0)
#else
#define STEP_SPAN_TABLE(tu)
#define STEP_SPAN(tu, i) do { } while (0)
#define STEP_SPAN_MARK(tu, i) do { } while (0)
#endif

namespace vargp {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);
// Zero `bytes` (multiple of 4) at `p` with a kernel.  Used instead of hipMemsetAsync everywhere: two adjacent
// memset nodes in a captured hipGraph were observed to lose the first one on replay (ROCm 7.2).
void zero_async(void* p, size_t bytes, hipStream_t st);
int check_launch(const char* what);

#define VARGP_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      vargp::set_error(__VA_ARGS__);        \
      return VARGP_EINVAL;                  \
    }                                       \
  } while (0)

static inline hipStream_t as_stream(vargp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions (64-lane wavefronts) -------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in every thread
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red /* >= NT/64 floats of LDS */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// internal GEMM parameter block (superset of vargp_gemm_desc: adds the fused RBF epilogue)
struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* D;
  int M, N, K;
  int lda, ldb, ldc, ldd;
  int nb1, nb2;
  int64_t sA[3], sB[3], sC[3], sD[3];
  float alpha, beta;
  int triA, triB, triC;
  // RBF epilogue: C = g2[b0] * exp(-0.5 * (na[row] + nb[col] - 2 acc)); A is scaled by kscale[k]
  const float* kscale;  // [nb0][ks_ld] : 1/sigma^2
  int64_t ks_ld;
  const float* g2;      // [nb0]
  const float* na;      // row norms, batch strides sNa
  const float* nbv;     // col norms, batch strides sNb
  int64_t sNa[3], sNb[3];
  int same_xy;
  int nofast;   // tuning aid: force the guarded (non-pipelined) slab loop
  int xcd_remap;   // set by launch_gemm: XCD-compact workgroup -> tile map
  int group_m;     // set by launch_gemm: > 1: tiles of a matrix are visited group_m tile rows at a time (L2 reuse on large grids)
  // symout products only: C_ii += diag_scale * diag_ptr[0] for i >= diag_from (diag_ptr: device scalar, NULL: off)
  const float* diag_ptr;
  float diag_scale;
  int diag_from;
  int tile;        // 0: launch_gemm's heuristic; 1 / 2 / 3: 128x128x16 / 128x64x32 / 64x64x64 (callers that measured)
  int symout;      // plain products: write C_ij = C_ji = value for j <= i, nothing from above the diagonal (use with triC = 2)
  // split-K: each split covers a BK-aligned share of [0, K).  RBF products write their partial inner products to
  // C + split * sSplit and a second kernel applies the epilogue; plain products (sSplit = 0) accumulate into a
  // pre-zeroed C with float atomics (no D)
  int splitk;
  int64_t sSplit;
};

// Fork / join of a side stream inside one library call (core.hip): independent launches of a call that should overlap on the
// device.  SideFork(st): the side stream waits for everything issued to `st` so far; side() = the stream to issue the branch to
// (== st when no side stream is to be had: then the branch simply runs in line); join(): `st` waits for the branch.  One side
// stream and two events per device, created on first use outside a stream capture (a first use under capture runs in line);
// under capture the fork / join become graph dependencies.  The issue sequence fork .. join holds a per-device mutex.
class SideFork {
 public:
  explicit SideFork(hipStream_t st, bool want = true);
  ~SideFork();
  hipStream_t side() const { return side_; }
  bool forked() const { return forked_; }
  int join();
 private:
  hipStream_t st_, side_;
  bool forked_;
  int dev_;
};

int launch_gemm(const GemmParams& p, int transA, int transB, int nbatch, bool rbf, hipStream_t st,
                const char* tag = "bgemm");
int launch_gemm_pair(const GemmParams& p0, int nbatch0, const GemmParams& p1, int nbatch1, int transA, int transB,
                     bool rbf, hipStream_t st, const char* tag0, const char* tag1);
int launch_gemm_pair2(const GemmParams& p0, int tA0, int tB0, int nbatch0, const GemmParams& p1, int tA1, int tB1,
                      int nbatch1, hipStream_t st, const char* tag);
bool chol_rbf_gemm_applicable(int n, const GemmParams& p);
// Extra destination for the factors L of the batch entries b >= first: ncopy copies, entry (b - first) of copy c at
// base + (b - first) * stride_b + c * stride_copy, row stride ld (the ELBO program wants L_S inside its RK operand).
// symmetric_input: both triangles of every input matrix are valid (coalesced direct load instead of the mirrored one).
// diag_only_before_first: for the batch entries b < first only the diagonal of L is wanted (the rest is written as 0).
// part != NULL: the matrices b < first arrive as nsplit (<= kCholPartMax) partial Gram matrices G_q at part + q * sSplit + b n^2
// (the K-split inner products of an RBF kernel matrix over ONE point set); the factorising workgroup forms
// K_ij = g2[b / part_C] exp(-(G_ii + G_jj - 2 G_ij) / 2), G = sum_q G_q, exactly g2 on the diagonal, as it loads, and stores
// Zero-fills that ride in some other launch's spare workgroups: rows x width floats at p, row stride ld (p == NULL: none)
// stair_nb > 0 (blocked factorisation, chol.hip): `rows` rows of batched stair_n x stair_n matrices (ld = stair_n); row r gets zeros
// right of its diagonal block only, columns [((r % stair_n) / stair_nb + 1) stair_nb, stair_n) -- one workgroup per row, float4 stores
struct ZeroJob { float* p; int64_t rows, width, ld; int stair_n, stair_nb; };
constexpr int kZeroJobs = 6;
struct ZeroJobs { ZeroJob j[kZeroJobs]; };
#ifdef __HIPCC__
__device__ __forceinline__ void zero_jobs_role(const ZeroJobs& z, int blk, int nblk) {
#pragma unroll
  for (int q = 0; q < kZeroJobs; ++q) {
    const ZeroJob t = z.j[q];
    if (!t.p) continue;
    if (t.stair_nb > 0) {
      const int n = t.stair_n;
      if ((n & 3) == 0 && (t.stair_nb & 3) == 0 && (reinterpret_cast<uintptr_t>(t.p) & 15) == 0) {
        // q float4 per row; short rows: G = 256 / q rows per pass, one float4 per thread; long rows: one row per pass
        const int q = n >> 2, G = q >= 256 ? 1 : 256 / q;
        const int sub = G > 1 ? (int)threadIdx.x / q : 0, l4 = G > 1 ? (int)threadIdx.x - sub * q : (int)threadIdx.x;
        for (int64_t rb = (int64_t)blk * G; rb < t.rows; rb += (int64_t)nblk * G) {
          const int64_t r = rb + sub;
          if (sub < G && r < t.rows) {
            const int c0 = (((int)(r % n)) / t.stair_nb + 1) * t.stair_nb;
            float* row = t.p + r * t.ld;
            for (int c = 4 * l4; c < n; c += 1024)
              if (c >= c0) *reinterpret_cast<float4*>(row + c) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      } else {
        for (int64_t r = blk; r < t.rows; r += nblk) {
          const int c0 = (((int)(r % n)) / t.stair_nb + 1) * t.stair_nb;
          float* row = t.p + r * t.ld;
          for (int c = c0 + (int)threadIdx.x; c < n; c += 256) row[c] = 0.f;
        }
      }
      continue;
    }
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < t.rows * t.width; i += (int64_t)nblk * 256)
      t.p[(i / t.width) * t.ld + i % t.width] = 0.f;
  }
}
#endif
// K to Kout + b n^2 (dense, both triangles).
constexpr int kCholPartMax = 4;
// arithmetic of the pivot chains inside the merged first-task launch (chol_small3.h: R): 0 = fp64, 1 = the reference's own fp32
// (torch.cholesky + triangular_solve in float32, gp_utils.py:5-11; measured accuracy = LAPACK fp32's).  Every other
// factorisation (stand-alone op, n <= 50 kernel, diagonal blocks of the blocked path) stays fp64.
constexpr int kCholF32Default = 1;
// 1: fp32 chains of 64 < n <= 100 run the blocked elimination on the matrix core (chol_blk16.h) wherever the matrix passes through
// the staging area in LDS; 0: the register-resident elimination of chol_small3.h everywhere
#ifndef VARGP_CHOL_BLK16
#define VARGP_CHOL_BLK16 1
#endif
constexpr int kProKuuMaxD = 4096;       // launch_pro_kuu: 1/sigma^2 of one hyper-sample staged in LDS by the norm role
struct CholExtra {
  float* base; int first; int ld; int64_t stride_b, stride_copy; int ncopy;
  int symmetric_input, diag_only_before_first;
  const float* part; int nsplit; int64_t sSplit; const float* g2; int part_C; float* Kout;
  // gram_z != NULL (chol_gram.h; instead of `part`): the matrices b < first = (s, c) are built by their factorising workgroup from the
  // inducing points gram_z [part_C][n][gram_D] and 1/sigma^2 gram_w [S][gram_Dp] (g2, part_C, Kout as for `part`)
  const float* gram_z; const float* gram_w; int gram_D; int64_t gram_Dp;
  // su_Lu != NULL (fp32 chains of 64 < n <= 100 only): the matrices b >= first are S_u[c] = Lu[c] Lu[c]^T, c = b - first, built by
  // their factorising workgroup from su_Lu [ncls][n][n] (lower triangular, zeros above) -- nothing is read from A for them
  const float* su_Lu;
};
int vargp_cu_count();       // CUs of the current device (gemm.hip)
int launch_chol_rbf_gemm(const float* A, float eps, float* L, float* T, int32_t* info, int nchol, int n,
                         const GemmParams& p, int nbatch, hipStream_t st, const CholExtra* extra = nullptr,
                         const ZeroJobs* zero = nullptr);
// number of K splits launch_gemm will use for an RBF product of this shape (1 = fused epilogue, no partials)
int rbf_splitk(int M, int N, int K, int nbatch);

constexpr int kRbfDirectD = 32;   // D <= this: kernel matrices from the direct (no-cancellation) distance form
int rbf_direct_launch(const float* X, const float* Y, const float* w, const float* g2, float* K, int64_t ldk, int S,
                      int C, int M, int N, int D, int64_t Dp, int y_shared, hipStream_t st);
int rbf_gram_fwd_impl(const float* theta, const float* X, const float* Y, float* K, int S, int C, int M, int N, int D,
                      int y_shared, void* ws, size_t ws_bytes, int sym_out, hipStream_t st);
int rbf_gram_bwd_impl(const float* theta, const float* X, const float* Y, const float* K, const float* gK, float* gX,
                      float* gY, float* gtheta, int S, int C, int M, int N, int D, int y_shared, int accumulate, void* ws,
                      size_t ws_bytes, int sym_gk, hipStream_t st);
int chol_inv_fwd_impl(const float* A, float eps, float* L, float* T, float* logdet, int32_t* info, int nbatch, int n,
                      void* ws, size_t ws_bytes, bool zero_info, hipStream_t st, const GemmParams* co = nullptr,
                      int co_nbatch = 0, int* co_done = nullptr, int nco = 1, bool chain_f32 = false);
// factorisations of nchol matrices (n in (50, 100]) with explicit leading dimensions / batch strides + one RBF GEMM
int launch_chol_rbf_gemm_ld(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T, int ldt,
                            int64_t sT, int32_t* info, int nchol, int n, const GemmParams& p, int nbatch, hipStream_t st,
                            bool chain_f32 = false, const ZeroJobs* zero = nullptr);
// pivot chains of nchol diagonal blocks (fp64, info_base as the stand-alone kernel) || a plain NN product they do not feed
bool chol_nn_gemm_applicable(int n, const GemmParams& p);
int launch_chol_nn_gemm(const float* A, int lda, int64_t sA, float eps, float* L, int ldl, int64_t sL, float* T, int ldt, int64_t sT,
                        int32_t* info, int info_base, int nchol, int n, const GemmParams& p, int nbatch, hipStream_t st,
                        const ZeroJobs* zero = nullptr);
// w = exp(-2 theta) (zero-padded to Dp), g2 = exp(2 theta_D) and the weighted squared row norms of x (xrows x D) and of
// y (yrows x D, may be 0 rows) for every hyper-sample, in one launch
// ys / xs (nullable): also write y o w, [S][yrows][D], and x o w, [S][xrows][D] (the pre-scaled operand of an unscaled RBF
// GEMM: GemmParams.kscale = NULL)
int rbf_prep_norm_launch(const float* theta, const float* x, int64_t xrows, const float* y, int64_t yrows, float* w,
                         float* g2, float* na, float* nb, int S, int D, int64_t Dp, hipStream_t st, float* ys = nullptr,
                         float* xs = nullptr);

int chol_inv_bwd_first(const float* T, const float* gT, int nbatch, int n, void* ws, size_t ws_bytes,
                       const GemmParams* other, int oA, int oB, int onb, hipStream_t st);
int chol_inv_bwd_impl(const float* L, const float* T, const float* gL, const float* gT, float* gA, int nbatch, int n,
                      void* ws, size_t ws_bytes, bool gl_lower, hipStream_t st, bool first_done);

// Per-matrix tail of the first-task backward (t0_bwd_mat.h) sharing ONE launch with a plain NN product that does not depend
// on it (gemm.hip: t0_bwdmat_gemm_kernel).  Matrices [first, first + nmat): ids < S C are the K_uu role of (s, c) = id, ids
// >= S C the S_u role of class id - S C (which reads nothing the K_uu roles write: one launch for all of them).
struct BwdMatArgs {
  const float* QP;      // forward small columns [a | . | G | G2]: the KL part of ga and gG2 is g a and g G2 (seed_kl / S)
  const float *TT, *LL, *gQP, *RK, *KS, *seeds;     // TT / LL / KS: [S C + C][M][M]
  const float* gTT;
  float *gKS, *Wuu, *r_uu, *gtheta;
  float *g_u_mean, *gLu_part;                       // [C][M]: sum over s, accumulated with atomics (pre-zeroed); [S][C][M][M]: per-sample shares (lower triangles written)
  float* gL_acc;                                    // NULL, or [C][M][M] pre-zeroed: the K_uu roles add their tril(T_s^T gG_s) (atomics) and the S_u
                                                    // roles -- a LATER launch -- read the sums instead of walking the samples themselves
  int S, C, M, D, NR, LD;
};
// queue: 8 ints, zero when the launch starts (work queue of the product's tiles: gemm_persist_body), or NULL
int launch_bwdmat_gemm(const BwdMatArgs& a, int first, int nmat, const GemmParams& p, int nbatch, hipStream_t st,
                       const char* tag, int* queue);

// Launch replay (vargp_prof_remember / vargp_prof_replay): while remembering, tagged launch sites store a closure that
// repeats the launch.
bool prof_remembering();
void prof_remember(const char* tag, std::function<void(hipStream_t)> relaunch);

// Optional per-kernel timing with hipEvents on the launch stream (vargp_prof_* in the C ABI).
// Disabled (one branch) unless vargp_prof_enable(1); skipped while the stream is being captured.
struct ProfScope {
  ProfScope(const char* tag, hipStream_t st);
  ~ProfScope();
  int slot_;
  hipStream_t st_;
};

}  // namespace vargp
