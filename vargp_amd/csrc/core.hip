// Error plumbing and small generic kernels (outer-sum reduction, fused Yogi step).
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <string>
#include <vector>
#include <string.h>

STEP_SPAN_TABLE(core)

namespace vargp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return VARGP_ELAUNCH;
  }
  return VARGP_OK;
}

__global__ void zero_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
void zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;
  if (n == 0) return;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n);
}

// ---- side stream (SideFork, common.h) ----------------------------------------------------------
namespace {
struct SideRes { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false, tried = false; std::mutex mu; };
SideRes g_side[64];
}
SideFork::SideFork(hipStream_t st, bool want) : st_(st), side_(st), forked_(false), dev_(-1) {
  if (!want) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  SideRes& r = g_side[dev];
  r.mu.lock();
  dev_ = dev;
  if (!r.tried) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (!capturing) {            // (resources are not created under a capture: that call runs its branch in line)
      r.tried = true;
      r.ok = hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&r.join, hipEventDisableTiming) == hipSuccess;
      if (!r.ok) (void)hipGetLastError();
    }
  }
  if (r.ok && hipEventRecord(r.fork, st) == hipSuccess && hipStreamWaitEvent(r.side, r.fork, 0) == hipSuccess) {
    side_ = r.side;
    forked_ = true;
  } else {
    (void)hipGetLastError();
  }
}
int SideFork::join() {
  if (!forked_) return VARGP_OK;
  forked_ = false;
  SideRes& r = g_side[dev_];
  const bool ok = hipEventRecord(r.join, side_) == hipSuccess && hipStreamWaitEvent(st_, r.join, 0) == hipSuccess;
  side_ = st_;
  if (!ok) { set_error("side stream: join failed: %s", hipGetErrorString(hipGetLastError())); return VARGP_ELAUNCH; }
  return VARGP_OK;
}
SideFork::~SideFork() {
  if (forked_) (void)join();
  if (dev_ >= 0) g_side[dev_].mu.unlock();
}

// ---- per-kernel event timing ---------------------------------------------------------------
struct ProfRec { std::string tag; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char* tag, hipStream_t st) : slot_(-1), st_(st) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  ProfRec r;
  r.tag = tag;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
  slot_ = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot_ >= 0) (void)hipEventRecord(g_prof[slot_].b, st_);
}

__global__ void sum_outer_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t outer, int64_t inner) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= inner) return;
  float acc = 0.f;
  for (int64_t r = 0; r < outer; ++r) acc += in[r * inner + i];
  out[i] = acc;
}

// Yogi (Zaheer et al. 2018): v <- v - (1-b2) sign(v - g^2) g^2 ; p <- p - lr/bias1 * m / (sqrt(v/bias2) + eps)
__global__ void yogi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bias1,
                            float bias2, const float* __restrict__ step) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (step) { const float t = step[0]; bias1 = 1.f - powf(b1, t); bias2 = 1.f - powf(b2, t); }
  const float gi = g[i], g2 = gi * gi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  float vi = v[i];
  const float df = vi - g2;
  const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
  vi = vi - (1.f - b2) * sg * g2;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bias2) + eps;
  p[i] -= (lr / bias1) * mi / denom;
}

struct YogiPack {
  float* p[8];
  const float* g[8];
  float* m[8];
  float* v[8];
  int64_t n[8];
  int blk_end[8];   // exclusive prefix of the blocks assigned to each tensor (kYogiPerBlock elements per block)
};
constexpr int kYogiPerBlock = 1024;   // 256 threads x 4 elements
struct YogiHyper { vargp_hyper_grad_desc h; int idx_mean, idx_logvar; };     // idx_mean < 0: off
// gradients of log_mean[d] and log_logvar[d]: the arithmetic of t0_hyper_bwd_kernel (elbo_shared.h).  tm / tv: the block's
// sums over (s, c) of the gamma^2 terms (only added at d = D)
__device__ __forceinline__ void yogi_hyper_grad(const vargp_hyper_grad_desc& h, int d, float tm, float tv, float& gm, float& gv) {
  const int D1 = h.D1, D = D1 - 1;
  const int dc = d < D1 ? d : D1 - 1;
  const float hs = h.map_est ? 0.f : 0.5f * expf(0.5f * h.log_logvar[dc]);
  gm = 0.f; gv = 0.f;
  for (int s0 = 0; s0 < h.S; s0 += 8) {          // eight samples per batch, loads first
    float gt[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = min(s0 + u, h.S - 1);
      gt[u] = h.gtheta[s * D1 + dc];
      ev[u] = h.map_est ? 0.f : h.eps_theta[s * D1 + dc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < h.S) {
        gm += gt[u];
        if (!h.map_est) gv = fmaf(gt[u] * hs, ev[u], gv);
      }
    }
  }
  if (d == D) { gm += tm; gv += tv; }
  if (!h.map_est) {
    const float g = h.seeds[0];
    gm += g * (h.log_mean[dc] - h.prior_log_mean[dc]) * expf(-h.prior_log_logvar[dc]);
    gv += g * 0.5f * (expf(h.log_logvar[dc] - h.prior_log_logvar[dc]) - 1.f);
  }
}
// step_mode 0: t = step[0].  1: t = step[0] + 1 (the caller advances the stored count some other way).
__global__ __launch_bounds__(256) void yogi_multi_kernel(YogiPack pk, int ntensors, float lr, float b1, float b2, float eps,
                                                         const float* __restrict__ step, int step_mode, const YogiHyper yh) {
  STEP_SPAN(core, 7);
  int t = 0;
  while (t + 1 < ntensors && (int)blockIdx.x >= pk.blk_end[t]) ++t;
  const int blk = (int)blockIdx.x - (t ? pk.blk_end[t - 1] : 0);
  const int64_t n = pk.n[t];
  float* __restrict__ p = pk.p[t];
  const float* __restrict__ g = pk.g[t];
  float* __restrict__ m = pk.m[t];
  float* __restrict__ v = pk.v[t];
  if (yh.idx_mean >= 0 && (t == yh.idx_mean || t == yh.idx_logvar)) {
    // the variational hyper-parameters: their gradient is finished here (deferred by the ELBO program's backward).
    // The whole block shares out the S C gamma^2 terms of the predictive variance (theta_D) when it holds d = D.
    __shared__ float red[4];
    const vargp_hyper_grad_desc& h = yh.h;
    const int D = h.D1 - 1;
    const float tt = step[0] + (step_mode ? 1.f : 0.f);
    const float bias1 = 1.f - powf(b1, tt), sb2 = sqrtf(1.f - powf(b2, tt));
    float tm = 0.f, tv = 0.f;
    if (blk == D / 256) {                     // (uniform)
      const float hsD = h.map_est ? 0.f : 0.5f * expf(0.5f * h.log_logvar[D]);
      float am = 0.f, av = 0.f;
      for (int e = threadIdx.x; e < h.S * h.C; e += 256) {
        const int s = e / h.C;
        const float tq = 2.f * h.g2[s] * h.gkd[e];
        am += tq;
        if (!h.map_est) av = fmaf(tq * hsD, h.eps_theta[s * h.D1 + D], av);
      }
      tm = block_sum<256>(am, red);
      __syncthreads();
      tv = block_sum<256>(av, red);
    }
    const bool lv = t == yh.idx_logvar;
    {     // ONE element per thread (these two tensors get a block per 256 elements): the gradient is a chain of dependent loads
      const int64_t i = (int64_t)blk * 256 + threadIdx.x;
      float gm, gv;
      yogi_hyper_grad(h, (int)min(i, n - 1), tm, tv, gm, gv);
      if (i < n) {
        const float gi = lv ? gv : gm;
        const_cast<float*>(g)[i] = gi;
        float pi = p[i], mi = m[i], vi = v[i];
        const float g2 = gi * gi;
        mi = b1 * mi + (1.f - b1) * gi;
        const float df = vi - g2;
        vi -= (1.f - b2) * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * g2;
        pi -= (lr / bias1) * mi / (sqrtf(vi) / sb2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
      }
    }
    return;
  }
  const float tt = step[0] + (step_mode ? 1.f : 0.f);
  const float bias1 = 1.f - powf(b1, tt), sb2 = sqrtf(1.f - powf(b2, tt));
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    const float g2 = gi * gi;
    mi = b1 * mi + (1.f - b1) * gi;
    const float df = vi - g2;
    vi -= (1.f - b2) * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * g2;
    pi -= (lr / bias1) * mi / (sqrtf(vi) / sb2 + eps);
  };
  const int64_t i0 = (int64_t)blk * kYogiPerBlock + 4 * threadIdx.x;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (aligned && i0 + 4 <= n) {           // 16 bytes per lane and array: one load / store instruction each
    float4 p4 = *reinterpret_cast<const float4*>(p + i0), m4 = *reinterpret_cast<const float4*>(m + i0);
    float4 v4 = *reinterpret_cast<const float4*>(v + i0);
    const float4 g4 = *reinterpret_cast<const float4*>(g + i0);
    upd(p4.x, g4.x, m4.x, v4.x); upd(p4.y, g4.y, m4.y, v4.y); upd(p4.z, g4.z, m4.z, v4.z); upd(p4.w, g4.w, m4.w, v4.w);
    *reinterpret_cast<float4*>(p + i0) = p4;
    *reinterpret_cast<float4*>(m + i0) = m4;
    *reinterpret_cast<float4*>(v + i0) = v4;
  } else {
    for (int64_t i = i0; i < i0 + 4 && i < n; ++i) {
      float pi = p[i], mi = m[i], vi = v[i];
      upd(pi, g[i], mi, vi);
      p[i] = pi; m[i] = mi; v[i] = vi;
    }
  }
}

}  // namespace vargp

using namespace vargp;

extern "C" int vargp_version(void) { return 110; }   // 110: native first-task program, launch replay, 4-GEMM Cholesky backward
extern "C" const char* vargp_last_error(void) { return g_err; }

extern "C" int vargp_sum_outer(const float* in, float* out, int64_t outer, int64_t inner, vargp_stream_t stream) {
  VARGP_REQUIRE(in && out && outer > 0 && inner > 0, "sum_outer: bad arguments");
  hipLaunchKernelGGL(sum_outer_kernel, dim3(cdiv(inner, 256)), dim3(256), 0, as_stream(stream), in, out, outer, inner);
  return check_launch("sum_outer");
}

extern "C" int vargp_yogi_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float bias1, float bias2, const float* step,
                               vargp_stream_t stream) {
  VARGP_REQUIRE(p && g && m && v && n > 0, "yogi_step: bad arguments");
  hipLaunchKernelGGL(yogi_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1,
                     beta2, eps, bias1, bias2, step);
  return check_launch("yogi_step");
}

static bool g_remember = false;
static std::vector<std::pair<std::string, std::function<void(hipStream_t)>>> g_replays;
bool vargp::prof_remembering() { return g_remember; }
void vargp::prof_remember(const char* tag, std::function<void(hipStream_t)> relaunch) {
  for (auto& r : g_replays)
    if (r.first == tag) { r.second = std::move(relaunch); return; }
  g_replays.emplace_back(tag, std::move(relaunch));
}
extern "C" int vargp_prof_remember(int on) {
  g_remember = on != 0;
  if (!g_remember) return VARGP_OK;
  g_replays.clear();
  return VARGP_OK;
}
extern "C" int vargp_prof_replay(const char* tag, int iters, double* avg_us, vargp_stream_t stream) {
  VARGP_REQUIRE(tag && iters > 0 && avg_us, "prof_replay: bad arguments");
  const std::function<void(hipStream_t)>* fn = nullptr;
  for (auto& r : g_replays)
    if (r.first == tag) fn = &r.second;
  VARGP_REQUIRE(fn, "prof_replay: no remembered launch tagged '%s'", tag);
  const bool was = g_remember;
  g_remember = false;                       // the replays themselves are not remembered
  hipStream_t st = as_stream(stream);
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return VARGP_ELAUNCH;
  for (int i = 0; i < 3; ++i) (*fn)(st);
  (void)hipEventRecord(a, st);
  for (int i = 0; i < iters; ++i) (*fn)(st);
  (void)hipEventRecord(b, st);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *avg_us = 1e3 * ms / iters;
  g_remember = was;
  return check_launch("prof_replay");
}

// step time line: one span table per translation unit (common.h: STEP_SPAN_TABLE)
extern "C" void vargp_debug_spans_gemm(unsigned long long* out, int mode);
extern "C" void vargp_debug_spans_t0(unsigned long long* out, int mode);
extern "C" int vargp_prof_spans(int mode, unsigned long long* out) {
  VARGP_REQUIRE(mode == 1 || mode == 2 || ((mode == 0 || mode == 3) && out),
                "prof_spans: mode 0 (read into out[12][2]) / 1 (clear + on) / 2 (off) / 3 (clock ticks into out[12][2])");
  void (*fns[3])(unsigned long long*, int) = {vargp_debug_spans_gemm, vargp_debug_spans_t0, vargp_debug_spans_core};
  if (mode == 3) {
    for (int i = 0; i < 24; ++i) out[i] = 0;
    for (auto fn : fns) {
      unsigned long long t[24];
      fn(t, 3);
      for (int i = 0; i < 12; ++i)
        if (t[2 * i + 1] != 0) { out[2 * i] = t[2 * i]; out[2 * i + 1] = t[2 * i + 1]; }
    }
    return check_launch("prof_spans");
  }
  if (mode) {
    for (auto fn : fns) fn(nullptr, mode);
    return check_launch("prof_spans");
  }
  for (int i = 0; i < 24; ++i) out[i] = 0;
  for (auto fn : fns) {
    unsigned long long t[24];
    fn(t, 0);
    for (int i = 0; i < 12; ++i)
      if (t[2 * i + 1] != 0) { out[2 * i] = t[2 * i]; out[2 * i + 1] = t[2 * i + 1]; }
  }
  return check_launch("prof_spans");
}

extern "C" int vargp_prof_enable(int on) {
  g_prof_on = on != 0;
  return VARGP_OK;
}
// Sum the elapsed time of every recorded launch whose tag equals `tag` (all tags if NULL/""), then
// drop those records.  Synchronises on the recorded events.
extern "C" int vargp_prof_read(const char* tag, double* total_ms, int64_t* launches) {
  double tot = 0.0;
  int64_t n = 0;
  std::vector<ProfRec> keep;
  for (auto& r : g_prof) {
    if (tag && tag[0] && r.tag != tag) { keep.push_back(r); continue; }
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++n; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.swap(keep);
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return VARGP_OK;
}

// Minibatch gather with the batch index read from device memory (include/vargp_hip.h: vargp_gather_minibatch): one wave per row
namespace vargp {
__global__ __launch_bounds__(256) void gather_minibatch_kernel(const float* __restrict__ data, const int64_t* __restrict__ targets,
                                                               const int64_t* __restrict__ perm, const float* __restrict__ step_now,
                                                               const float* __restrict__ step_base, int64_t n, int B, int D,
                                                               float* __restrict__ x, int64_t* __restrict__ y) {
  const int r = (int)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= B) return;
  const int64_t i = (int64_t)(step_now[0] - step_base[0]);
  int64_t pos = i * B + r;
  pos = pos < 0 ? 0 : (pos >= n ? n - 1 : pos);
  const int64_t src = perm[pos];
  const float* s = data + src * D;
  float* dst = x + (int64_t)r * D;
  if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
    for (int d = 4 * lane; d < D; d += 256) *reinterpret_cast<float4*>(dst + d) = *reinterpret_cast<const float4*>(s + d);
  } else {
    for (int d = lane; d < D; d += 64) dst[d] = s[d];
  }
  if (lane == 0) y[r] = targets[src];
}
}  // namespace vargp

extern "C" int vargp_gather_minibatch(const float* data, const int64_t* targets, const int64_t* perm, const float* step_now,
                                      const float* step_base, int64_t n, int B, int D, float* x, int64_t* y,
                                      vargp_stream_t stream) {
  VARGP_REQUIRE(data && targets && perm && step_now && step_base && x && y && n > 0 && B > 0 && D > 0, "gather_minibatch: bad arguments");
  hipLaunchKernelGGL(vargp::gather_minibatch_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), data, targets, perm, step_now,
                     step_base, n, B, D, x, y);
  return check_launch("gather_minibatch");
}

// up to 8 parameter tensors in ONE launch (the VAR-GP model has 5); step = device pointer to the step count t
extern "C" int vargp_yogi_step_multi(int ntensors, float* const* p, const float* const* g, float* const* m,
                                     float* const* v, const int64_t* n, float lr, float beta1, float beta2, float eps,
                                     const float* step, int step_mode, vargp_stream_t stream) {
  VARGP_REQUIRE(ntensors > 0 && ntensors <= 8 && p && g && m && v && n && step, "yogi_step_multi: bad arguments");
  VARGP_REQUIRE(step_mode == 0 || step_mode == 1, "yogi_step_multi: bad step_mode");
  YogiPack pk{};
  int nblk = 0;
  for (int i = 0; i < ntensors; ++i) {
    pk.p[i] = p[i]; pk.g[i] = g[i]; pk.m[i] = m[i]; pk.v[i] = v[i]; pk.n[i] = n[i];
    nblk += cdiv(n[i], kYogiPerBlock);
    pk.blk_end[i] = nblk;
  }
  if (nblk == 0) return VARGP_OK;
  YogiHyper yh{};
  yh.idx_mean = -1; yh.idx_logvar = -1;
  hipLaunchKernelGGL(yogi_multi_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), pk, ntensors, lr, beta1, beta2, eps,
                     step, step_mode, yh);
  return check_launch("yogi_step_multi");
}

extern "C" int vargp_yogi_step_multi_hyper(int ntensors, float* const* p, float* const* g, float* const* m, float* const* v,
                                           const int64_t* n, float lr, float beta1, float beta2, float eps, const float* step,
                                           int step_mode, const vargp_hyper_grad_desc* h, int idx_mean, int idx_logvar,
                                           vargp_stream_t stream) {
  VARGP_REQUIRE(ntensors > 0 && ntensors <= 8 && p && g && m && v && n && step && h, "yogi_step_multi_hyper: bad arguments");
  VARGP_REQUIRE(step_mode == 0 || step_mode == 1, "yogi_step_multi_hyper: bad step_mode");
  VARGP_REQUIRE(idx_mean >= 0 && idx_mean < ntensors && idx_logvar < ntensors && n[idx_mean] == h->D1 &&
                    (idx_logvar < 0 || n[idx_logvar] == h->D1) && (idx_logvar >= 0 || h->map_est),
                "yogi_step_multi_hyper: tensor indices / sizes do not match the hyper-parameter descriptor");
  VARGP_REQUIRE(h->S > 0 && h->C > 0, "yogi_step_multi_hyper: bad dims");
  VARGP_REQUIRE(h->log_mean && h->gtheta && h->g2 && h->gkd && h->seeds &&
                    (h->map_est || (h->log_logvar && h->prior_log_mean && h->prior_log_logvar && h->eps_theta)),
                "yogi_step_multi_hyper: null pointer in the hyper-parameter descriptor");
  YogiPack pk{};
  int nblk = 0;
  for (int i = 0; i < ntensors; ++i) {
    pk.p[i] = p[i]; pk.g[i] = g[i]; pk.m[i] = m[i]; pk.v[i] = v[i]; pk.n[i] = n[i];
    nblk += cdiv(n[i], (i == idx_mean || i == idx_logvar) ? 256 : kYogiPerBlock);
    pk.blk_end[i] = nblk;
  }
  if (nblk == 0) return VARGP_OK;
  YogiHyper yh{};
  yh.h = *h; yh.idx_mean = idx_mean; yh.idx_logvar = idx_logvar;
  hipLaunchKernelGGL(yogi_multi_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), pk, ntensors, lr, beta1, beta2, eps,
                     step, step_mode, yh);
  return check_launch("yogi_step_multi_hyper");
}
