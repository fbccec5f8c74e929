// Standalone kernel micro-benchmarks against libvargp_hip.so (no torch): used for tuning under rocprofv3.
//   ./bench_kernels [case] [iters]      case: all | kuf | gemm4k | small | chol | kufbwd | hot | hotS | tiles
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vargp_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
  float* d; CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
template <class F> static double time_us(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return 1e3 * ms / iters;
}
static void gemm(const float* A, const float* B, float* C, int M, int N, int K, int tA, int tB, int nb, long sA, long sB, long sC,
                 int triA = 0, int triB = 0) {
  vargp_gemm_desc d; memset(&d, 0, sizeof(d));
  d.M = M; d.N = N; d.K = K; d.transA = tA; d.transB = tB; d.A = A; d.B = B; d.C = C; d.D = nullptr;
  d.lda = tA ? M : K; d.ldb = tB ? K : N; d.ldc = N; d.nb[0] = nb; d.nb[1] = 1; d.nb[2] = 1;
  d.sA[0] = sA; d.sB[0] = sB; d.sC[0] = sC; d.alpha = 1.f; d.triA = triA; d.triB = triB;
  int rc = vargp_bgemm(&d, nullptr);
  if (rc) { printf("bgemm failed: %s\n", vargp_last_error()); exit(1); }
}

int main(int argc, char** argv) {
  std::string which = argc > 1 ? argv[1] : "all";
  int iters = argc > 2 ? atoi(argv[2]) : 50;
  auto want = [&](const char* n) { return which == "all" || which == n; };
  if (want("kuf")) {   // Cfg2 K_uf: S3 C10 M100 B512 D784
    const int S = 3, C = 10, M = 100, B = 512, D = 784;
    float* th = dev_rand((size_t)S * (D + 1), 0.05f, 1);
    float* z = dev_rand((size_t)C * M * D, 0.02f, 2);
    float* x = dev_rand((size_t)B * D, 0.02f, 3);
    float* K; CK(hipMalloc(&K, (size_t)S * C * M * B * 4));
    size_t wsb = vargp_rbf_workspace_bytes(S, C, M, B, D, 0);
    void* ws; CK(hipMalloc(&ws, wsb));
    double us = time_us([&] { vargp_rbf_gram_fwd(th, z, x, K, S, C, M, B, D, 1, ws, wsb, nullptr); }, iters);
    printf("kuf   rbf_gram_fwd(all 4 launches) %8.1f us  -> %.1f TFLOP/s on 2.408e9 flop\n", us, 2.408e9 / us * 1e-6);
  }
  if (which == "kufD") {   // K_uf with varying input dimension D: separates the per-launch fixed cost from the K loop
    const int S = 3, C = 10, M = 100, B = 512;
    for (int D : {16, 64, 128, 256, 512, 784, 1568}) {
      float* th = dev_rand((size_t)S * (D + 1), 0.05f, 1);
      float* z = dev_rand((size_t)C * M * D, 0.02f, 2);
      float* x = dev_rand((size_t)B * D, 0.02f, 3);
      float* K; CK(hipMalloc(&K, (size_t)S * C * M * B * 4));
      size_t wsb = vargp_rbf_workspace_bytes(S, C, M, B, D, 0);
      void* ws; CK(hipMalloc(&ws, wsb));
      vargp_prof_enable(1);
      double us = time_us([&] { vargp_rbf_gram_fwd(th, z, x, K, S, C, M, B, D, 1, ws, wsb, nullptr); }, iters);
      double ms; long long n; vargp_prof_read("rbf_kuf_gemm", &ms, (int64_t*)&n);
      vargp_prof_enable(0);
      printf("kufD D=%5d total %7.1f us   gemm launch (events) %7.1f us\n", D, us, 1e3 * ms / (double)n);
    }
  }
  if (want("kufplain")) {   // same shape as K_uf but plain NT GEMM (no scale, no exp epilogue)
    float* Z = dev_rand((size_t)1000 * 784, 1.f, 4);
    float* X = dev_rand((size_t)512 * 784, 1.f, 5);
    float* P; CK(hipMalloc(&P, (size_t)3 * 1000 * 512 * 4));
    double us = time_us([&] { gemm(Z, X, P, 1000, 512, 784, 0, 1, 3, 0, 0, 1000L * 512); }, iters);
    printf("kufplain [1000x784]x[512x784]^T b3 %8.1f us  -> %.1f TFLOP/s\n", us, 2.0 * 3 * 1000 * 784 * 512 / us * 1e-6);
  }
  if (want("kufbwd")) {   // P = W . Y  : [1000 x 512] x [512 x 784], batch 3
    float* W = dev_rand((size_t)3 * 1000 * 512, 1.f, 4);
    float* Y = dev_rand((size_t)512 * 784, 1.f, 5);
    float* P; CK(hipMalloc(&P, (size_t)3 * 1000 * 784 * 4));
    double us = time_us([&] { gemm(W, Y, P, 1000, 784, 512, 0, 0, 3, 1000L * 512, 0, 1000L * 784); }, iters);
    printf("kufbwd [1000x512]x[512x784] b3     %8.1f us  -> %.1f TFLOP/s\n", us, 2.0 * 3 * 1000 * 784 * 512 / us * 1e-6);
  }
  if (which == "hot") {   // the two big products of the Cfg2 step as plain GEMMs, every tile shape (timing experiments)
    float* Z = dev_rand((size_t)1000 * 784, 1.f, 4);
    float* X = dev_rand((size_t)512 * 784, 1.f, 5);
    float* W = dev_rand((size_t)3 * 1000 * 512, 1.f, 4);
    float* P; CK(hipMalloc(&P, (size_t)3 * 1000 * 784 * 4));
    for (int tile = 0; tile <= 3; ++tile) {
      vargp_tune_gemm_tile(tile);
      double a = time_us([&] { gemm(Z, X, P, 1000, 512, 784, 0, 1, 3, 0, 0, 1000L * 512); }, iters);
      double b = time_us([&] { gemm(W, X, P, 1000, 784, 512, 0, 0, 3, 1000L * 512, 0, 1000L * 784); }, iters);
      printf("hot tile %d   K_uf-shaped NT [1000x784]x[512x784]^T b3 %7.1f us (%.1f TF)   P_uf-shaped NN [1000x512]x[512x784] b3 %7.1f us (%.1f TF)\n",
             tile, a, 2.408e9 / a * 1e-6, b, 2.408e9 / b * 1e-6);
    }
    vargp_tune_gemm_tile(0);
  }
  if (which == "hotS") {   // the two big products of the first-task step at many hyper-samples (throughput-bound), every tile shape
    const int SM = 64;
    float* Z = dev_rand((size_t)1000 * 784, 1.f, 4);
    float* X = dev_rand((size_t)SM * 512 * 784, 1.f, 5);
    float* W = dev_rand((size_t)SM * 1000 * 512, 1.f, 4);
    float* P; CK(hipMalloc(&P, (size_t)SM * 1000 * 784 * 4));
    for (int S : {8, 16, 64}) {
      for (int tile = 0; tile <= 3; ++tile) {
        vargp_tune_gemm_tile(tile);
        double a = time_us([&] { gemm(Z, X, P, 1000, 512, 784, 0, 1, S, 0, 512L * 784, 1000L * 512); }, iters);
        double b = time_us([&] { gemm(W, X, P, 1000, 784, 512, 0, 0, S, 1000L * 512, 0, 1000L * 784); }, iters);
        const double fl = 2.0 * S * 1000 * 784 * 512;
        printf("hotS S=%2d tile %d   K_uf-shaped NT [1000x784]x[512x784]^T %8.1f us (%.1f TF = %.2f)   P_uf-shaped NN [1000x512]x[512x784] %8.1f us (%.1f TF = %.2f)\n",
               S, tile, a, fl / a * 1e-6, fl / a * 1e-6 / 157.3, b, fl / b * 1e-6, fl / b * 1e-6 / 157.3);
      }
    }
    vargp_tune_gemm_tile(0);
  }
  if (want("gemm4k")) {
    const int n = 4096;
    float* A = dev_rand((size_t)n * n, 1.f, 6);
    float* B = dev_rand((size_t)n * n, 1.f, 7);
    float* C; CK(hipMalloc(&C, (size_t)n * n * 4));
    for (int tb = 0; tb < 2; ++tb) {
      double us = time_us([&] { gemm(A, B, C, n, n, n, 0, tb, 1, 0, 0, 0); }, 10);
      printf("gemm4k NN/NT=%d                      %8.1f us  -> %.1f TFLOP/s\n", tb, us, 2.0 * n * n * (double)n / us * 1e-6);
    }
  }
  if (want("small")) {   // the (M x M) products of the ELBO: 100x100x100 / 100x512x100, batch 30
    float* A = dev_rand((size_t)30 * 100 * 100, 1.f, 8);
    float* B = dev_rand((size_t)30 * 100 * 512, 1.f, 9);
    float* C; CK(hipMalloc(&C, (size_t)30 * 100 * 512 * 4));
    for (int tA = 0; tA < 2; ++tA) for (int tB = 0; tB < 2; ++tB) {
      double us = time_us([&] { gemm(A, B, C, 100, 100, 100, tA, tB, 30, 10000, 10000, 10000); }, iters);
      printf("small 100x100x100 b30 tA%d tB%d       %8.1f us\n", tA, tB, us);
    }
    double us = time_us([&] { gemm(A, B, C, 100, 512, 100, 0, 0, 30, 10000, 51200, 51200, 1, 0); }, iters);
    printf("small T.Kuf 100x512x100 b30 (triA)   %8.1f us  -> %.1f TFLOP/s\n", us, 2.0 * 30 * 100 * 512 * 100 / us * 1e-6);
    us = time_us([&] { gemm(A, B, C, 100, 512, 100, 1, 0, 30, 10000, 51200, 51200, 2, 0); }, iters);
    printf("small G^T.P 100x512x100 b30 (triA up)%8.1f us\n", us);
  }
  if (which == "tiles") {   // tile-shape sweep on the mid-size batched products of the t > 0 program
    struct Case { const char* name; int M, N, K, tA, tB, nb, triA; };
    const Case cases[] = {
        {"P=T.K      200x512x200  b30 ", 200, 512, 200, 0, 0, 30, 1},  {"sq         200x200x200  b30 ", 200, 200, 200, 0, 0, 30, 0},
        {"P=T.K      400x512x400  b100", 400, 512, 400, 0, 0, 100, 1}, {"sq         400x400x400  b100", 400, 400, 400, 0, 0, 100, 0},
        {"gT NT      400x400x512  b100", 400, 400, 512, 0, 1, 100, 0}, {"V2=T^T.P   400x512x400  b100", 400, 512, 400, 1, 0, 100, 2},
        {"P=T.K      500x512x500  b30 ", 500, 512, 500, 0, 0, 30, 1},  {"P=T.K      600x512x600  b100", 600, 512, 600, 0, 0, 100, 1},
        {"P=T.K     1000x512x1000 b100", 1000, 512, 1000, 0, 0, 100, 1}, {"sq        1000x1000x1000 b100", 1000, 1000, 1000, 0, 0, 100, 0},
        {"W.z        400x784x400  b100", 400, 784, 400, 0, 0, 100, 0}, {"W.x       4000x784x512  b10 ", 4000, 784, 512, 0, 0, 10, 0},
        {"W.x       2000x784x512  b3  ", 2000, 784, 512, 0, 0, 3, 0},  {"W.z        200x784x200  b30 ", 200, 784, 200, 0, 0, 30, 0},
        {"blk        200x512x200  b200", 200, 512, 200, 0, 0, 200, 1}, {"blk NT     200x200x512  b200", 200, 200, 512, 0, 1, 200, 0},
        {"blk        100x512x100  b60 ", 100, 512, 100, 0, 0, 60, 1},
        {"s64 P=T.K  100x512x100  b640", 100, 512, 100, 0, 0, 640, 1}, {"s64 T^T.P  100x512x100  b640", 100, 512, 100, 1, 0, 640, 2},
        {"s64 NT     100x100x512  b640", 100, 100, 512, 0, 1, 640, 0}, {"s64 dense  100x512x100  b640", 100, 512, 100, 0, 0, 640, 0},
    };
    float* A = dev_rand((size_t)100 * 1000 * 1000, 1.f, 8);
    float* B = dev_rand((size_t)100 * 1000 * 1000, 1.f, 9);
    float* C; CK(hipMalloc(&C, (size_t)100 * 1000 * 1000 * 4));
    for (const Case& c : cases) {
      printf("%s", c.name);
      for (int tile = 1; tile <= 3; ++tile) {
        vargp_tune_gemm_tile(tile);
        const long sA = (long)c.M * c.K, sB = (long)c.K * c.N, sC = (long)c.M * c.N;
        double us = time_us([&] { gemm(A, B, C, c.M, c.N, c.K, c.tA, c.tB, c.nb, sA, sB, sC, c.triA, 0); }, iters);
        printf("   t%d %8.1f us", tile, us);
      }
      vargp_tune_gemm_tile(0);
      const long sA = (long)c.M * c.K, sB = (long)c.K * c.N, sC = (long)c.M * c.N;
      double us = time_us([&] { gemm(A, B, C, c.M, c.N, c.K, c.tA, c.tB, c.nb, sA, sB, sC, c.triA, 0); }, iters);
      printf("   auto %8.1f us\n", us);
    }
  }
  if (which == "stream") {   // what the memory system gives a kernel that reads 157 MB and writes 131 MB (the S = 64 P = T K_uf product)
    float* X; CK(hipMalloc(&X, (size_t)640 * 100 * 512 * 4));
    float* Y; CK(hipMalloc(&Y, (size_t)640 * 100 * 512 * 4));
    double us = time_us([&] { CK(hipMemcpyAsync(Y, X, (size_t)640 * 100 * 512 * 4, hipMemcpyDeviceToDevice, 0)); }, iters);
    printf("stream copy 131 MB -> 131 MB          %8.1f us  -> %.2f TB/s (read + write)\n", us, 2.0 * 640 * 100 * 512 * 4 / us * 1e-6);
  }
  if (which == "kufbig") {   // the stress-config K_uf tile: [10*2048 x 784] x [8192 x 784]^T, RBF epilogue vs plain NT product
    const int S = 1, C = 10, M = 2048, B = 8192, D = 784;
    float* th = dev_rand((size_t)S * (D + 1), 0.05f, 1);
    float* z = dev_rand((size_t)C * M * D, 0.02f, 2);
    float* x = dev_rand((size_t)B * D, 0.02f, 3);
    float* K; CK(hipMalloc(&K, (size_t)S * C * M * B * 4));
    size_t wsb = vargp_rbf_workspace_bytes(S, C, M, B, D, 0);
    void* ws; CK(hipMalloc(&ws, wsb));
    const double fl = 2.0 * C * M * (double)B * D;
    for (int tile = 0; tile <= 2; ++tile) {
      vargp_tune_gemm_tile(tile);
      double us = time_us([&] { vargp_rbf_gram_fwd(th, z, x, K, S, C, M, B, D, 1, ws, wsb, nullptr); }, iters);
      printf("kufbig rbf   tile %d  %8.1f us -> %.1f TFLOP/s\n", tile, us, fl / us * 1e-6);
      us = time_us([&] { gemm(z, x, K, C * M, B, D, 0, 1, 1, 0, 0, 0); }, iters);
      printf("kufbig plain tile %d  %8.1f us -> %.1f TFLOP/s\n", tile, us, fl / us * 1e-6);
    }
    vargp_tune_gemm_tile(0);
  }
  if (which == "chol2048") {   // the blocked factorisation + inverse at the stress size (10 matrices of 2048 x 2048)
    const int n = 2048, nb = 10;
    std::vector<float> h((size_t)nb * n * n);
    for (int b = 0; b < nb; ++b) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
      h[((size_t)b * n + i) * n + j] = (i == j ? 1.f : 0.f) + 0.5f * expf(-0.05f * (i - j) * (i - j));
    float *A, *L, *T; int* info; void* ws;
    const size_t wsb = vargp_chol_workspace_bytes(nb, n, 0);
    CK(hipMalloc(&A, h.size() * 4)); CK(hipMalloc(&L, h.size() * 4)); CK(hipMalloc(&T, h.size() * 4)); CK(hipMalloc(&info, nb * 4));
    CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double us = time_us([&] { vargp_chol_inv_fwd(A, 1e-4f, L, T, nullptr, info, nb, n, ws, wsb, nullptr); }, iters);
    const double fl = nb * (2.0 / 3.0) * (double)n * n * n;      // n^3/3 (L) + n^3/3 (T = L^-1), flops = 2 x MACs
    printf("chol2048 batch %d  L+T %8.1f us -> %.1f TFLOP/s\n", nb, us, fl / us * 1e-6);
    return 0;
  }
  if (want("chol")) {
    for (int n : {20, 40, 64, 100}) {
      const int nb = 30;
      std::vector<float> h((size_t)nb * n * n);
      for (int b = 0; b < nb; ++b) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
        h[((size_t)b * n + i) * n + j] = (i == j ? 1.f : 0.f) + 0.5f * expf(-0.05f * (i - j) * (i - j));
      float *A, *L, *T; int* info;
      CK(hipMalloc(&A, h.size() * 4)); CK(hipMalloc(&L, h.size() * 4)); CK(hipMalloc(&T, h.size() * 4)); CK(hipMalloc(&info, nb * 4));
      CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      double us = time_us([&] { vargp_chol_inv_fwd(A, 1e-4f, L, T, nullptr, info, nb, n, nullptr, 0, nullptr); }, iters);
      double us2 = time_us([&] { vargp_chol_inv_fwd(A, 1e-4f, L, nullptr, nullptr, info, nb, n, nullptr, 0, nullptr); }, iters);
      printf("chol  n=%3d batch %d  L+T %8.1f us   L only %8.1f us (incl. info memset)\n", n, nb, us, us2);
      if (void* f = dlsym(RTLD_DEFAULT, "vargp_debug_chol_stamps")) {   // only in -DVARGP_CHOL_STAMPS builds of the library
        unsigned long long st[8];
        vargp_chol_inv_fwd(A, 1e-4f, L, T, nullptr, info, nb, n, nullptr, 0, nullptr);
        hipDeviceSynchronize();
        reinterpret_cast<void (*)(unsigned long long*)>(f)(st);
        printf("   cycles per pivot (wave 0): loop %.0f  look-ahead %.0f  barrier %.0f  readlanes %.0f  update %.0f\n",
               st[0] / (double)n, st[1] / (double)n, st[2] / (double)n, st[3] / (double)n, st[4] / (double)n);
      }
    }
  }
  return 0;
}
