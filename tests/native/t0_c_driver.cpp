// The C ABI without Python or torch: reads a first-task problem from a raw file, runs vargp_elbo_t0_fwd / _bwd with the
// library's native noise on plain hipMalloc'd buffers and writes (kl_hypers, kl_u, nll) and the five gradients back.
// tests/test_hip_cabi_driver.py compares the output with the Python route on the same seed.
//   file format (little endian): int32 S C M D B F, uint64 seed, then float32 log_mean[D+1] log_logvar[D+1]
//   prior_log_mean[D+1] prior_log_logvar[D+1] z[C*M*D] u_mean[C*M] u_tril_vec[C*M(M+1)/2] x[B*D], int64 y[B]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/vargp_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <typename T> static T* to_dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("problem"); return 1; }
  int32_t dims[6];
  uint64_t seed;
  if (fread(dims, 4, 6, f) != 6 || fread(&seed, 8, 1, f) != 1) return 1;
  const int S = dims[0], C = dims[1], M = dims[2], D = dims[3], B = dims[4], F = dims[5], D1 = D + 1;
  auto rd = [&](size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) exit(1); return v; };
  auto log_mean = rd(D1), log_logvar = rd(D1), pm = rd(D1), pv = rd(D1);
  auto z = rd((size_t)C * M * D), um = rd((size_t)C * M), uv = rd((size_t)C * M * (M + 1) / 2), x = rd((size_t)B * D);
  std::vector<int64_t> y(B);
  if (fread(y.data(), 8, B, f) != (size_t)B) return 1;
  fclose(f);

  vargp_elbo_t0_desc d{};
  d.S = S; d.C = C; d.M = M; d.D = D; d.B = B; d.F = F; d.jitter = 1e-4f;
  d.log_mean = to_dev(log_mean); d.log_logvar = to_dev(log_logvar); d.prior_log_mean = to_dev(pm); d.prior_log_logvar = to_dev(pv);
  d.z = to_dev(z); d.u_mean = to_dev(um); d.u_tril_vec = to_dev(uv); d.x = to_dev(x); d.y = to_dev(y);
  d.eps_theta = nullptr; d.eps_f = nullptr;                 // native noise
  d.rng_seed = seed; d.rng_sample_offset = 0;
  uint32_t* counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
  d.rng_counter = counter;
  float* scalars; CK(hipMalloc(&scalars, 12));
  int32_t* info; CK(hipMalloc(&info, 4 * (S * C + C)));
  d.scalars = scalars; d.info = info;
  d.ws_bytes = vargp_elbo_t0_workspace_bytes(S, C, M, D, B, F);
  CK(hipMalloc(&d.ws, d.ws_bytes));
  const float seeds_h[3] = {2.0f, 1.0f, 7.0f};                // d total / d (kl_hypers, kl_u, nll)
  float* seeds; CK(hipMalloc(&seeds, 12)); CK(hipMemcpy(seeds, seeds_h, 12, hipMemcpyHostToDevice));
  const size_t ng[5] = {(size_t)D1, (size_t)D1, z.size(), um.size(), uv.size()};
  float* g[5];
  for (int i = 0; i < 5; ++i) CK(hipMalloc(&g[i], ng[i] * 4));

  // early hand-over of the Cholesky status: pinned host words + an event recorded right behind the factorisation launch
  int32_t* info_host; CK(hipHostMalloc(&info_host, 4 * (S * C + C)));
  for (int i = 0; i < S * C + C; ++i) info_host[i] = -7;
  hipEvent_t info_ev; CK(hipEventCreateWithFlags(&info_ev, hipEventDisableTiming));
  d.info_host = info_host; d.info_event = info_ev;
  if (vargp_elbo_t0_fwd(&d, nullptr) != VARGP_OK) { fprintf(stderr, "fwd: %s\n", vargp_last_error()); return 3; }
  CK(hipEventSynchronize(info_ev));                           // (the forward's remaining launches may still be running)
  int32_t early_bad = 0; for (int i = 0; i < S * C + C; ++i) early_bad += info_host[i] != 0;
  if (vargp_elbo_t0_bwd(&d, seeds, g[0], g[1], g[2], g[3], g[4], nullptr) != VARGP_OK) { fprintf(stderr, "bwd: %s\n", vargp_last_error()); return 3; }
  CK(hipDeviceSynchronize());

  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror("result"); return 1; }
  float sc[3]; CK(hipMemcpy(sc, scalars, 12, hipMemcpyDeviceToHost));
  std::vector<int32_t> ih(S * C + C); CK(hipMemcpy(ih.data(), info, 4 * ih.size(), hipMemcpyDeviceToHost));
  int32_t bad = 0; for (int32_t v : ih) bad += v != 0;
  for (size_t i = 0; i < ih.size(); ++i)
    if (ih[i] != info_host[i]) { fprintf(stderr, "info_host[%zu] = %d, info = %d\n", i, info_host[i], ih[i]); return 4; }
  if (early_bad != bad) { fprintf(stderr, "early status %d != %d\n", early_bad, bad); return 4; }
  fwrite(sc, 4, 3, o); fwrite(&bad, 4, 1, o);
  for (int i = 0; i < 5; ++i) {
    std::vector<float> h(ng[i]); CK(hipMemcpy(h.data(), g[i], ng[i] * 4, hipMemcpyDeviceToHost));
    fwrite(h.data(), 4, ng[i], o);
  }
  fclose(o);
  printf("kl_hypers %.6f kl_u %.6f nll %.6f (failed factorisations: %d)\n", sc[0], sc[1], sc[2], bad);
  return 0;
}
