// Host-side AddressSanitizer run of libvargp_hip's launcher code (argument checks, workspace carving, descriptor handling):
// everything that executes BEFORE a kernel launch, driven without a GPU.  Built by `make -C tests/native asan` with
// -fsanitize=address on the host side only (-fno-gpu-sanitize: device ASan needs xnack, which this pool does not offer) and run
// by tests/test_host_asan.py on CPU.  Every call below must return an error code (bad arguments) or a size; none may launch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vargp_hip.h"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

int main() {
  EXPECT(vargp_version() >= 100);
  // workspace queries over a grid of shapes (incl. degenerate ones): sizes must be positive and monotone in the batch
  const int dims[] = {1, 3, 20, 64, 65, 100, 101, 200, 1000, 2048};
  for (int n : dims) {
    for (int nb : {1, 7, 100}) {
      const size_t f = vargp_chol_workspace_bytes(nb, n, 0), b = vargp_chol_workspace_bytes(nb, n, 1);
      EXPECT(f > 0 && b > 0 && b >= (size_t)2 * nb * n * n * 4);
    }
  }
  for (int M : {1, 12, 100, 200}) for (int B : {1, 37, 512}) for (int D : {1, 2, 33, 784}) {
    EXPECT(vargp_rbf_workspace_bytes(3, 10, M, B, D, 0) > 0);
    EXPECT(vargp_rbf_workspace_bytes(3, 10, M, B, D, 1) > vargp_rbf_workspace_bytes(1, 1, M, B, D, 1) / 2);
    const size_t t0 = vargp_elbo_t0_workspace_bytes(3, 10, M, D, B, 10);
    EXPECT(t0 > 0 && vargp_elbo_t0_workspace_bytes(6, 10, M, D, B, 10) > t0);
    for (int nblk : {1, 2, 5}) EXPECT(vargp_elbo_tn_workspace_bytes(3, 10, M, D, B, 10, nblk) > 0);
  }
  // descriptors with missing pieces: EINVAL before anything is touched
  vargp_gemm_desc g; memset(&g, 0, sizeof(g));
  EXPECT(vargp_bgemm(nullptr, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_bgemm(&g, nullptr) == VARGP_EINVAL);                       // null operands
  g.M = -1; EXPECT(vargp_bgemm(&g, nullptr) == VARGP_EINVAL);
  EXPECT(strlen(vargp_last_error()) > 0);
  vargp_elbo_t0_desc t; memset(&t, 0, sizeof(t));
  EXPECT(vargp_elbo_t0_fwd(nullptr, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_elbo_t0_fwd(&t, nullptr) == VARGP_EINVAL);                // zero dims
  t.S = 3; t.C = 10; t.M = 100; t.D = 784; t.B = 512; t.F = 10;
  EXPECT(vargp_elbo_t0_fwd(&t, nullptr) == VARGP_EINVAL);                // null pointers
  float seeds_dummy = 0.f;
  EXPECT(vargp_elbo_t0_bwd(&t, &seeds_dummy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == VARGP_EINVAL);
  vargp_elbo_tn_desc n; memset(&n, 0, sizeof(n));
  EXPECT(vargp_elbo_tn_fwd(nullptr, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_elbo_tn_fwd(&n, nullptr) == VARGP_EINVAL);
  n.S = 2; n.C = 3; n.M = 20; n.D = 8; n.B = 16; n.F = 2; n.nblk = 2;
  EXPECT(vargp_elbo_tn_fwd(&n, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_elbo_tn_begin(&n, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_elbo_tn_tile(&n, nullptr, nullptr, nullptr, nullptr, 4, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_elbo_tn_end(&n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == VARGP_EINVAL);
  // a descriptor that is complete except for a workspace that is too small: rejected by the size check (host memory stands in
  // for the device pointers: nothing is dereferenced before the check fails)
  std::vector<float> buf(4096, 0.f);
  std::vector<int64_t> yb(16, 0);
  std::vector<int32_t> info(64, 0);
  n.log_mean = n.log_logvar = n.prior_log_mean = n.prior_log_logvar = buf.data();
  n.z = n.u_mean = n.u_tril_vec = n.x = n.eps_theta = n.eps_f = buf.data();
  n.z_all = n.rk_all = n.scalars = buf.data();
  n.y = yb.data(); n.info = info.data(); n.ws = buf.data(); n.ws_bytes = 64;
  EXPECT(vargp_elbo_tn_fwd(&n, nullptr) == VARGP_EINVAL);
  EXPECT(strstr(vargp_last_error(), "workspace") != nullptr);
  // small entry points with null arguments
  EXPECT(vargp_chol_inv_fwd(nullptr, 1e-4f, nullptr, nullptr, nullptr, nullptr, 1, 8, nullptr, 0, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_chol_inv_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 8, nullptr, 0, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_trsm_lower_fwd(nullptr, nullptr, nullptr, 1, 8, 4, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_rbf_gram_fwd(nullptr, nullptr, nullptr, nullptr, 1, 1, 1, 1, 1, 0, nullptr, 0, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_bias_act_fwd(nullptr, nullptr, nullptr, 4, 4, 1, nullptr) == VARGP_EINVAL);
  EXPECT(vargp_prof_enable(0) == VARGP_OK && vargp_prof_remember(0) == VARGP_OK && vargp_tune_gemm_tile(0) == VARGP_OK);
  printf(fails ? "asan_host: %d FAILED\n" : "asan_host: ok\n", fails);
  return fails ? 1 : 0;
}
