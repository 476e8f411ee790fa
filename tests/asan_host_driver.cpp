// Host-side sanitizer job: exercises the HOST logic of the C-ABI -- plan
// building, arena carving, descriptor validation, argument checks, the MT19937 seed vector, the
// health record -- in a build of the library whose host code is instrumented with
// -fsanitize=address,undefined (make -C precondition_amd/csrc asan-host).  No GPU is needed:
// the workspace queries are pure host code, and the compute entry points are called with
// invalid arguments / without a device to walk their rejection paths.  Any sanitizer report
// aborts the process (halt_on_error), so exit code 0 means a clean run.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/ps_api.h"

#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) { fprintf(stderr, "FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
  } while (0)

int main() {
  CHECK(ps_version() == PS_VERSION);
  for (int c = -8; c <= 2; ++c) CHECK(ps_error_string(c) != nullptr);

  // v0 of the power iteration (MT19937): prefix property and range
  std::vector<float> v0(2048), v1(100);
  CHECK(ps_power_iteration_v0(2048, v0.data()) == PS_OK);
  CHECK(ps_power_iteration_v0(100, v1.data()) == PS_OK);
  CHECK(memcmp(v0.data(), v1.data(), 100 * sizeof(float)) == 0);
  CHECK(ps_power_iteration_v0(-1, v0.data()) == PS_EINVAL);
  for (float x : v0) CHECK(x >= -1.f && x < 1.f);

  // ViT-B/16 census (BASELINE.json configs[3]): 395 statistics
  std::vector<int32_t> n, p, pad;
  auto add = [&](int count, int size, int exp) {
    for (int i = 0; i < count; ++i) { n.push_back(size); p.push_back(exp); pad.push_back(size); }
  };
  add(172, 768, 4); add(112, 768, 2); add(72, 1024, 4); add(36, 1024, 2);
  add(1, 1000, 4); add(1, 1000, 2); add(1, 197, 4);
  CHECK(n.size() == 395);
  const size_t ws_vit = ps_newton_root_workspace_bytes(395, n.data(), p.data(), pad.data());
  CHECK(ws_vit > (size_t)8 << 30 && ws_vit < (size_t)24 << 30);
  // padding_start below n shrinks the effective size; nullptr = no padding
  std::vector<int32_t> pad2(pad);
  for (auto& x : pad2) x /= 2;
  CHECK(ps_newton_root_workspace_bytes(395, n.data(), p.data(), pad2.data()) < ws_vit);
  CHECK(ps_newton_root_workspace_bytes(395, n.data(), p.data(), nullptr) == ws_vit);
  CHECK(ps_newton_root_workspace_bytes(0, n.data(), p.data(), nullptr) == 0);
  // ragged / tiny / large sizes, exponents 1..8
  std::vector<int32_t> rn, rp;
  for (int i = 1; i <= 64; ++i) { rn.push_back(1 + (i * 37) % 700); rp.push_back(1 + i % 8); }
  CHECK(ps_newton_root_workspace_bytes(64, rn.data(), rp.data(), nullptr) > 0);
  CHECK(ps_power_iteration_workspace_bytes(64, rn.data()) > 0);
  CHECK(ps_power_iteration_workspace_bytes(395, n.data()) > 0);

  // eigh: cfg3 (64 x 2048), mixed small / big blocks, odd sizes
  std::vector<int32_t> en(64, 2048);
  const size_t ws_e = ps_eigh_root_workspace_bytes(64, en.data());
  CHECK(ws_e > (size_t)5 << 30 && ws_e < (size_t)8 << 30);
  std::vector<int32_t> em = {64, 96, 128, 129, 200, 256, 300, 1000, 1, 2048};
  CHECK(ps_eigh_root_workspace_bytes((int)em.size(), em.data()) > 0);
  CHECK(ps_eigh_root_workspace_bytes(0, em.data()) == 0);

  // grouped descriptors (host arrays of device pointers: never dereferenced by the queries)
  float* fake = reinterpret_cast<float*>(0x10000);
  std::vector<ps_stats_desc> sd;
  for (int i = 0; i < 395; ++i) {
    ps_stats_desc d;
    memset(&d, 0, sizeof(d));
    d.g = fake; d.layout = i & 1; d.d = n[i]; d.k = (i % 3 == 0) ? 1 : 768; d.nseg = 1;
    d.ld = 1024; d.seg_stride = 0; d.stat_in = fake; d.stat_out = fake; d.lds = n[i];
    sd.push_back(d);
  }
  (void)ps_stats_update_grouped_workspace_bytes(sd.data(), (int)sd.size());
  (void)ps_stats_update_grouped_workspace_bytes(sd.data(), 0);
  std::vector<ps_gemm_desc> gd;
  for (int i = 0; i < 200; ++i) {
    ps_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.a = fake; d.b = fake; d.c = fake;
    d.m = (i % 5 == 0) ? 1 : 768; d.n = 128 + 64 * (i % 7); d.k = (i % 4 == 0) ? 4096 : 768;
    d.transa = i & 1; d.transb = (i >> 1) & 1; d.lda = 4096; d.ldb = 4096; d.ldc = 4096;
    gd.push_back(d);
  }
  (void)ps_gemm_grouped_workspace_bytes(gd.data(), (int)gd.size());
  (void)ps_gemm_grouped_workspace_bytes(gd.data(), 1);   // a single tall-skinny product: split-K
  std::vector<ps_transform_desc> td(200);
  memset(td.data(), 0, sizeof(ps_transform_desc) * td.size());
  for (size_t i = 0; i < td.size(); ++i) { td[i].grad = fake; td[i].upd_out = fake; td[i].numel = 1 + 1000 * i; }
  (void)ps_transform_grads_workspace_bytes(td.data(), (int)td.size());
  std::vector<ps_quant_desc> qd(395);
  memset(qd.data(), 0, sizeof(ps_quant_desc) * qd.size());
  for (size_t i = 0; i < qd.size(); ++i) {
    qd[i].fvalue = fake; qd[i].codes = fake; qd[i].diagonal = fake; qd[i].bucket_size = fake;
    qd[i].rows = n[i]; qd[i].cols = n[i]; qd[i].ld = n[i]; qd[i].ldq = n[i];
    qd[i].bits = (i & 1) ? 8 : 16; qd[i].extract_diagonal = 1;
  }
  (void)ps_quantize_workspace_bytes(qd.data(), (int)qd.size());
  (void)ps_dequantize_workspace_bytes(qd.data(), (int)qd.size());
  CHECK(ps_mat_power_workspace_bytes(512, 6) > 0);

  // rejection paths of the compute entry points (no device here: PS_EDEVICE / a HIP error, or
  // PS_EINVAL where the argument check comes first) -- they must return, not crash
  std::vector<const float*> aptr(4, fake);
  std::vector<float*> optr(4, fake);
  std::vector<int32_t> four_n = {128, 128, 128, 128}, four_p = {4, 4, 4, 4};
  float metrics[4 * PS_METRICS_STRIDE];
  int rc = ps_newton_root_batched_f32(nullptr, aptr.data(), four_n.data(), four_n.data(),
                                      four_p.data(), nullptr, 4, 100, 1e-6f, 1e-6f, 1,
                                      PS_SYMMETRY_VERIFY, optr.data(), four_n.data(), metrics,
                                      nullptr, 0, nullptr);
  CHECK(rc != PS_OK);
  rc = ps_eigh_root_batched_f32(nullptr, aptr.data(), four_n.data(), four_n.data(), four_p.data(),
                                nullptr, 4, 1e-6f, 1e-6f, 1, optr.data(), four_n.data(), metrics,
                                nullptr, 0);
  CHECK(rc != PS_OK);
  rc = ps_stats_update_grouped_f32(nullptr, sd.data(), 4, 0.9f, 0.1f, nullptr, 0);
  CHECK(rc != PS_OK);
  CHECK(ps_comm_allgather(nullptr, nullptr, fake, fake, 16) != PS_OK);
  (void)ps_comm_last_error();

  // health record of the resident power iteration
  unsigned expired = 99; int coll = 99, res = 99;
  CHECK(ps_collective_in_flight(1) == 1);
  CHECK(ps_power_iteration_health(&expired, &coll, &res) == PS_OK);
  CHECK(coll == 1 && res == 0);
  CHECK(ps_collective_in_flight(-1) == 0);
  CHECK(ps_power_iteration_reset_health() == PS_OK);
  CHECK(ps_newton_averaged_steps() >= 0);
  printf("asan host driver: ok (ViT-B Newton workspace %.2f GiB, cfg3 eigh workspace %.2f GiB)\n",
         ws_vit / 1073741824.0, ws_e / 1073741824.0);
  return 0;
}
