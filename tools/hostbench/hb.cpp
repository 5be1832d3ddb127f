#define AVRF_SHA_MB_IMPL
#include "../../ark_vrf_amd/csrc/host_sha512_mb.h"
#include "../../ark_vrf_amd/csrc/host_sha512.h"
#include <chrono>
#include <cstdio>
#include <vector>
#include <random>
using namespace avrf;
int main() {
  const size_t n = 65536, len = 27 + n * 64;
  std::vector<std::vector<uint8_t>> msgs(16, std::vector<uint8_t>(len));
  std::mt19937_64 rng(1);
  for (auto &m : msgs) for (auto &b : m) b = (uint8_t)rng();
  WeightJob jobs[16]; WeightJob *pj[16];
  for (int i = 0; i < 16; i++) { jobs[i] = WeightJob{}; jobs[i].msg = msgs[i].data(); jobs[i].msg_len = len; pj[i] = &jobs[i]; }
  auto t = [&](const char *name, auto f, int k) {
    f(); auto t0 = std::chrono::steady_clock::now(); int reps = 5; for (int r = 0; r < reps; r++) f();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-28s %8.3f ms per call, %7.3f ms per batch  (digest %02x%02x)\n", name, ms, ms / k, jobs[0].digest[0], jobs[k-1].digest[5]);
  };
  t("x8 (scalar transpose)", [&] { sha512_weights_x8_impl(pj, 8); }, 8);
  t("x16 path, 8 jobs (x8r)", [&] { sha512_weights_x16_impl(pj, 8); }, 8);
  t("x16 path, 16 jobs", [&] { sha512_weights_x16_impl(pj, 16); }, 16);
  uint8_t dg[64];
  t("scalar chain (HostSha512)", [&] { HostSha512 h; h.update(msgs[0].data(), len); h.final(dg); jobs[0].digest[0] = dg[0]; }, 1);
  return 0;
}
