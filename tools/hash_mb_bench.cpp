// standalone benchmark: scalar chain vs x8 vs interleaved 2x8 (x16) multi-buffer SHA-512 over contiguous 4 MB messages
#include "../ark_vrf_amd/csrc/host_sha512.h"
#include "../ark_vrf_amd/csrc/host_sha512_mb.h"
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace avrf;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t len = 24 + 65536 * 64;
  const int K = 16;
  std::vector<std::vector<uint8_t>> msgs(K, std::vector<uint8_t>(len));
  for (int k = 0; k < K; k++) for (size_t i = 0; i < len; i++) msgs[k][i] = (uint8_t)(rand() >> 7);
  uint8_t ref[K][64];
  double t0 = now();
  for (int k = 0; k < K; k++) { HostSha512 h; h.update(msgs[k].data(), len); h.final(ref[k]); }
  double ts = (now() - t0) / K;
  printf("scalar: %.3f ms per message\n", ts * 1e3);
  WeightJob jobs[K]; WeightJob *pj[K];
  for (int k = 0; k < K; k++) { jobs[k].msg = msgs[k].data(); jobs[k].msg_len = len - (k % 3) * 64; pj[k] = &jobs[k]; }
  for (int k = 0; k < K; k++) { HostSha512 h; h.update(msgs[k].data(), jobs[k].msg_len); h.final(ref[k]); }
  for (int rep = 0; rep < 2; rep++) {
    t0 = now();
    sha512_weights_x8(pj, 8); sha512_weights_x8(pj + 8, 8);
    double t8 = (now() - t0) / 16;
    int bad = 0; for (int k = 0; k < K; k++) bad += memcmp(jobs[k].digest, ref[k], 64) != 0;
    printf("x8: %.3f ms per message (%.2fx scalar), bad %d\n", t8 * 1e3, ts / t8, bad);
  }
#ifdef HAVE_X16
  for (int rep = 0; rep < 2; rep++) {
    for (int k = 0; k < K; k++) memset(jobs[k].digest, 0, 64);
    t0 = now();
    sha512_weights_x16(pj, 16);
    double t16 = (now() - t0) / 16;
    int bad = 0; for (int k = 0; k < K; k++) bad += memcmp(jobs[k].digest, ref[k], 64) != 0;
    printf("x16: %.3f ms per message (%.2fx scalar), bad %d\n", t16 * 1e3, ts / t16, bad);
  }
  for (int cnt : {1, 3, 9, 13}) {
    for (int k = 0; k < K; k++) memset(jobs[k].digest, 0, 64);
    sha512_weights_x16(pj, cnt);
    int bad = 0; for (int k = 0; k < cnt; k++) bad += memcmp(jobs[k].digest, ref[k], 64) != 0;
    printf("x16 count %d: bad %d\n", cnt, bad);
  }
#endif
}
