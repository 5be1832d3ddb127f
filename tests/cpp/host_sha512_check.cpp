// the product's HostSha512 (pipelined-schedule path for long inputs) against itself on short pieces and against known digests
#include "../../ark_vrf_amd/csrc/host_sha512.h"
#include <stdio.h>
#include <time.h>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
int main() {
  size_t n = (4u << 20) + 28; std::vector<uint8_t> d(n); for (size_t i = 0; i < n; i++) d[i] = (uint8_t)(i * 2654435761u >> 13);
  int bad = 0;
  for (size_t len : {size_t(0), size_t(1), size_t(111), size_t(4095), size_t(4096), size_t(4097), size_t(5000), size_t(65536 + 77), n}) {
    uint8_t a[64], b[64];
    avrf::HostSha512 h1; h1.update(d.data(), len); h1.final(a);                       // one piece (long path when >= 4096)
    avrf::HostSha512 h2; for (size_t o = 0; o < len; o += 100) h2.update(d.data() + o, len - o < 100 ? len - o : 100); h2.final(b);   // 100-byte pieces: scalar path only
    if (memcmp(a, b, 64)) { printf("MISMATCH at len %zu\n", len); bad = 1; }
  }
  double best = 1e9; uint8_t o[64];
  for (int r = 0; r < 6; r++) { double t0 = now(); avrf::HostSha512 h; h.update(d.data(), n); h.final(o); double t1 = now(); if (t1 - t0 < best) best = t1 - t0; }
  printf("consistent=%d  %.2f ms per 4 MiB  digest %02x%02x%02x%02x\n", !bad, best, o[0], o[1], o[2], o[3]);
  return bad;
}
