// tests/test_host_logic.py: host_shake128.h (Keccak-f with the lanes in locals) against hashlib.shake_128, and the three-product
// reduction of a transcript challenge's 48 big-endian bytes (ring.hip fr_from_be48, restated here over host_te.h) against Python integers.
//   host_shake_check shake <hex message> <out bytes> <cut>   -> hex of the XOF output (message absorbed in pieces of `cut` bytes)
//   host_shake_check be48 <field 0|1> <96 hex digits>        -> hex (little-endian limbs, canonical) of int_BE(bytes) mod r
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../ark_vrf_amd/csrc/host_shake128.h"
#include "../../ark_vrf_amd/csrc/host_te.h"
using namespace avrf;
static std::vector<uint8_t> unhex(const char *s) {
  std::vector<uint8_t> v; size_t n = strlen(s);
  for (size_t i = 0; i + 1 < n; i += 2) { unsigned x; sscanf(s + i, "%2x", &x); v.push_back((uint8_t)x); }
  return v;
}
template <class F> static void be48(const uint8_t b[48]) {
  using Fr = HostField<F>;
  H256 hi = {{0, 0, 0, 0}}, lo;
  for (int i = 0; i < 2; i++) { uint64_t v; memcpy(&v, b + 8 * i, 8); hi.l[1 - i] = __builtin_bswap64(v); }
  for (int i = 0; i < 4; i++) { uint64_t v; memcpy(&v, b + 16 + 8 * i, 8); lo.l[3 - i] = __builtin_bswap64(v); }
  const H256 r2 = Fr::r2();
  const H256 m = Fr::add(Fr::mul(Fr::mul(hi, r2), r2), Fr::mul(lo, r2)), c = Fr::from_mont(m);
  for (int i = 3; i >= 0; i--) printf("%016llx", (unsigned long long)c.l[i]);
  printf("\n");
}
int main(int argc, char **argv) {
  if (argc >= 5 && !strcmp(argv[1], "shake")) {
    std::vector<uint8_t> m = unhex(argv[2]); size_t n = (size_t)atol(argv[3]), cut = (size_t)atol(argv[4]);
    HostShake128 h;
    for (size_t o = 0; o < m.size(); o += cut) h.update(m.data() + o, m.size() - o < cut ? m.size() - o : cut);
    std::vector<uint8_t> out(n); h.squeeze_copy(out.data(), n);
    for (uint8_t x : out) printf("%02x", x);
    printf("\n");
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "be48")) {
    std::vector<uint8_t> b = unhex(argv[3]); if (b.size() != 48) return 2;
    if (atoi(argv[2]) == 0) be48<FqBandersnatch>(b.data()); else be48<FqBabyJubJub>(b.data());
    return 0;
  }
  return 2;
}
