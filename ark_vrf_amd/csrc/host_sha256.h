// host_sha256.h -- SHA-256 on the host: the batch verifiers' weight transcript of the `testing` suite
// (HashTranscript<Sha256>, src/suites/testing.rs; DigestXof: seed = H(absorbed), block_i = H(seed || LE64(i)), 32-byte blocks).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace avrf {

struct HostSha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len = 0;
  HostSha256() { static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}; memcpy(h, iv, sizeof iv); }
  static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t *b) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
        0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    memcpy(a, h, sizeof a);
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = a[7] + (ror(a[4], 6) ^ ror(a[4], 11) ^ ror(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K[i] + w[i];
      uint32_t t2 = (ror(a[0], 2) ^ ror(a[0], 13) ^ ror(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
      a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += a[i];
  }
  void update(const void *d, size_t n) { const uint8_t *p = (const uint8_t *)d; for (size_t i = 0; i < n; i++) { buf[len & 63] = p[i]; if ((++len & 63) == 0) block(buf); } }
  void final(uint8_t out[32]) const {
    HostSha256 c = *this; const uint64_t bits = len * 8; const uint8_t pad = 0x80, z = 0;
    c.update(&pad, 1); while ((c.len & 63) != 56) c.update(&z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    c.update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(c.h[i] >> 24); out[4 * i + 1] = (uint8_t)(c.h[i] >> 16); out[4 * i + 2] = (uint8_t)(c.h[i] >> 8); out[4 * i + 3] = (uint8_t)c.h[i]; }
  }
  // the first n bytes of the counter-mode squeeze stream of the absorbed data
  void squeeze_copy(uint8_t *out, size_t n) const {
    uint8_t seed[32]; final(seed);
    for (uint64_t ctr = 0; n; ctr++) {
      HostSha256 b; b.update(seed, 32); uint8_t c8[8]; for (int i = 0; i < 8; i++) c8[i] = (uint8_t)(ctr >> (8 * i)); b.update(c8, 8);
      uint8_t blk[32]; b.final(blk);
      const size_t k = n < 32 ? n : 32; memcpy(out, blk, k); out += k; n -= k;
    }
  }
};

}  // namespace avrf
