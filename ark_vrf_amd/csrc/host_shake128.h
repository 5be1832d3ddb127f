// host_shake128.h -- SHAKE128 on the host: the ark-transcript of the ring proof (ring.hip) and the batch verifiers'
// weight transcript of the SHAKE128 suite (capi.hip; XofTranscript<HostShake128>, src/utils/transcript.rs:292-293).
#pragma once
#include <stdint.h>
#include <string.h>
#include <stddef.h>

namespace avrf {

struct HostShake128 {
  uint64_t s[25]; uint8_t buf[168]; size_t fill = 0;
  HostShake128() { memset(s, 0, sizeof s); }
  static void keccakf(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL,
      0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL,
      0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL,
      0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
      uint64_t bc[5];
      for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
      for (int i = 0; i < 5; i++) { uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63)); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
      uint64_t t = st[1];
      for (int i = 0; i < 24; i++) { int j = PIL[i]; uint64_t b = st[j]; st[j] = (t << ROT[i]) | (t >> (64 - ROT[i])); t = b; }
      for (int j = 0; j < 25; j += 5) { for (int i = 0; i < 5; i++) bc[i] = st[j + i]; for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5]; }
      st[0] ^= RC[r];
    }
  }
  void absorb_block() { for (int i = 0; i < 21; i++) { uint64_t v; memcpy(&v, buf + 8 * i, 8); s[i] ^= v; } keccakf(s); fill = 0; }
  void update(const void *d, size_t n) { const uint8_t *p = (const uint8_t *)d; while (n) { size_t k = 168 - fill; if (k > n) k = n; memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 168) absorb_block(); } }
  // squeeze the first `n` bytes of the XOF output of a COPY of the state (the transcript continues)
  void squeeze_copy(uint8_t *out, size_t n) const {
    HostShake128 c = *this;
    memset(c.buf + c.fill, 0, 168 - c.fill); c.buf[c.fill] ^= 0x1f; c.buf[167] ^= 0x80; c.absorb_block();
    while (n) { size_t k = n < 168 ? n : 168; memcpy(out, c.s, k); out += k; n -= k; if (n) keccakf(c.s); }
  }
};

}  // namespace avrf
