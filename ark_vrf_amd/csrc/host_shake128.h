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
  static uint64_t rol(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
  // Keccak-f[1600], the 25 lanes in locals (a[x + 5 y]), one round = theta, rho + pi into b, chi: 350 ns against 625 ns of the table-driven
  // loop form on the build host -- a ring proof's transcript is ~30 permutations, on the prover and on the verifier
  static void keccakf(uint64_t a[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL,
      0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL,
      0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL,
      0x8000000080008008ULL};
    uint64_t a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a04 = a[4], a05 = a[5], a06 = a[6], a07 = a[7], a08 = a[8], a09 = a[9], a10 = a[10], a11 = a[11],
             a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18], a19 = a[19], a20 = a[20], a21 = a[21], a22 = a[22],
             a23 = a[23], a24 = a[24];
    for (int r = 0; r < 24; r++) {
      const uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22, c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23,
                     c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
      const uint64_t d0 = c4 ^ rol(c1, 1), d1 = c0 ^ rol(c2, 1), d2 = c1 ^ rol(c3, 1), d3 = c2 ^ rol(c4, 1), d4 = c3 ^ rol(c0, 1);
      // rho + pi: lane (x, y) rotated goes to (y, 2 x + 3 y)
      const uint64_t b00 = a00 ^ d0, b10 = rol(a01 ^ d1, 1), b20 = rol(a02 ^ d2, 62), b05 = rol(a03 ^ d3, 28), b15 = rol(a04 ^ d4, 27);
      const uint64_t b16 = rol(a05 ^ d0, 36), b01 = rol(a06 ^ d1, 44), b11 = rol(a07 ^ d2, 6), b21 = rol(a08 ^ d3, 55), b06 = rol(a09 ^ d4, 20);
      const uint64_t b07 = rol(a10 ^ d0, 3), b17 = rol(a11 ^ d1, 10), b02 = rol(a12 ^ d2, 43), b12 = rol(a13 ^ d3, 25), b22 = rol(a14 ^ d4, 39);
      const uint64_t b23 = rol(a15 ^ d0, 41), b08 = rol(a16 ^ d1, 45), b18 = rol(a17 ^ d2, 15), b03 = rol(a18 ^ d3, 21), b13 = rol(a19 ^ d4, 8);
      const uint64_t b14 = rol(a20 ^ d0, 18), b24 = rol(a21 ^ d1, 2), b09 = rol(a22 ^ d2, 61), b19 = rol(a23 ^ d3, 56), b04 = rol(a24 ^ d4, 14);
      a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
      a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
      a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
      a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
      a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
      a00 ^= RC[r];
    }
    a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04; a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09; a[10] = a10; a[11] = a11; a[12] = a12;
    a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19; a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
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
