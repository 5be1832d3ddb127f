// sha256_dev.h -- per-lane SHA-256 and the reference's HashTranscript<Sha256> on the device (the crate's `testing` suite,
// src/suites/testing.rs; DigestXof: seed = H(absorbed), block_i = H(seed || LE64(i)) with 32-byte blocks,
// src/utils/transcript.rs:227-274).  Same transcript interface as sha512_dev.h / shake_dev.h (tr_* / rd_*).
#pragma once
#include "sha512_dev.h"

namespace avrf {

static __device__ __constant__ uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct Sha256 {
  uint32_t h[8];
  uint32_t w[16];   // current block, big-endian words
  uint32_t fill;    // bytes in the block
  uint32_t total;
};
struct Sha256H { uint32_t v[8]; };
struct Sha256W { uint32_t v[16]; };
AVRF_DI uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__device__ __noinline__ static Sha256H sha256_compress_nf(Sha256H hin, Sha256W win) {
  uint32_t a = hin.v[0], b = hin.v[1], c = hin.v[2], d = hin.v[3], e = hin.v[4], f = hin.v[5], g = hin.v[6], hh = hin.v[7];
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = win.v[i];
#pragma unroll 1
  for (int r = 0; r < 64; r += 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (r) {
        uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        uint32_t s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3), s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
        w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
      }
      uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25), ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + SHA256_K[r + i] + w[i];
      uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
  }
  Sha256H o;
  o.v[0] = hin.v[0] + a; o.v[1] = hin.v[1] + b; o.v[2] = hin.v[2] + c; o.v[3] = hin.v[3] + d;
  o.v[4] = hin.v[4] + e; o.v[5] = hin.v[5] + f; o.v[6] = hin.v[6] + g; o.v[7] = hin.v[7] + hh;
  return o;
}
AVRF_DI void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
  Sha256H hi; Sha256W wi;
#pragma unroll
  for (int i = 0; i < 8; i++) hi.v[i] = h[i];
#pragma unroll
  for (int i = 0; i < 16; i++) wi.v[i] = w[i];
  Sha256H o = sha256_compress_nf(hi, wi);
#pragma unroll
  for (int i = 0; i < 8; i++) h[i] = o.v[i];
}
AVRF_DI void tr_init(Sha256 &s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0; s.total = 0;
}
AVRF_DI void sha256_put(Sha256 &s, uint32_t pos, uint8_t b) {
  const uint32_t wi = pos >> 2, v = (uint32_t)b << (24 - 8 * (pos & 3));
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] |= (wi == (uint32_t)i) ? v : 0u;
}
AVRF_DI void sha256_flush(Sha256 &s) {
  sha256_compress(s.h, s.w);
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
}
AVRF_DI void tr_byte(Sha256 &s, uint8_t b) {
  sha256_put(s, s.fill, b);
  s.fill++; s.total++;
  if (s.fill == 64) sha256_flush(s);
}
// digest of a COPY (eight big-endian words)
AVRF_DI void sha256_final(Sha256 s, uint32_t (&out)[8]) {
  const uint32_t bits = s.total * 8;
  sha256_put(s, s.fill, 0x80);
  if (s.fill >= 56) sha256_flush(s);
  s.w[15] = bits;
  sha256_compress(s.h, s.w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = s.h[i];
}
// DigestXof reader: block i = H(seed || LE64(i)), 32 bytes: 16-byte chunk c sits in block c / 2
struct Sha256Reader { uint32_t seed[8]; uint32_t blk[8]; uint32_t have; };
AVRF_DI Sha256Reader tr_reader(const Sha256 &s) { Sha256Reader r; sha256_final(s, r.seed); r.have = 0; return r; }
AVRF_DI void rd_chunk16(Sha256Reader &r, uint32_t i, uint32_t (&w)[4]) {
  const uint32_t b = i >> 1;
  if (r.have != b + 1) {
    Sha256 s; tr_init(s);
#pragma unroll
    for (int k = 0; k < 8; k++) s.w[k] = r.seed[k];
    s.w[8] = __builtin_bswap32(b);          // LE64(counter): low word first, as big-endian message words
    s.w[9] = 0;
    s.w[10] = 0x80000000u;
    s.w[15] = 40 * 8;
    sha256_compress(s.h, s.w);
#pragma unroll
    for (int k = 0; k < 8; k++) r.blk[k] = s.h[k];
    r.have = b + 1;
  }
  // bytes [16 (i & 1), +16) of the digest as little-endian u32 words
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) v |= ((uint32_t)q == 4 * (i & 1) + (uint32_t)k) ? r.blk[q] : 0u;
    w[k] = __builtin_bswap32(v);
  }
}

}  // namespace avrf
