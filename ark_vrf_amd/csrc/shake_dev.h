// shake_dev.h -- per-lane SHAKE128 and the reference's XofTranscript<Shake128> on the device
// (src/utils/transcript.rs:103-195,292-293: absorb = sponge absorb, squeeze = the XOF reader; suite
// Bandersnatch-SHAKE128-ELL2, src/suites/bandersnatch_shake128.rs).  The 25 lanes of the state live in VGPRs; bytes are
// XORed in with shifts, the lane picked by a select chain so that the state array is only indexed by constants.
// The transcript interface (tr_* / rd_*) is shared with sha512_dev.h; kernels are generic over it (proto_dev.h).
#pragma once
#include "sha512_dev.h"

namespace avrf {

static __device__ __constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

struct KState { uint64_t a[25]; };
AVRF_DI uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600]; out of line like the SHA-512 compression (reached from every absorb / squeeze site)
__device__ __noinline__ static KState keccak_f_nf(KState s) {
  uint64_t *a = s.a;
#pragma unroll 1
  for (int r = 0; r < 24; r++) {
    uint64_t c[5], d[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    // rho + pi (rotation offsets of lane x + 5 y), then chi
    uint64_t b[25];
    b[0] = a[0];
    b[10] = rol64(a[1], 1);   b[20] = rol64(a[2], 62);  b[5] = rol64(a[3], 28);   b[15] = rol64(a[4], 27);
    b[16] = rol64(a[5], 36);  b[1] = rol64(a[6], 44);   b[11] = rol64(a[7], 6);   b[21] = rol64(a[8], 55);  b[6] = rol64(a[9], 20);
    b[7] = rol64(a[10], 3);   b[17] = rol64(a[11], 10); b[2] = rol64(a[12], 43);  b[12] = rol64(a[13], 25); b[22] = rol64(a[14], 39);
    b[23] = rol64(a[15], 41); b[8] = rol64(a[16], 45);  b[18] = rol64(a[17], 15); b[3] = rol64(a[18], 21);  b[13] = rol64(a[19], 8);
    b[14] = rol64(a[20], 18); b[24] = rol64(a[21], 2);  b[9] = rol64(a[22], 61);  b[19] = rol64(a[23], 56); b[4] = rol64(a[24], 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
#pragma unroll
      for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    a[0] ^= KECCAK_RC[r];
  }
  return s;
}

struct Shake128 {
  KState s;
  uint32_t pos;     // bytes absorbed into the current block (rate 168)
};
AVRF_DI void tr_init(Shake128 &h) {
#pragma unroll
  for (int i = 0; i < 25; i++) h.s.a[i] = 0;
  h.pos = 0;
}
AVRF_DI void shake_xor_lane(KState &s, uint32_t lane, uint64_t v) {
#pragma unroll
  for (int i = 0; i < 21; i++) s.a[i] ^= (lane == (uint32_t)i) ? v : 0ULL;
}
AVRF_DI void tr_byte(Shake128 &h, uint8_t b) {
  shake_xor_lane(h.s, h.pos >> 3, (uint64_t)b << (8 * (h.pos & 7)));
  if (++h.pos == 168) { h.s = keccak_f_nf(h.s); h.pos = 0; }
}

// the XOF reader after finalisation (pad 0x1f ... 0x80, permute): 8-byte words of the output stream, asked for in
// non-decreasing block order (every use here reads forward)
struct ShakeReader {
  KState s;
  uint32_t blk;     // index of the 168-byte output block the state currently holds
};
AVRF_DI ShakeReader tr_reader(const Shake128 &h) {
  ShakeReader r; r.s = h.s; r.blk = 0;
  shake_xor_lane(r.s, h.pos >> 3, (uint64_t)0x1f << (8 * (h.pos & 7)));
  r.s.a[20] ^= 0x8000000000000000ULL;               // byte 167
  r.s = keccak_f_nf(r.s);
  return r;
}
AVRF_DI uint64_t rd_word(ShakeReader &r, uint32_t w) {  // bytes [8 w, 8 w + 8) of the stream, little-endian
  const uint32_t b = w / 21, lane = w - 21 * b;
  while (r.blk < b) { r.s = keccak_f_nf(r.s); r.blk++; }
  uint64_t v = 0;
#pragma unroll
  for (int i = 0; i < 21; i++) v |= (lane == (uint32_t)i) ? r.s.a[i] : 0ULL;
  return v;
}
// i-th 16-byte chunk as four little-endian u32 words
AVRF_DI void rd_chunk16(ShakeReader &r, uint32_t i, uint32_t (&w)[4]) {
  const uint64_t lo = rd_word(r, 2 * i), hi = rd_word(r, 2 * i + 1);
  w[0] = (uint32_t)lo; w[1] = (uint32_t)(lo >> 32); w[2] = (uint32_t)hi; w[3] = (uint32_t)(hi >> 32);
}

}  // namespace avrf
