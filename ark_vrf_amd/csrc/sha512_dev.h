// sha512_dev.h -- per-lane SHA-512 and the reference's HashTranscript<Sha512> on the device.
//
// Mirrors src/utils/transcript.rs:176-195 (new / absorb_raw / squeeze_raw) and :227-274
// (DigestXof: seed = H(absorbed); block_i = H(seed || LE64(i))).  One transcript per lane:
// the eight state words and the 16-word message block live in VGPRs; bytes are appended
// with shifts so that the block array is only indexed by compile-time constants inside the
// compression function (fully unrolled 16-round groups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avrf {

#ifndef AVRF_DI
#define AVRF_DI __device__ __forceinline__
#endif

static __device__ __constant__ uint64_t SHA512_K[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
    0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
    0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
    0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
    0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
    0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
    0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
    0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
    0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
    0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
    0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
    0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
    0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
    0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
    0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
    0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
    0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

// 64-bit rotate / shift by a compile-time amount as two v_alignbit_b32 on the halves (the uint64_t expressions compiled to
// 64-bit shifts plus an OR: 81 instructions per round against 56 in this form, profiles/r3_* k_thin_prepare)
AVRF_DI uint64_t ror64(uint64_t x, int n) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  uint32_t rl, rh;
  if (n == 32) { rl = hi; rh = lo; }
  else if (n < 32) { rl = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)n); rh = __builtin_amdgcn_alignbit(lo, hi, (uint32_t)n); }
  else { rl = __builtin_amdgcn_alignbit(lo, hi, (uint32_t)(n - 32)); rh = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(n - 32)); }
  return ((uint64_t)rh << 32) | rl;
}
AVRF_DI uint64_t shr64(uint64_t x, int n) {              // 0 < n < 32
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  return ((uint64_t)(hi >> n) << 32) | __builtin_amdgcn_alignbit(hi, lo, (uint32_t)n);
}

// three-input boolean functions of 64-bit words as ONE v_bitop3_b32 per half (gfx950): x ^ y ^ z (truth table 0x96) for the four sigma
// functions, the majority (0xE8) and the choice (0xCA) -- the compiler's own selection left two v_xor_b32 per xor-of-three and four
// and / or / bfi per majority: 60 -> 48 vector instructions per round (tools/kernel_counts.py on sha512_compress_nf)
AVRF_DI uint64_t bitop3_64(uint64_t x, uint64_t y, uint64_t z, const uint32_t tt) {
  uint32_t lo, hi;
  if (tt == 0x96) { lo = __builtin_amdgcn_bitop3_b32((uint32_t)x, (uint32_t)y, (uint32_t)z, 0x96); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(x >> 32), (uint32_t)(y >> 32), (uint32_t)(z >> 32), 0x96); }
  else if (tt == 0xE8) { lo = __builtin_amdgcn_bitop3_b32((uint32_t)x, (uint32_t)y, (uint32_t)z, 0xE8); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(x >> 32), (uint32_t)(y >> 32), (uint32_t)(z >> 32), 0xE8); }
  else { lo = __builtin_amdgcn_bitop3_b32((uint32_t)x, (uint32_t)y, (uint32_t)z, 0xCA); hi = __builtin_amdgcn_bitop3_b32((uint32_t)(x >> 32), (uint32_t)(y >> 32), (uint32_t)(z >> 32), 0xCA); }
  typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
  u32x2_ v; v.x = lo; v.y = hi;
  return __builtin_bit_cast(uint64_t, v);                     // (a register pair, no arithmetic: `hi << 32 | lo` came out as a v_lshl_add_u64)
}

struct Sha512 {
  uint64_t h[8];
  uint64_t w[16];   // current block, big-endian words
  uint32_t fill;    // bytes in the block
  uint32_t total;   // total bytes absorbed (messages here are far below 2^32 bytes)
};

struct ShaH { uint64_t v[8]; };
struct ShaW { uint64_t v[16]; };

// One compression; deliberately NOT inlined (it is ~2k instructions and is reached from dozens of
// absorb sites; arguments and result travel in VGPRs).
__device__ __noinline__ static ShaH sha512_compress_nf(ShaH hin, ShaW win) {
  uint64_t a = hin.v[0], b = hin.v[1], c = hin.v[2], d = hin.v[3], e = hin.v[4], f = hin.v[5], g = hin.v[6], hh = hin.v[7];
  uint64_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = win.v[i];
#pragma unroll 1
  for (int r = 0; r < 80; r += 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (r) {
        uint64_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        uint64_t s0 = bitop3_64(ror64(w15, 1), ror64(w15, 8), shr64(w15, 7), 0x96);
        uint64_t s1 = bitop3_64(ror64(w2, 19), ror64(w2, 61), shr64(w2, 6), 0x96);
        w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
      }
      uint64_t S1 = bitop3_64(ror64(e, 14), ror64(e, 18), ror64(e, 41), 0x96);
      uint64_t ch = bitop3_64(e, f, g, 0xCA);                 // (e & f) | (~e & g)
      uint64_t t1 = hh + S1 + ch + SHA512_K[r + i] + w[i];
      uint64_t S0 = bitop3_64(ror64(a, 28), ror64(a, 34), ror64(a, 39), 0x96);
      uint64_t mj = bitop3_64(a, b, c, 0xE8);                 // majority
      uint64_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
  }
  ShaH o;
  o.v[0] = hin.v[0] + a; o.v[1] = hin.v[1] + b; o.v[2] = hin.v[2] + c; o.v[3] = hin.v[3] + d;
  o.v[4] = hin.v[4] + e; o.v[5] = hin.v[5] + f; o.v[6] = hin.v[6] + g; o.v[7] = hin.v[7] + hh;
  return o;
}
AVRF_DI void sha512_compress(uint64_t (&h)[8], uint64_t (&w)[16]) {
  ShaH hi; ShaW wi;
#pragma unroll
  for (int i = 0; i < 8; i++) hi.v[i] = h[i];
#pragma unroll
  for (int i = 0; i < 16; i++) wi.v[i] = w[i];
  ShaH o = sha512_compress_nf(hi, wi);
#pragma unroll
  for (int i = 0; i < 8; i++) h[i] = o.v[i];
}

AVRF_DI void sha512_init(Sha512 &s) {
  s.h[0] = 0x6a09e667f3bcc908ULL; s.h[1] = 0xbb67ae8584caa73bULL; s.h[2] = 0x3c6ef372fe94f82bULL; s.h[3] = 0xa54ff53a5f1d36f1ULL;
  s.h[4] = 0x510e527fade682d1ULL; s.h[5] = 0x9b05688c2b3e6c1fULL; s.h[6] = 0x1f83d9abfb41bd6bULL; s.h[7] = 0x5be0cd19137e2179ULL;
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0; s.total = 0;
}

// OR a byte into block position `pos` (0..127) without dynamic register indexing.  (The select chains are two thirds of
// k_thin_prepare's instruction stream; the alternative -- the block as a dynamically indexed byte array in the lane's private
// memory, one store per absorbed word -- drops the kernel to 111 VGPRs and measured SLOWER: 141 us against 102 us per 65 536
// items, tools/r3_run18.sh: the block is re-read, byte-swapped and cleared through scratch at every compression.)
AVRF_DI void sha512_put(Sha512 &s, uint32_t pos, uint8_t b) {
  uint32_t wi = pos >> 3;
  uint64_t v = (uint64_t)b << (56 - 8 * (pos & 7));
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] |= (wi == (uint32_t)i) ? v : 0ULL;
}
AVRF_DI void sha512_flush(Sha512 &s) {
  sha512_compress(s.h, s.w);
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
}
AVRF_DI void sha512_byte(Sha512 &s, uint8_t b) {
  sha512_put(s, s.fill, b);
  s.fill++; s.total++;
  if (s.fill == 128) sha512_flush(s);
}
AVRF_DI void sha512_bytes(Sha512 &s, const uint8_t *p, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) sha512_byte(s, p[i]);
}
// absorb a 32-bit little-endian word (4 bytes, lowest first)
AVRF_DI void sha512_u32le(Sha512 &s, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) sha512_byte(s, (uint8_t)(v >> (8 * i)));
}
AVRF_DI void sha512_u64le(Sha512 &s, uint64_t v) { sha512_u32le(s, (uint32_t)v); sha512_u32le(s, (uint32_t)(v >> 32)); }

// digest as eight big-endian words; does not modify the caller's state (pass by value)
AVRF_DI void sha512_final(Sha512 s, uint64_t (&out)[8]) {
  uint64_t bits = (uint64_t)s.total * 8;
  sha512_put(s, s.fill, 0x80);
  if (s.fill >= 112) { sha512_flush(s); }
  s.w[15] = bits;
  sha512_compress(s.h, s.w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = s.h[i];
}

// squeeze block: H(seed(64 bytes, given as eight big-endian words) || LE64(counter))
AVRF_DI void sha512_xof_block(const uint64_t (&seed)[8], uint64_t counter, uint64_t (&out)[8]) {
  Sha512 s; sha512_init(s);
#pragma unroll
  for (int i = 0; i < 8; i++) s.w[i] = seed[i];
  s.w[8] = __builtin_bswap64(counter);   // LE64(counter) as a big-endian word
  s.w[9] = 0x8000000000000000ULL;
  s.w[15] = 72 * 8;
  sha512_compress(s.h, s.w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = s.h[i];
}
// bytes [off, off+16) of a digest (off multiple of 16) as four little-endian u32 words
AVRF_DI void digest_le128(const uint64_t (&dg)[8], int off16, uint32_t (&w)[4]) {
  uint64_t hi = 0, lo = 0;   // dg words are big-endian: byte k of the digest is the (k&7)-th MSB of dg[k>>3]
#pragma unroll
  for (int i = 0; i < 4; i++) { if (off16 == i) { hi = dg[2 * i]; lo = dg[2 * i + 1]; } }
  uint64_t b0 = __builtin_bswap64(hi), b1 = __builtin_bswap64(lo);   // little-endian integers of bytes 0..7, 8..15
  w[0] = (uint32_t)b0; w[1] = (uint32_t)(b0 >> 32); w[2] = (uint32_t)b1; w[3] = (uint32_t)(b1 >> 32);
}

// ---- the transcript interface shared with shake_dev.h (XofTranscript<H>, src/utils/transcript.rs:103-195): kernels are
// generic over the state type T (Sha512 = HashTranscript<Sha512>, Shake128 = Shake128Transcript) through
//   tr_init(T&), tr_byte(T&, b)                      absorb
//   tr_reader(const T&) -> reader R                  finalise a COPY into its squeeze reader (the transcript continues)
//   rd_chunk16(R&, i, w[4])                          bytes [16 i, 16 i + 16) of the squeeze stream, read forward
AVRF_DI void tr_init(Sha512 &s) { sha512_init(s); }
AVRF_DI void tr_byte(Sha512 &s, uint8_t b) { sha512_byte(s, b); }
struct ShaReader { uint64_t seed[8]; uint64_t blk[8]; uint32_t have; };   // DigestXof: block i = H(seed || LE64(i)); `have` = cached block + 1
AVRF_DI ShaReader tr_reader(const Sha512 &s) { ShaReader r; sha512_final(s, r.seed); r.have = 0; return r; }
AVRF_DI void rd_chunk16(ShaReader &r, uint32_t i, uint32_t (&w)[4]) {
  if (r.have != (i >> 2) + 1) { sha512_xof_block(r.seed, i >> 2, r.blk); r.have = (i >> 2) + 1; }
  digest_le128(r.blk, (int)(i & 3), w);
}
// a 32-bit little-endian word at a 4-byte-aligned position: ONE select chain over the block words instead of four (field
// elements, counts and lengths are absorbed as such words; the positions stay aligned until a variable-length string -- the
// additional data -- has gone in)
AVRF_DI void tr_u32le(Sha512 &s, uint32_t v) {
  if ((s.fill & 3) == 0) {
    const uint64_t x = (uint64_t)__builtin_bswap32(v) << ((s.fill & 4) ? 0 : 32);
    const uint32_t wi = s.fill >> 3;
#pragma unroll
    for (int i = 0; i < 16; i++) s.w[i] |= (wi == (uint32_t)i) ? x : 0ULL;
    s.fill += 4; s.total += 4;
    if (s.fill == 128) sha512_flush(s);
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) sha512_byte(s, (uint8_t)(v >> (8 * i)));
  }
}
template <class T> AVRF_DI void tr_bytes(T &s, const uint8_t *p, uint32_t n) { for (uint32_t i = 0; i < n; i++) tr_byte(s, p[i]); }
template <class T> AVRF_DI void tr_u32le(T &s, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) tr_byte(s, (uint8_t)(v >> (8 * i)));
}
template <class T> AVRF_DI void tr_u64le(T &s, uint64_t v) { tr_u32le(s, (uint32_t)v); tr_u32le(s, (uint32_t)(v >> 32)); }

}  // namespace avrf
