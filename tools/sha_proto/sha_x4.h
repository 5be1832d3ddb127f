// prototype: SHA-512 over many blocks, message schedule of the NEXT four blocks on the vector pipes while the scalar rounds of
// the current four run
#pragma once
#include <immintrin.h>
#include <stdint.h>
#include <string.h>
namespace shax {
static const uint64_t K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
        0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
        0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
        0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
        0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
        0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
        0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
        0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
        0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
        0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
        0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
        0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
        0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
static inline uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static inline uint64_t be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }

#ifndef SHAX_VARIANT
#define SHAX_VARIANT 1
#endif
#if SHAX_VARIANT == 0
#define SHAX_RND(A, B, C, D, E, F, G, H, kw)                                             \
  {                                                                                      \
    uint64_t t1 = H + (ror(E, 14) ^ ror(E, 18) ^ ror(E, 41)) + (G ^ (E & (F ^ G))) + (kw); \
    uint64_t t2 = (ror(A, 28) ^ ror(A, 34) ^ ror(A, 39)) + ((A & B) | (C & (A | B)));    \
    D += t1; H = t1 + t2;                                                                \
  }
#else
// additions associated so that the two rotate-xor sums join LAST: e' = (d + h + kw + Ch) + S1(e), a' = (h + kw + Ch + Maj) + (S1 + S0)
#define SHAX_RND(A, B, C, D, E, F, G, H, kw)                                             \
  {                                                                                      \
    const uint64_t s1 = ror(E, 14) ^ ror(E, 18) ^ ror(E, 41);                            \
    const uint64_t s0 = ror(A, 28) ^ ror(A, 34) ^ ror(A, 39);                            \
    const uint64_t t = (H + (kw)) + (G ^ (E & (F ^ G)));                                 \
    const uint64_t mj = (A & B) | (C & (A | B));                                         \
    D = (D + t) + s1;                                                                    \
    H = (t + mj) + (s1 + s0);                                                            \
  }
#endif

// one schedule step for four blocks: wk[t] = W[t] + K[t], W ring in wv[16]
__attribute__((target("avx512f,avx512vl,avx2,bmi2"))) static inline void sched_step(int t, const uint8_t *p, __m256i *wv, uint64_t (*wk)[4]) {
  __m256i w;
  if (t < 16) w = _mm256_set_epi64x((long long)be64(p + 384 + 8 * t), (long long)be64(p + 256 + 8 * t), (long long)be64(p + 128 + 8 * t), (long long)be64(p + 8 * t));
  else {
    const __m256i w15 = wv[(t + 1) & 15], w2 = wv[(t + 14) & 15];
    const __m256i s0 = _mm256_xor_si256(_mm256_xor_si256(_mm256_ror_epi64(w15, 1), _mm256_ror_epi64(w15, 8)), _mm256_srli_epi64(w15, 7));
    const __m256i s1 = _mm256_xor_si256(_mm256_xor_si256(_mm256_ror_epi64(w2, 19), _mm256_ror_epi64(w2, 61)), _mm256_srli_epi64(w2, 6));
    w = _mm256_add_epi64(_mm256_add_epi64(wv[t & 15], wv[(t + 9) & 15]), _mm256_add_epi64(s0, s1));
  }
  wv[t & 15] = w;
  _mm256_storeu_si256((__m256i *)wk[t], _mm256_add_epi64(w, _mm256_set1_epi64x((long long)K[t])));
}

// h: chaining state; p: nblocks * 128 bytes, nblocks a multiple of 4 and >= 4
__attribute__((target("avx512f,avx512vl,avx2,bmi2"))) static void blocks_x4(uint64_t h[8], const uint8_t *p, size_t nblocks) {
  alignas(32) uint64_t wk[2][80][4];
  __m256i wv[16];
  for (int t = 0; t < 80; t++) sched_step(t, p, wv, wk[0]);
  const size_t groups = nblocks / 4;
  for (size_t g = 0; g < groups; g++) {
    uint64_t (*cur)[4] = wk[g & 1], (*nxt)[4] = wk[(g + 1) & 1];
    const bool more = g + 1 < groups;
    const uint8_t *pn = p + (g + 1) * 512;
    for (int j = 0; j < 4; j++) {
      uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], gg = h[6], hh = h[7];
      for (int r = 0; r < 80; r += 8) {
        SHAX_RND(a, b, c, d, e, f, gg, hh, cur[r + 0][j]) SHAX_RND(hh, a, b, c, d, e, f, gg, cur[r + 1][j])
        SHAX_RND(gg, hh, a, b, c, d, e, f, cur[r + 2][j]) SHAX_RND(f, gg, hh, a, b, c, d, e, cur[r + 3][j])
        if (more) sched_step(j * 20 + r / 4, pn, wv, nxt);
        SHAX_RND(e, f, gg, hh, a, b, c, d, cur[r + 4][j]) SHAX_RND(d, e, f, gg, hh, a, b, c, cur[r + 5][j])
        SHAX_RND(c, d, e, f, gg, hh, a, b, cur[r + 6][j]) SHAX_RND(b, c, d, e, f, gg, hh, a, cur[r + 7][j])
        if (more) sched_step(j * 20 + r / 4 + 1, pn, wv, nxt);
      }
      h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += gg; h[7] += hh;
    }
  }
}
}  // namespace shax
