// host_sha512.h -- host-side SHA-512 for the one inherently sequential hash on the path: the
// batch-verifier weight transcript, which absorbs (c_j, s_j[, sb_j]) of ALL items in order into
// a single Merkle-Damgard chain (src/thin.rs:274-279, src/pedersen.rs:361-367) and therefore
// cannot be spread over GPU lanes.  Product code (not the oracle).
#pragma once
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace avrf {

// Long messages (the 4 MiB weight transcript of a 65 536-item batch), four blocks at a time: the MESSAGE SCHEDULE of the next
// four blocks does not depend on the chaining state, so it runs on the vector pipes -- one 64-bit lane per block, rotates as
// single vprorq (AVX-512VL) -- interleaved step by step with the 320 scalar rounds of the current four, which read K + W
// from a buffer.  The rounds stay one dependency chain (~5 cycles each); what leaves the scalar ports is the third of the
// instructions that were schedule.  EPYC 9575F, 4 MiB: 4.35 -> 3.50 ms (OpenSSL's hand-scheduled assembly: 2.90 ms;
// tools/sha_proto).  An earlier attempt that vectorised the schedule WITHIN one block (two words per step, rotates as three
// AVX2 operations) measured 45 % slower than scalar: the dependency on W[t-2] leaves two lanes, and they compete with the rounds.
namespace sha512_x4 {
#define AVRF_SHA_X4_TARGET __attribute__((target("avx512f,avx512vl,avx2,bmi2")))
inline bool available() { return __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512f"); }
static inline uint64_t ror_(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static inline uint64_t be64_(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }
// wk[t] = W[t] + K[t] of four consecutive blocks at p (lane j = block j); the last sixteen W of the group in wv
AVRF_SHA_X4_TARGET static inline void sched_step(const uint64_t *K, int t, const uint8_t *p, __m256i *wv, uint64_t (*wk)[4]) {
  __m256i w;
  if (t < 16) w = _mm256_set_epi64x((long long)be64_(p + 384 + 8 * t), (long long)be64_(p + 256 + 8 * t), (long long)be64_(p + 128 + 8 * t), (long long)be64_(p + 8 * t));
  else {
    const __m256i w15 = wv[(t + 1) & 15], w2 = wv[(t + 14) & 15];
    const __m256i s0 = _mm256_xor_si256(_mm256_xor_si256(_mm256_ror_epi64(w15, 1), _mm256_ror_epi64(w15, 8)), _mm256_srli_epi64(w15, 7));
    const __m256i s1 = _mm256_xor_si256(_mm256_xor_si256(_mm256_ror_epi64(w2, 19), _mm256_ror_epi64(w2, 61)), _mm256_srli_epi64(w2, 6));
    w = _mm256_add_epi64(_mm256_add_epi64(wv[t & 15], wv[(t + 9) & 15]), _mm256_add_epi64(s0, s1));
  }
  wv[t & 15] = w;
  _mm256_storeu_si256((__m256i *)wk[t], _mm256_add_epi64(w, _mm256_set1_epi64x((long long)K[t])));
}
#define AVRF_X4_RND(A, B, C, D, E, F, G, H, kw)                                                   \
  {                                                                                               \
    uint64_t t1 = H + (ror_(E, 14) ^ ror_(E, 18) ^ ror_(E, 41)) + (G ^ (E & (F ^ G))) + (kw);     \
    uint64_t t2 = (ror_(A, 28) ^ ror_(A, 34) ^ ror_(A, 39)) + ((A & B) | (C & (A | B)));          \
    D += t1; H = t1 + t2;                                                                         \
  }
// h: chaining state; p: groups x 512 bytes (groups >= 1)
AVRF_SHA_X4_TARGET static inline void blocks(const uint64_t *K, uint64_t h[8], const uint8_t *p, size_t groups) {
  alignas(32) uint64_t wk[2][80][4];
  __m256i wv[16];
  for (int t = 0; t < 80; t++) sched_step(K, t, p, wv, wk[0]);
  for (size_t g = 0; g < groups; g++) {
    uint64_t (*cur)[4] = wk[g & 1], (*nxt)[4] = wk[(g + 1) & 1];
    const bool more = g + 1 < groups;
    const uint8_t *pn = p + (g + 1) * 512;
    for (int j = 0; j < 4; j++) {
      uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], gg = h[6], hh = h[7];
      for (int r = 0; r < 80; r += 8) {
        AVRF_X4_RND(a, b, c, d, e, f, gg, hh, cur[r + 0][j]) AVRF_X4_RND(hh, a, b, c, d, e, f, gg, cur[r + 1][j])
        AVRF_X4_RND(gg, hh, a, b, c, d, e, f, cur[r + 2][j]) AVRF_X4_RND(f, gg, hh, a, b, c, d, e, cur[r + 3][j])
        if (more) sched_step(K, j * 20 + r / 4, pn, wv, nxt);
        AVRF_X4_RND(e, f, gg, hh, a, b, c, d, cur[r + 4][j]) AVRF_X4_RND(d, e, f, gg, hh, a, b, c, cur[r + 5][j])
        AVRF_X4_RND(c, d, e, f, gg, hh, a, b, cur[r + 6][j]) AVRF_X4_RND(b, c, d, e, f, gg, hh, a, cur[r + 7][j])
        if (more) sched_step(K, j * 20 + r / 4 + 1, pn, wv, nxt);
      }
      h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += gg; h[7] += hh;
    }
  }
}
#undef AVRF_X4_RND
}  // namespace sha512_x4

class HostSha512 {
 public:
  HostSha512() { reset(); }
  void reset() {
    static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    memcpy(h_, iv, sizeof iv); len_ = 0;
  }
  void update(const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    size_t fill = (size_t)(len_ & 127);
    len_ += n;
    if (fill) {
      size_t take = 128 - fill; if (take > n) take = n;
      memcpy(buf_ + fill, p, take); p += take; n -= take; fill += take;
      if (fill == 128) compress(buf_); else return;
    }
    if (n >= 4096 && sha512_x4::available()) {              // long message: four blocks per step, schedule on the vector pipes
      const size_t groups = n / 512;
      sha512_x4::blocks(ktab(), h_, p, groups);
      p += groups * 512; n -= groups * 512;
    }
    for (; n >= 128; p += 128, n -= 128) compress(p);
    if (n) memcpy(buf_, p, n);
  }
  // digest as 64 bytes
  void final(uint8_t out[64]) {
    size_t fill = (size_t)(len_ & 127); uint64_t bits = len_ * 8;
    buf_[fill++] = 0x80;
    if (fill > 112) { memset(buf_ + fill, 0, 128 - fill); compress(buf_); fill = 0; }
    memset(buf_ + fill, 0, 120 - fill);
    for (int i = 0; i < 8; i++) buf_[120 + i] = (uint8_t)(bits >> (56 - 8 * i));
    compress(buf_);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h_[i] >> (56 - 8 * j));
  }
  const uint64_t *state() const { return h_; }

 private:
  static inline uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  static inline uint64_t be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }
  static const uint64_t *ktab() {
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
    return K;
  }
  void compress(const uint8_t *blk) {
    const uint64_t *K = ktab();
    uint64_t w[16];
    for (int i = 0; i < 16; i++) w[i] = be64(blk + 8 * i);
    uint64_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], hh = h_[7];
    // one round with the eight working variables passed in rotated order (no register shuffling between rounds)
#define AVRF_RND(A, B, C, D, E, F, G, H, kw)                                             \
  {                                                                                      \
    uint64_t t1 = H + (ror(E, 14) ^ ror(E, 18) ^ ror(E, 41)) + (G ^ (E & (F ^ G))) + (kw); \
    uint64_t t2 = (ror(A, 28) ^ ror(A, 34) ^ ror(A, 39)) + ((A & B) | (C & (A | B)));    \
    D += t1; H = t1 + t2;                                                                \
  }
#define AVRF_W(i) (w[(i) & 15] += (ror(w[((i) + 1) & 15], 1) ^ ror(w[((i) + 1) & 15], 8) ^ (w[((i) + 1) & 15] >> 7)) + w[((i) + 9) & 15] + \
                                  (ror(w[((i) + 14) & 15], 19) ^ ror(w[((i) + 14) & 15], 61) ^ (w[((i) + 14) & 15] >> 6)))
#define AVRF_8(r, W)                                                                      \
    AVRF_RND(a, b, c, d, e, f, g, hh, K[r + 0] + W(r + 0)) AVRF_RND(hh, a, b, c, d, e, f, g, K[r + 1] + W(r + 1)) \
    AVRF_RND(g, hh, a, b, c, d, e, f, K[r + 2] + W(r + 2)) AVRF_RND(f, g, hh, a, b, c, d, e, K[r + 3] + W(r + 3)) \
    AVRF_RND(e, f, g, hh, a, b, c, d, K[r + 4] + W(r + 4)) AVRF_RND(d, e, f, g, hh, a, b, c, K[r + 5] + W(r + 5)) \
    AVRF_RND(c, d, e, f, g, hh, a, b, K[r + 6] + W(r + 6)) AVRF_RND(b, c, d, e, f, g, hh, a, K[r + 7] + W(r + 7))
#define AVRF_W0(i) w[(i) & 15]
    AVRF_8(0, AVRF_W0) AVRF_8(8, AVRF_W0)
    for (int r = 16; r < 80; r += 16) { AVRF_8(r, AVRF_W) AVRF_8(r + 8, AVRF_W) }
#undef AVRF_W0
#undef AVRF_8
#undef AVRF_W
#undef AVRF_RND
    h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += hh;
  }
  uint64_t h_[8]; uint8_t buf_[128]; uint64_t len_;
};

}  // namespace avrf
