// host_sha512_mb.h -- eight weight transcripts at once: multi-buffer SHA-512 on AVX-512.
//
// The batch verifiers' weight transcript (src/thin.rs:274-279, src/pedersen.rs:361-367) is ONE Merkle-Damgard chain over
// all (c_j, s_j) of a batch: 4 MB, 4.6 ms on a core, and nothing inside one chain runs in parallel.  But a GPU is fed by many
// contexts, each with its own batch and its own chain, and the chains of DIFFERENT batches are independent: lane l of a
// 512-bit register carries the state of batch l, and one pass of the 80 rounds advances eight chains.  Measured on the
// bench host the scalar hashes of 16 contexts cost the pipeline 0.15-0.25 ms per batch (tools/gpu_only_rate.py: 0.64 ms per
// batch without them, 0.8-0.89 ms with).  This file is the attempt to remove that; it hashes 3.6x more per core-second but
// did NOT pay in the closed loop of 16 contexts (capi.hip, WeightHashService: opt-in, with the numbers).
//
// The routines are compiled by g++ (host_hash.cpp -> host_hash.o, csrc/Makefile), not by hipcc's clang: on the bench host (EPYC 9575F)
// g++'s code hashes a batch in 0.82 ms of one core with eight lanes and 0.61 ms with two interleaved groups of eight, clang's in
// 1.01 / 1.14 ms (it spills the second group's message schedule); tools/hostbench/run.sh.  Units compiled by hipcc see the
// declarations only.
//
// Message of a lane: prefix (suite id || 0x50) followed by n records  c(16) || 0(16) || resp(rsz)   (rsz = 32 Thin, 64 Pedersen).
#pragma once
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace avrf {

struct WeightJob {
  const uint8_t *prefix; size_t prefix_len;
  const uint8_t *c16; const uint8_t *resp; size_t n, rsz;
  // or the whole message in one piece (prefix and records contiguous, as the prepare kernels deliver it): msg != nullptr
  const uint8_t *msg = nullptr; size_t msg_len = 0;
  uint8_t digest[64];
};

inline bool sha512_mb_available() { return __builtin_cpu_supports("avx512f"); }
inline bool sha512_mb16_available() { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw"); }
// digests of up to eight / sixteen weight transcripts (lanes may differ in length: a finished lane keeps its state); sixteen = two
// groups of eight advanced in one interleaved round loop (count <= 8: one group through the same transposing loader)
void sha512_weights_x8(WeightJob *const *jobs, int count);
void sha512_weights_x16(WeightJob *const *jobs, int count);

#ifdef AVRF_SHA_MB_IMPL
namespace mb_detail {

static const uint64_t K512[80] = {
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

// the byte stream of one lane, handed out in 128-byte blocks, padding included
struct LaneFeed {
  const WeightJob *job = nullptr;
  size_t rec = 0;             // next record to append
  bool prefix_done = false, padded = false, finished = true;
  uint8_t buf[384]; size_t have = 0;
  uint64_t total = 0;         // message bytes appended so far (without padding)
  size_t pos = 0;             // contiguous form: bytes of job->msg handed out so far
  void start(const WeightJob *j) { job = j; rec = 0; prefix_done = false; padded = false; finished = false; have = 0; total = 0; pos = 0; }
  // contiguous form: a pointer to the next whole block inside the message, or nullptr when only the tail (+ padding) is left
  const uint8_t *next_direct() {
    if (finished || !job->msg || job->msg_len - pos < 128) return nullptr;
    const uint8_t *p = job->msg + pos; pos += 128; return p;
  }
  // the next block; false when the message (with its padding) is exhausted
  bool next(uint8_t out[128]) {
    if (finished) return false;
    if (job->msg && !padded) {                                   // tail of a contiguous message: the last < 128 bytes, then the padding
      have = job->msg_len - pos; memcpy(buf, job->msg + pos, have); pos = job->msg_len;
      const uint64_t bits = (uint64_t)job->msg_len * 8;
      buf[have++] = 0x80;
      while (have % 128 != 112) buf[have++] = 0;
      memset(buf + have, 0, 8); have += 8;
      for (int i = 0; i < 8; i++) buf[have++] = (uint8_t)(bits >> (56 - 8 * i));
      padded = true;
    }
    while (have < 128 && !padded) {
      if (!prefix_done) { memcpy(buf + have, job->prefix, job->prefix_len); have += job->prefix_len; total += job->prefix_len; prefix_done = true; continue; }
      if (rec < job->n) {
        uint8_t *p = buf + have;
        memcpy(p, job->c16 + 16 * rec, 16); memset(p + 16, 0, 16); memcpy(p + 32, job->resp + job->rsz * rec, job->rsz);
        have += 32 + job->rsz; total += 32 + job->rsz; rec++;
        continue;
      }
      // end of the message: 0x80, zeros up to 112 mod 128, the bit length as a 128-bit big-endian integer
      const uint64_t bits = total * 8;
      buf[have++] = 0x80;
      while (have % 128 != 112) buf[have++] = 0;
      memset(buf + have, 0, 8); have += 8;
      for (int i = 0; i < 8; i++) buf[have++] = (uint8_t)(bits >> (56 - 8 * i));
      padded = true;
    }
    memcpy(out, buf, 128);
    have -= 128; memmove(buf, buf + 128, have);
    if (padded && have == 0) finished = true;
    return true;
  }
};

#define AVRF_MB_TARGET __attribute__((target("avx512f")))

AVRF_MB_TARGET static inline __m512i xor3(__m512i a, __m512i b, __m512i c) { return _mm512_ternarylogic_epi64(a, b, c, 0x96); }

// one compression on eight lanes; W: the sixteen message words, word t of lane l in lane l of W[t]
AVRF_MB_TARGET static inline void compress_x8(__m512i (&H)[8], __m512i (&W)[16]) {
  __m512i a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
#pragma unroll
  for (int r = 0; r < 80; r++) {
    const int i = r & 15;
    if (r >= 16) {
      const __m512i w15 = W[(i + 1) & 15], w2 = W[(i + 14) & 15];
      const __m512i s0 = xor3(_mm512_ror_epi64(w15, 1), _mm512_ror_epi64(w15, 8), _mm512_srli_epi64(w15, 7));
      const __m512i s1 = xor3(_mm512_ror_epi64(w2, 19), _mm512_ror_epi64(w2, 61), _mm512_srli_epi64(w2, 6));
      W[i] = _mm512_add_epi64(_mm512_add_epi64(W[i], s0), _mm512_add_epi64(W[(i + 9) & 15], s1));
    }
    const __m512i S1 = xor3(_mm512_ror_epi64(e, 14), _mm512_ror_epi64(e, 18), _mm512_ror_epi64(e, 41));
    const __m512i ch = _mm512_ternarylogic_epi64(e, f, g, 0xCA);
    const __m512i t1 = _mm512_add_epi64(_mm512_add_epi64(h, S1), _mm512_add_epi64(_mm512_add_epi64(ch, _mm512_set1_epi64((long long)K512[r])), W[i]));
    const __m512i S0 = xor3(_mm512_ror_epi64(a, 28), _mm512_ror_epi64(a, 34), _mm512_ror_epi64(a, 39));
    const __m512i mj = _mm512_ternarylogic_epi64(a, b, c, 0xE8);
    const __m512i t2 = _mm512_add_epi64(S0, mj);
    h = g; g = f; f = e; e = _mm512_add_epi64(d, t1); d = c; c = b; b = a; a = _mm512_add_epi64(t1, t2);
  }
  H[0] = _mm512_add_epi64(H[0], a); H[1] = _mm512_add_epi64(H[1], b); H[2] = _mm512_add_epi64(H[2], c); H[3] = _mm512_add_epi64(H[3], d);
  H[4] = _mm512_add_epi64(H[4], e); H[5] = _mm512_add_epi64(H[5], f); H[6] = _mm512_add_epi64(H[6], g); H[7] = _mm512_add_epi64(H[7], h);
}

}  // namespace mb_detail


// digests of up to eight weight transcripts (lanes may differ in length: a finished lane keeps its state)
AVRF_MB_TARGET static inline void sha512_weights_x8_impl(WeightJob *const *jobs, int count) {
  using namespace mb_detail;
  static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  LaneFeed feed[8];
  for (int l = 0; l < count; l++) feed[l].start(jobs[l]);
  __m512i H[8];
  for (int i = 0; i < 8; i++) H[i] = _mm512_set1_epi64((long long)iv[i]);
  alignas(64) uint64_t words[16][8];
  memset(words, 0, sizeof words);
  uint8_t blk[128];
  for (;;) {
    unsigned active = 0;
    for (int l = 0; l < count; l++) {
      const uint8_t *p = feed[l].next_direct();
      if (!p) { if (!feed[l].next(blk)) continue; p = blk; }
      active |= 1u << l;
      for (int t = 0; t < 16; t++) { uint64_t v; memcpy(&v, p + 8 * t, 8); words[t][l] = __builtin_bswap64(v); }
    }
    if (!active) break;
    __m512i W[16], Hn[8];
    for (int t = 0; t < 16; t++) W[t] = _mm512_load_si512((const void *)words[t]);
    for (int i = 0; i < 8; i++) Hn[i] = H[i];
    compress_x8(Hn, W);
    for (int i = 0; i < 8; i++) H[i] = _mm512_mask_blend_epi64((__mmask8)active, H[i], Hn[i]);
  }
  alignas(64) uint64_t out[8][8];
  for (int i = 0; i < 8; i++) _mm512_store_si512((void *)out[i], H[i]);
  for (int l = 0; l < count; l++)
    for (int i = 0; i < 8; i++) for (int k = 0; k < 8; k++) jobs[l]->digest[8 * i + k] = (uint8_t)(out[i][l] >> (56 - 8 * k));
}


// ---------------------------------------------------------------- sixteen chains: two groups of eight, interleaved (round 4)
// One 8-lane compression is a latency chain: e -> Sigma1 / Ch -> three additions -> e of the next round, about 12 cycles per
// round on the bench host against ~7 of pure port time (28 vector operations).  Two INDEPENDENT groups advanced in the same
// loop give the out-of-order core a second chain to fill the gaps with; the message words live in memory (2 x 16 registers
// of schedule + 2 x 8 of state do not fit in 32 zmm), and the 16 x 128 message bytes of a group are transposed with
// unpack / 128-bit shuffles and byte-swapped with one vpshufb per register instead of 128 scalar load-swap-store triples.
#define AVRF_MB16_TARGET __attribute__((target("avx512f,avx512bw")))

namespace mb_detail {

// rows[l] = the next 128-byte block of lane l (two loads each) -> W[t] lane l = big-endian word t of lane l
AVRF_MB16_TARGET static inline void load_transpose_x8(const uint8_t *const rows[8], __m512i W[16]) {
  const __m512i bswap = _mm512_set_epi8(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7,
                                        8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
  for (int half = 0; half < 2; half++) {
    __m512i r[8], t[8], u[8];
    for (int l = 0; l < 8; l++) r[l] = _mm512_shuffle_epi8(_mm512_loadu_si512((const void *)(rows[l] + 64 * half)), bswap);
    for (int k = 0; k < 4; k++) { t[2 * k] = _mm512_unpacklo_epi64(r[2 * k], r[2 * k + 1]); t[2 * k + 1] = _mm512_unpackhi_epi64(r[2 * k], r[2 * k + 1]); }
    // t[2k + p]: 128-bit block b = words (2b + p) of lanes 2k, 2k + 1
    for (int p = 0; p < 2; p++) {
      u[0 + p] = _mm512_shuffle_i64x2(t[0 + p], t[2 + p], 0x88); u[2 + p] = _mm512_shuffle_i64x2(t[0 + p], t[2 + p], 0xDD);
      u[4 + p] = _mm512_shuffle_i64x2(t[4 + p], t[6 + p], 0x88); u[6 + p] = _mm512_shuffle_i64x2(t[4 + p], t[6 + p], 0xDD);
      W[8 * half + 0 + p] = _mm512_shuffle_i64x2(u[0 + p], u[4 + p], 0x88);
      W[8 * half + 4 + p] = _mm512_shuffle_i64x2(u[0 + p], u[4 + p], 0xDD);
      W[8 * half + 2 + p] = _mm512_shuffle_i64x2(u[2 + p], u[6 + p], 0x88);
      W[8 * half + 6 + p] = _mm512_shuffle_i64x2(u[2 + p], u[6 + p], 0xDD);
    }
  }
}

#define AVRF_MB_ROUND(a, b, c, d, e, f, g, h, W, r)                                                                              \
  do {                                                                                                                         \
    const int i_ = (r) & 15;                                                                                                   \
    if ((r) >= 16) {                                                                                                           \
      const __m512i w15 = W[(i_ + 1) & 15], w2 = W[(i_ + 14) & 15];                                                            \
      const __m512i s0 = xor3(_mm512_ror_epi64(w15, 1), _mm512_ror_epi64(w15, 8), _mm512_srli_epi64(w15, 7));                  \
      const __m512i s1 = xor3(_mm512_ror_epi64(w2, 19), _mm512_ror_epi64(w2, 61), _mm512_srli_epi64(w2, 6));                   \
      W[i_] = _mm512_add_epi64(_mm512_add_epi64(W[i_], s0), _mm512_add_epi64(W[(i_ + 9) & 15], s1));                           \
    }                                                                                                                          \
    const __m512i hk = _mm512_add_epi64(h, _mm512_add_epi64(W[i_], _mm512_set1_epi64((long long)K512[r]))); /* off the chain */ \
    const __m512i dhk = _mm512_add_epi64(d, hk);                                                            /* off the chain */ \
    const __m512i S1 = xor3(_mm512_ror_epi64(e, 14), _mm512_ror_epi64(e, 18), _mm512_ror_epi64(e, 41));                        \
    const __m512i sc = _mm512_add_epi64(S1, _mm512_ternarylogic_epi64(e, f, g, 0xCA));                                         \
    const __m512i S0 = xor3(_mm512_ror_epi64(a, 28), _mm512_ror_epi64(a, 34), _mm512_ror_epi64(a, 39));                        \
    const __m512i t2 = _mm512_add_epi64(_mm512_add_epi64(S0, _mm512_ternarylogic_epi64(a, b, c, 0xE8)), hk);                   \
    d = _mm512_add_epi64(dhk, sc);                            /* e of the next round: rotate, xor3, add, add */                \
    h = _mm512_add_epi64(t2, sc);                             /* a of the next round */                                        \
  } while (0)

// one compression of two independent groups; state rotates through the argument order instead of being moved
AVRF_MB16_TARGET static inline void compress_x16(__m512i (&HA)[8], __m512i (&WA)[16], __m512i (&HB)[8], __m512i (&WB)[16]) {
  __m512i a = HA[0], b = HA[1], c = HA[2], d = HA[3], e = HA[4], f = HA[5], g = HA[6], h = HA[7];
  __m512i p = HB[0], q = HB[1], s = HB[2], t = HB[3], u = HB[4], v = HB[5], w = HB[6], x = HB[7];
#pragma unroll
  for (int r = 0; r < 80; r += 8) {
    AVRF_MB_ROUND(a, b, c, d, e, f, g, h, WA, r + 0); AVRF_MB_ROUND(p, q, s, t, u, v, w, x, WB, r + 0);
    AVRF_MB_ROUND(h, a, b, c, d, e, f, g, WA, r + 1); AVRF_MB_ROUND(x, p, q, s, t, u, v, w, WB, r + 1);
    AVRF_MB_ROUND(g, h, a, b, c, d, e, f, WA, r + 2); AVRF_MB_ROUND(w, x, p, q, s, t, u, v, WB, r + 2);
    AVRF_MB_ROUND(f, g, h, a, b, c, d, e, WA, r + 3); AVRF_MB_ROUND(v, w, x, p, q, s, t, u, WB, r + 3);
    AVRF_MB_ROUND(e, f, g, h, a, b, c, d, WA, r + 4); AVRF_MB_ROUND(u, v, w, x, p, q, s, t, WB, r + 4);
    AVRF_MB_ROUND(d, e, f, g, h, a, b, c, WA, r + 5); AVRF_MB_ROUND(t, u, v, w, x, p, q, s, WB, r + 5);
    AVRF_MB_ROUND(c, d, e, f, g, h, a, b, WA, r + 6); AVRF_MB_ROUND(s, t, u, v, w, x, p, q, WB, r + 6);
    AVRF_MB_ROUND(b, c, d, e, f, g, h, a, WA, r + 7); AVRF_MB_ROUND(q, s, t, u, v, w, x, p, WB, r + 7);
  }
  HA[0] = _mm512_add_epi64(HA[0], a); HA[1] = _mm512_add_epi64(HA[1], b); HA[2] = _mm512_add_epi64(HA[2], c); HA[3] = _mm512_add_epi64(HA[3], d);
  HA[4] = _mm512_add_epi64(HA[4], e); HA[5] = _mm512_add_epi64(HA[5], f); HA[6] = _mm512_add_epi64(HA[6], g); HA[7] = _mm512_add_epi64(HA[7], h);
  HB[0] = _mm512_add_epi64(HB[0], p); HB[1] = _mm512_add_epi64(HB[1], q); HB[2] = _mm512_add_epi64(HB[2], s); HB[3] = _mm512_add_epi64(HB[3], t);
  HB[4] = _mm512_add_epi64(HB[4], u); HB[5] = _mm512_add_epi64(HB[5], v); HB[6] = _mm512_add_epi64(HB[6], w); HB[7] = _mm512_add_epi64(HB[7], x);
}
AVRF_MB16_TARGET static inline void compress_x8r(__m512i (&HA)[8], __m512i (&WA)[16]) {
  __m512i a = HA[0], b = HA[1], c = HA[2], d = HA[3], e = HA[4], f = HA[5], g = HA[6], h = HA[7];
#pragma unroll
  for (int r = 0; r < 80; r += 8) {
    AVRF_MB_ROUND(a, b, c, d, e, f, g, h, WA, r + 0); AVRF_MB_ROUND(h, a, b, c, d, e, f, g, WA, r + 1);
    AVRF_MB_ROUND(g, h, a, b, c, d, e, f, WA, r + 2); AVRF_MB_ROUND(f, g, h, a, b, c, d, e, WA, r + 3);
    AVRF_MB_ROUND(e, f, g, h, a, b, c, d, WA, r + 4); AVRF_MB_ROUND(d, e, f, g, h, a, b, c, WA, r + 5);
    AVRF_MB_ROUND(c, d, e, f, g, h, a, b, WA, r + 6); AVRF_MB_ROUND(b, c, d, e, f, g, h, a, WA, r + 7);
  }
  HA[0] = _mm512_add_epi64(HA[0], a); HA[1] = _mm512_add_epi64(HA[1], b); HA[2] = _mm512_add_epi64(HA[2], c); HA[3] = _mm512_add_epi64(HA[3], d);
  HA[4] = _mm512_add_epi64(HA[4], e); HA[5] = _mm512_add_epi64(HA[5], f); HA[6] = _mm512_add_epi64(HA[6], g); HA[7] = _mm512_add_epi64(HA[7], h);
}

}  // namespace mb_detail


// digests of up to sixteen weight transcripts: lanes 0-7 and 8-15 are two groups advanced in one interleaved round loop
// (count <= 8: one group through the same transposing loader)
AVRF_MB16_TARGET static inline void sha512_weights_x16_impl(WeightJob *const *jobs, int count) {
  using namespace mb_detail;
  static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  alignas(64) static const uint8_t zero_block[128] = {0};
  LaneFeed feed[16];
  for (int l = 0; l < count; l++) feed[l].start(jobs[l]);
  const int groups = count > 8 ? 2 : 1;
  __m512i H[2][8];
  for (int gI = 0; gI < 2; gI++) for (int i = 0; i < 8; i++) H[gI][i] = _mm512_set1_epi64((long long)iv[i]);
  alignas(64) uint8_t tail[16][128];
  for (;;) {
    unsigned active = 0;
    const uint8_t *rows[16];
    for (int l = 0; l < 8 * groups; l++) {
      rows[l] = zero_block;
      if (l >= count) continue;
      const uint8_t *p = feed[l].next_direct();
      if (!p) { if (!feed[l].next(tail[l])) continue; p = tail[l]; }
      rows[l] = p; active |= 1u << l;
    }
    if (!active) break;
    __m512i W[2][16], Hn[2][8];
    for (int gI = 0; gI < groups; gI++) { load_transpose_x8(rows + 8 * gI, W[gI]); for (int i = 0; i < 8; i++) Hn[gI][i] = H[gI][i]; }
    if (groups == 2 && (active >> 8)) compress_x16(Hn[0], W[0], Hn[1], W[1]);
    else compress_x8r(Hn[0], W[0]);
    for (int gI = 0; gI < groups; gI++)
      for (int i = 0; i < 8; i++) H[gI][i] = _mm512_mask_blend_epi64((__mmask8)(active >> (8 * gI)), H[gI][i], Hn[gI][i]);
  }
  alignas(64) uint64_t out[8][8];
  for (int gI = 0; gI < groups; gI++) {
    for (int i = 0; i < 8; i++) _mm512_store_si512((void *)out[i], H[gI][i]);
    for (int l = 8 * gI; l < count && l < 8 * gI + 8; l++)
      for (int i = 0; i < 8; i++) for (int k = 0; k < 8; k++) jobs[l]->digest[8 * i + k] = (uint8_t)(out[i][l - 8 * gI] >> (56 - 8 * k));
  }
}

#endif  // AVRF_SHA_MB_IMPL

}  // namespace avrf
