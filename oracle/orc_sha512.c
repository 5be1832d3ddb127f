/*
 * oracle/orc_sha512.c -- SHA-512 (FIPS 180-4) and the reference's HashTranscript (TEST ORACLE).
 *
 * SHA-512 restates the third-party `sha2` 0.10 crate (Cargo.toml:19-21) from the
 * published standard.  The transcript follows src/utils/transcript.rs:
 *   XofTranscript::new        :176-182  -> tr_new
 *   absorb_raw                :184-189  -> tr_absorb   (absorb after squeeze panics there; here it is ignored+flagged)
 *   squeeze_raw               :191-194  -> tr_squeeze
 *   DigestXof::finalize_xof   :227-240  -> first squeeze: seed = H(absorbed), block_0 = H(seed || LE64(0))
 *   DigestXofReader::read     :252-273  -> block_i = H(seed || LE64(i)), consumed bytewise
 */
#include "orc.h"
#include <string.h>

static const uint64_t K[80] = {
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

#define ROR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))

static void compress(uint64_t h[8], const uint8_t blk[128]) {
    uint64_t w[80];
    for (int i = 0; i < 16; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | blk[8 * i + j];
        w[i] = v;
    }
    for (int i = 16; i < 80; i++) {
        uint64_t s0 = ROR(w[i - 15], 1) ^ ROR(w[i - 15], 8) ^ (w[i - 15] >> 7);
        uint64_t s1 = ROR(w[i - 2], 19) ^ ROR(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; i++) {
        uint64_t S1 = ROR(e, 14) ^ ROR(e, 18) ^ ROR(e, 41);
        uint64_t ch = (e & f) ^ (~e & g);
        uint64_t t1 = hh + S1 + ch + K[i] + w[i];
        uint64_t S0 = ROR(a, 28) ^ ROR(a, 34) ^ ROR(a, 39);
        uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint64_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void sha512_init(sha512_t *c) {
    static const uint64_t iv[8] = {
        0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    memcpy(c->h, iv, sizeof iv); c->len = 0;
}
void sha512_update(sha512_t *c, const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    size_t fill = (size_t)(c->len % 128);
    c->len += n;
    if (fill) {
        size_t take = 128 - fill; if (take > n) take = n;
        memcpy(c->buf + fill, p, take); p += take; n -= take; fill += take;
        if (fill == 128) compress(c->h, c->buf); else return;
    }
    while (n >= 128) { compress(c->h, p); p += 128; n -= 128; }
    if (n) memcpy(c->buf, p, n);
}
void sha512_final(const sha512_t *cc, uint8_t out[64]) {
    sha512_t c = *cc;
    size_t fill = (size_t)(c.len % 128);
    uint64_t bits = c.len * 8;
    c.buf[fill++] = 0x80;
    if (fill > 112) { memset(c.buf + fill, 0, 128 - fill); compress(c.h, c.buf); fill = 0; }
    memset(c.buf + fill, 0, 120 - fill);
    for (int i = 0; i < 8; i++) c.buf[120 + i] = (uint8_t)(bits >> (56 - 8 * i));
    compress(c.h, c.buf);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(c.h[i] >> (56 - 8 * j));
}

/* ---- HashTranscript<Sha512> ---- */
/* ---- Keccak-f[1600] / SHAKE128 (FIPS 202): rate 168 bytes, domain suffix 0x1f, final bit 0x80 ---- */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_f(uint64_t a[25]) {
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; y++) for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[r];
    }
}
#define SHAKE_RATE 168
static void shake_absorb(uint64_t ks[25], size_t *pos, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        ks[*pos >> 3] ^= (uint64_t)d[i] << (8 * (*pos & 7));
        if (++*pos == SHAKE_RATE) { keccak_f(ks); *pos = 0; }
    }
}
static void shake_pad(uint64_t ks[25], size_t *pos) {
    ks[*pos >> 3] ^= (uint64_t)0x1f << (8 * (*pos & 7));
    ks[(SHAKE_RATE - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((SHAKE_RATE - 1) & 7));
    keccak_f(ks); *pos = 0;
}
static void shake_read(uint64_t ks[25], size_t *pos, uint8_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (*pos == SHAKE_RATE) { keccak_f(ks); *pos = 0; }
        o[i] = (uint8_t)(ks[*pos >> 3] >> (8 * (*pos & 7))); ++*pos;
    }
}
void shake128(uint8_t *out, size_t out_len, const uint8_t *const *parts, const size_t *lens, int n_parts) {
    uint64_t ks[25]; size_t pos = 0; memset(ks, 0, sizeof ks);
    for (int i = 0; i < n_parts; i++) shake_absorb(ks, &pos, parts[i], lens[i]);
    shake_pad(ks, &pos); shake_read(ks, &pos, out, out_len);
}

/* ---- SHA-256 (FIPS 180-4), for HashTranscript<Sha256>: seed = H(label || absorbed), block_i = H(seed || LE64(i)), 32-byte blocks ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t b[64]) {
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    memcpy(a, h, sizeof a);
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ror32(a[4], 6) ^ ror32(a[4], 11) ^ ror32(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
        uint32_t t1 = a[7] + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ror32(a[0], 2) ^ ror32(a[0], 13) ^ ror32(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
        uint32_t t2 = S0 + mj;
        a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += a[i];
}
static void sha256_init_(uint32_t h[8], uint64_t *len) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, iv, sizeof iv); *len = 0;
}
static void sha256_update_(uint32_t h[8], uint8_t buf[64], uint64_t *len, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { buf[*len & 63] = d[i]; if ((++*len & 63) == 0) sha256_block(h, buf); }
}
static void sha256_final_(const uint32_t hin[8], const uint8_t bufin[64], uint64_t len, uint8_t out[32]) {
    uint32_t h[8]; uint8_t buf[64]; memcpy(h, hin, sizeof h); memcpy(buf, bufin, 64);
    uint64_t l = len; uint8_t pad = 0x80, z = 0;
    sha256_update_(h, buf, &l, &pad, 1);
    while ((l & 63) != 56) sha256_update_(h, buf, &l, &z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)((len * 8) >> (56 - 8 * i));
    sha256_update_(h, buf, &l, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

void tr_new_mode(transcript_t *t, const void *label, size_t n, int shake) {
    tr_new(t, label, n);
    t->shake = shake;
    if (shake == 1) { memset(t->ks, 0, sizeof t->ks); t->kpos = 0; shake_absorb(t->ks, &t->kpos, (const uint8_t *)label, n); }
    if (shake == 2) { sha256_init_(t->h2, &t->len2); sha256_update_(t->h2, t->buf2, &t->len2, (const uint8_t *)label, n); }
}
void tr_new(transcript_t *t, const void *label, size_t n) {
    t->shake = 0;
    sha512_init(&t->h); t->squeezing = 0; t->counter = 0; t->off = 64;
    sha512_update(&t->h, label, n);
}
void tr_absorb(transcript_t *t, const void *d, size_t n) {
    if (t->squeezing) { t->squeezing = 2; return; } /* reference panics (transcript.rs:187) */
    if (t->shake == 1) { shake_absorb(t->ks, &t->kpos, (const uint8_t *)d, n); return; }
    if (t->shake == 2) { sha256_update_(t->h2, t->buf2, &t->len2, (const uint8_t *)d, n); return; }
    sha512_update(&t->h, d, n);
}
static void tr_block(transcript_t *t) {
    sha512_t h; uint8_t ctr[8];
    for (int i = 0; i < 8; i++) ctr[i] = (uint8_t)(t->counter >> (8 * i));
    sha512_init(&h); sha512_update(&h, t->seed, 64); sha512_update(&h, ctr, 8);
    sha512_final(&h, t->block);
    t->counter++; t->off = 0;
}
void tr_squeeze(transcript_t *t, void *out, size_t n) {
    uint8_t *o = (uint8_t *)out;
    if (t->shake == 1) {
        if (!t->squeezing) { shake_pad(t->ks, &t->kpos); t->squeezing = 1; }
        shake_read(t->ks, &t->kpos, o, n);
        return;
    }
    if (t->shake == 2) {   /* DigestXof<Sha256>: 32-byte seed and blocks */
        if (!t->squeezing) { sha256_final_(t->h2, t->buf2, t->len2, t->seed); t->squeezing = 1; t->counter = 0; t->off = 32; }
        while (n) {
            if (t->off >= 32) {
                uint32_t h[8]; uint8_t buf[64]; uint64_t l; uint8_t ctr[8];
                for (int i = 0; i < 8; i++) ctr[i] = (uint8_t)(t->counter >> (8 * i));
                sha256_init_(h, &l); sha256_update_(h, buf, &l, t->seed, 32); sha256_update_(h, buf, &l, ctr, 8);
                sha256_final_(h, buf, l, t->block); t->counter++; t->off = 0;
            }
            size_t take = 32 - t->off; if (take > n) take = n;
            memcpy(o, t->block + t->off, take); t->off += take; o += take; n -= take;
        }
        return;
    }
    if (!t->squeezing) { sha512_final(&t->h, t->seed); t->squeezing = 1; t->counter = 0; t->off = 64; }
    while (n) {
        if (t->off >= 64) tr_block(t);
        size_t take = 64 - t->off; if (take > n) take = n;
        memcpy(o, t->block + t->off, take); t->off += take; o += take; n -= take;
    }
}
