/*
 * oracle/orc_vrf.c -- protocol layer of the ark-vrf hot path, restated (TEST ORACLE).
 *
 * Follows, function by function:
 *   src/utils/common.rs   : challenge_scalar :72-76, nonce_scalar :63-67, vrf_transcript_base :159-173,
 *                           vrf_transcript_from_iter :181-202, chain_ios :231-240, challenge :270-280,
 *                           point_to_hash :290-305, nonce :313-328, DelinearizeScalars :335-369,
 *                           absorb_ios :377-383, merge_ios :389-419
 *   src/lib.rs            : Secret::from_seed :346-369, from_scalar :331-334, output :391-393
 *   src/thin.rs           : prove :111-129, verify :131-165, BatchVerifier::{prepare,verify} :209-226,257-325
 *   src/pedersen.rs       : blinding :51-54, prove :136-186, verify :188-249, BatchItem::new :276-293,
 *                           BatchVerifier::verify :341-426
 *   src/utils/hash_to_curve.rs : tai :34-57, ell2_xmd :66-100 (+ ark-ec Elligator2Map, ark-ff
 *                           DefaultFieldHasher / expand_message_xmd, third-party)
 * Byte formats per SURVEY.md Appendix A.1.  All exported functions take/return
 * canonical encodings (compressed 32-byte points, LE32 scalars).
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

#define FQ (&s->fq)
#define FR (&s->fr)

enum { DS_TINY = 0x00, DS_THIN = 0x01, DS_PEDERSEN = 0x02, DS_NONCE_EXPAND = 0x10, DS_NONCE = 0x11,
       DS_PEDERSEN_BLINDING = 0x12, DS_POINT_TO_HASH = 0x20, DS_DELINEARIZE = 0x30, DS_CHALLENGE = 0x40,
       DS_BATCH_VERIFY = 0x50, DS_HASH_TO_CURVE = 0x60 }; /* src/utils/common.rs:128-152 */

static void absorb_u8(transcript_t *t, uint8_t b) { tr_absorb(t, &b, 1); }
static void absorb_u64(transcript_t *t, uint64_t v) {
    uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); tr_absorb(t, b, 8);
}
static void absorb_point(transcript_t *t, const te_aff *p, const suite_t *s) {
    if (s->sw_codec || s->sw_native) { uint8_t b[33]; sw_encode(b, p, s); tr_absorb(t, b, 33); return; }   /* Affine = SWAffine: 33-byte form */
    uint8_t b[32]; te_encode(b, p, s); tr_absorb(t, b, 32);
}
static void absorb_scalar(transcript_t *t, const u256 *k_mont, const suite_t *s) {
    u256 k; uint8_t b[32]; mont_from(&k, k_mont, FR); u256_to_le(b, &k); tr_absorb(t, b, 32);
}
/* common.rs:72-76 */
static void challenge_scalar(u256 *o, transcript_t *t, const suite_t *s) {
    uint8_t b[16]; tr_squeeze(t, b, 16); mont_from_le_bytes_mod_order(o, b, 16, FR);
}
/* common.rs:57-70: ceil((MODULUS_BIT_SIZE + 128)/8) bytes */
static void nonce_scalar(u256 *o, transcript_t *t, const suite_t *s) {
    uint8_t b[64]; size_t n = (size_t)(s->fr.bits + 128 + 7) / 8;
    tr_squeeze(t, b, n); mont_from_le_bytes_mod_order(o, b, n, FR);
}
/* common.rs:313-328; `t` is consumed (caller passes a clone) */
static void nonce(u256 *o, const u256 *sk_mont, transcript_t t, const suite_t *s) {
    transcript_t te = t; uint8_t skh[64];
    absorb_u8(&te, DS_NONCE_EXPAND); absorb_scalar(&te, sk_mont, s); tr_squeeze(&te, skh, 64);
    absorb_u8(&t, DS_NONCE); tr_absorb(&t, skh, 64);
    nonce_scalar(o, &t, s);
}
/* common.rs:270-280 */
static void challenge(u256 *o, const te_aff *const *pts, size_t n, transcript_t t, const suite_t *s) {
    absorb_u8(&t, DS_CHALLENGE);
    for (size_t i = 0; i < n; i++) absorb_point(&t, pts[i], s);
    challenge_scalar(o, &t, s);
}

typedef struct { te_aff in, out; } vrf_io;

/* common.rs:159-173: returns the transcript (ad absorbed) and the delinearisation fork */
static void transcript_base(transcript_t *t, transcript_t *delin, uint8_t scheme, const vrf_io *ios, size_t n,
                            const uint8_t *ad, size_t ad_len, const suite_t *s) {
    tr_new_mode(t, s->suite_id, s->suite_id_len, s->xof_shake);
    absorb_u8(t, scheme);
    absorb_u64(t, (uint64_t)n);                                    /* absorb_ios :377-383 */
    for (size_t i = 0; i < n; i++) { absorb_point(t, &ios[i].in, s); absorb_point(t, &ios[i].out, s); }
    absorb_u64(t, (uint64_t)ad_len); tr_absorb(t, ad, ad_len);
    *delin = *t; absorb_u8(delin, DS_DELINEARIZE);                 /* DelinearizeScalars::new :345-351 */
}
/* z_0 = 1, z_i = challenge_scalar(stream) :354-363 */
static void delin_take(u256 *zs, size_t n, transcript_t *delin, const suite_t *s) {
    for (size_t i = 0; i < n; i++) { if (i == 0) zs[0] = s->fr.r1; else challenge_scalar(&zs[i], delin, s); }
}
/* common.rs:181-202 + merge_ios :389-419 (fold and MSM branches give the same point) */
static void transcript_merged(transcript_t *t, vrf_io *merged, uint8_t scheme, const vrf_io *ios, size_t n,
                              const uint8_t *ad, size_t ad_len, const suite_t *s) {
    transcript_t delin; transcript_base(t, &delin, scheme, ios, n, ad, ad_len, s);
    if (n == 0) { memset(merged, 0, sizeof *merged); if (!s->sw_native) { merged->in.y = s->fq.r1; merged->out.y = s->fq.r1; } return; }
    if (n == 1) { *merged = ios[0]; return; }
    te_ext acc[2]; te_identity(&acc[0], s); te_identity(&acc[1], s);
    if (n < 16) {                                                  /* MSM_THRESHOLD, common.rs:397; fold :400-404 */
        for (size_t i = 0; i < n; i++) {
            u256 z, zp; if (i == 0) z = s->fr.r1; else challenge_scalar(&z, &delin, s);
            mont_from(&zp, &z, FR);
            te_ext a, b; te_smul(&a, &ios[i].in, &zp, s); te_smul(&b, &ios[i].out, &zp, s);
            te_add(&acc[0], &acc[0], &a, s); te_add(&acc[1], &acc[1], &b, s);
        }
    } else {                                                       /* two n-point msm_unchecked, common.rs:405-412 */
        te_aff *pi = (te_aff *)malloc(n * sizeof(te_aff)), *po = (te_aff *)malloc(n * sizeof(te_aff));
        u256 *zs = (u256 *)malloc(n * sizeof(u256));
        for (size_t i = 0; i < n; i++) {
            u256 z; if (i == 0) z = s->fr.r1; else challenge_scalar(&z, &delin, s);
            mont_from(&zs[i], &z, FR); pi[i] = ios[i].in; po[i] = ios[i].out;
        }
        orc_msm_pippenger(&acc[0], pi, zs, n, s); orc_msm_pippenger(&acc[1], po, zs, n, s);
        free(pi); free(po); free(zs);
    }
    te_aff norm[2]; te_batch_to_aff(norm, acc, 2, s);
    merged->in = norm[0]; merged->out = norm[1];
}

/* Point wire format of the batch entry points: compressed 32-byte (default) or, for the *_xy
 * wrappers at the end of this file, the C-ABI's 64-byte LE32(x)||LE32(y) -- what a caller of the
 * reference holds AFTER deserialisation (the reference's BatchVerifier takes typed points, so
 * decompression is outside `prepare`/`verify`; benches/thin.rs:46-90). */
static __thread int g_xy = 0;
#define PL ((size_t)s->pt_len)                 /* compressed point at the entry points: 32, or 33 for a short-Weierstrass suite */
#define PSZ ((size_t)(g_xy ? 64 : s->pt_len))
static int pt_dec(te_aff *o, const uint8_t *b, const suite_t *s) { return g_xy ? te_decode_xy(o, b, s) : te_decode(o, b, s); }
static int decode_ios(vrf_io *o, const uint8_t *b, size_t n, const suite_t *s) {
    for (size_t i = 0; i < n; i++) {
        if (pt_dec(&o[i].in, b + 2 * PSZ * i, s) || pt_dec(&o[i].out, b + 2 * PSZ * i + PSZ, s)) return ORC_INVALID_DATA;
    }
    return ORC_OK;
}
static int decode_scalar(u256 *o_mont, const uint8_t b[32], const suite_t *s) {
    u256 k; u256_from_le(&k, b);
    if (u256_cmp(&k, &s->fr.p) >= 0) return ORC_INVALID_DATA;
    mont_to(o_mont, &k, FR); return ORC_OK;
}
static void encode_scalar(uint8_t b[32], const u256 *k_mont, const suite_t *s) {
    u256 k; mont_from(&k, k_mont, FR); u256_to_le(b, &k);
}
static void smul_mont(te_ext *o, const te_aff *p, const u256 *k_mont, const suite_t *s) {
    u256 k; mont_from(&k, k_mont, FR); te_smul(o, p, &k, s);
}

/* ------------------------------------------------------------------ keys, io */

/* src/lib.rs:346-369 */
int orc_from_seed(int suite, const uint8_t seed[32], uint8_t sk_out[32], uint8_t pk_out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 sk0, sk; mont_from_le_bytes_mod_order(&sk0, seed, 32, FR);
    for (unsigned cnt = 0;; cnt++) {
        if (cnt > 255) return -1;
        transcript_t t; tr_new_mode(&t, s->suite_id, s->suite_id_len, s->xof_shake);
        tr_absorb(&t, seed, 32);
        if (cnt > 0) absorb_u8(&t, (uint8_t)cnt);
        nonce(&sk, &sk0, t, s);
        if (!u256_is_zero(&sk)) break;
    }
    encode_scalar(sk_out, &sk, s);
    te_ext P; smul_mont(&P, &s->G, &sk, s); te_aff pa; te_to_aff(&pa, &P, s); te_encode(pk_out, &pa, s);
    return 0;
}
/* src/lib.rs:331-334 */
int orc_sk_to_pk(int suite, const uint8_t sk[32], uint8_t pk_out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 k; if (decode_scalar(&k, sk, s)) return ORC_INVALID_DATA;
    te_ext P; smul_mont(&P, &s->G, &k, s); te_aff pa; te_to_aff(&pa, &P, s); te_encode(pk_out, &pa, s);
    return 0;
}
/* src/lib.rs:391-393 */
int orc_vrf_output(int suite, const uint8_t sk[32], const uint8_t input[32], uint8_t out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 k; te_aff in; if (decode_scalar(&k, sk, s) || te_decode(&in, input, s)) return ORC_INVALID_DATA;
    te_ext P; smul_mont(&P, &in, &k, s); te_aff pa; te_to_aff(&pa, &P, s); te_encode(out, &pa, s);
    return 0;
}
/* src/utils/common.rs:290-305 (mul_by_cofactor = false, src/lib.rs:247-249) */
int orc_point_to_hash(int suite, const uint8_t pt[32], uint8_t *out, size_t n) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    transcript_t t; tr_new_mode(&t, s->suite_id, s->suite_id_len, s->xof_shake);
    absorb_u8(&t, DS_POINT_TO_HASH);
    if (s->sw_codec) { te_aff p; if (te_decode(&p, pt, s)) return ORC_INVALID_DATA; absorb_point(&t, &p, s); }
    else tr_absorb(&t, pt, PL);
    tr_squeeze(&t, out, n);
    return 0;
}
/* the suite's own wire form of a point <-> the 32-byte twisted-Edwards form every other oracle entry point takes */
int orc_sw_encode(int suite, const uint8_t te32[32], uint8_t out33[33]) {
    const suite_t *s = orc_suite(suite); if (!s || !s->sw_codec) return -1;
    te_aff p; if (te_decode(&p, te32, s)) return ORC_INVALID_DATA;
    sw_encode(out33, &p, s); return 0;
}
int orc_sw_decode(int suite, const uint8_t in33[33], uint8_t te32[32]) {
    const suite_t *s = orc_suite(suite); if (!s || !s->sw_codec) return -1;
    te_aff p; if (sw_decode(&p, in33, s)) return ORC_INVALID_DATA;
    te_encode(te32, &p, s); return 0;
}

/* ------------------------------------------------------------------ hash to curve */

/* RFC 9380 5.3.1 expand_message_xmd with SHA-512, as instantiated by ark-ff
 * DefaultFieldHasher: Z_pad = len_per_base_elem zero bytes (48), not the hash
 * block size (probe-verified). */
static void expand_xmd(uint8_t *out, size_t len, const uint8_t *msg, size_t msg_len,
                       const uint8_t *dst, size_t dst_len, size_t zpad) {
    uint8_t b0[64], bi[64], z[128] = {0}; sha512_t h;
    size_t ell = (len + 63) / 64;
    uint8_t lib[2] = {(uint8_t)(len >> 8), (uint8_t)len}, zero = 0, dl = (uint8_t)dst_len;
    sha512_init(&h); sha512_update(&h, z, zpad); sha512_update(&h, msg, msg_len);
    sha512_update(&h, lib, 2); sha512_update(&h, &zero, 1); sha512_update(&h, dst, dst_len); sha512_update(&h, &dl, 1);
    sha512_final(&h, b0);
    for (size_t i = 1; i <= ell; i++) {
        uint8_t x[64], ib = (uint8_t)i;
        for (int j = 0; j < 64; j++) x[j] = (i == 1) ? b0[j] : (uint8_t)(b0[j] ^ bi[j]);
        sha512_init(&h); sha512_update(&h, x, 64); sha512_update(&h, &ib, 1);
        sha512_update(&h, dst, dst_len); sha512_update(&h, &dl, 1);
        sha512_final(&h, bi);
        size_t off = (i - 1) * 64, take = len - off < 64 ? len - off : 64;
        memcpy(out + off, bi, take);
    }
}

/* ark-ec Elligator2Map::map_to_curve for a TE curve with Montgomery model (J, K), Z non-square
 * (SURVEY.md A.6).  Output is a TE point (not yet cofactor-cleared). */
static void ell2_map(te_aff *o, const u256 *u, const suite_t *s) {
    u256 one = s->fq.r1, k_inv, j_on_k, k2_inv, den, x1, x2, gx1, gx2, t0, t1, xs, ys;
    mont_inv(&k_inv, &s->ell2_k, FQ); mont_mul(&j_on_k, &s->ell2_j, &k_inv, FQ);
    mont_sqr(&k2_inv, &k_inv, FQ);
    mont_sqr(&t0, u, FQ); mont_mul(&t0, &t0, &s->ell2_z, FQ); mont_add(&den, &one, &t0, FQ);
    if (u256_is_zero(&den)) den = one;
    mont_inv(&den, &den, FQ); mont_mul(&x1, &j_on_k, &den, FQ); mont_neg(&x1, &x1, FQ);
    /* gx1 = x1^3 + (J/K) x1^2 + x1/K^2 */
    mont_sqr(&t0, &x1, FQ); mont_mul(&t1, &t0, &x1, FQ); mont_mul(&t0, &t0, &j_on_k, FQ);
    mont_add(&gx1, &t1, &t0, FQ); mont_mul(&t0, &x1, &k2_inv, FQ); mont_add(&gx1, &gx1, &t0, FQ);
    mont_neg(&x2, &x1, FQ); mont_sub(&x2, &x2, &j_on_k, FQ);
    int want_odd;
    if (mont_is_square(&gx1, FQ)) { xs = x1; mont_sqrt(&ys, &gx1, FQ); want_odd = 1; }
    else {
        mont_sqr(&t0, &x2, FQ); mont_mul(&t1, &t0, &x2, FQ); mont_mul(&t0, &t0, &j_on_k, FQ);
        mont_add(&gx2, &t1, &t0, FQ); mont_mul(&t0, &x2, &k2_inv, FQ); mont_add(&gx2, &gx2, &t0, FQ);
        xs = x2; mont_sqrt(&ys, &gx2, FQ); want_odd = 0;
    }
    { u256 yp; mont_from(&yp, &ys, FQ); if ((int)(yp.l[0] & 1) != want_odd) mont_neg(&ys, &ys, FQ); }
    u256 sx, ty; mont_mul(&sx, &xs, &s->ell2_k, FQ); mont_mul(&ty, &ys, &s->ell2_k, FQ);
    /* Montgomery (s,t) -> TE (v,w) = (s/t, (s-1)/(s+1)); (0,1) if t(s+1) == 0 */
    u256 sp1; mont_add(&sp1, &sx, &one, FQ); mont_mul(&t0, &ty, &sp1, FQ);
    if (u256_is_zero(&t0)) { memset(&o->x, 0, sizeof o->x); o->y = one; return; }
    mont_inv(&t0, &t0, FQ);                     /* 1/(t (s+1)) */
    mont_mul(&t1, &t0, &sp1, FQ);               /* 1/t */
    mont_mul(&o->x, &sx, &t1, FQ);
    mont_mul(&t1, &t0, &ty, FQ);                /* 1/(s+1) */
    u256 sm1; mont_sub(&sm1, &sx, &one, FQ); mont_mul(&o->y, &sm1, &t1, FQ);
}

static void clear_cofactor(te_ext *p, const suite_t *s) {
    for (int c = s->cofactor; c > 1; c >>= 1) te_dbl(p, p, s);
}

/* src/utils/hash_to_curve.rs:66-100 / :34-57 */
int orc_hash_to_curve(int suite, const uint8_t *data, size_t n, uint8_t out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    if (s->h2c == ORC_H2C_ELL2) {
        uint8_t dst[64]; size_t dl = s->suite_id_len;
        memcpy(dst, s->suite_id, dl); dst[dl++] = DS_HASH_TO_CURVE;
        size_t L = (size_t)(s->fq.bits + 128 + 7) / 8; /* 48 */
        uint8_t ub[128];
        if (s->xof_shake == 1) {   /* XofFieldHasher, expand_message_xof (src/utils/hash_to_curve.rs:103-150): H(msg || I2OSP(len, 2) || DST || I2OSP(len(DST), 1)) */
            uint8_t lib[2] = {(uint8_t)((2 * L) >> 8), (uint8_t)(2 * L)}, dlb = (uint8_t)dl;
            const uint8_t *parts[4] = {data, lib, dst, &dlb}; size_t lens[4] = {n, 2, dl, 1};
            shake128(ub, 2 * L, parts, lens, 4);
        } else expand_xmd(ub, 2 * L, data, n, dst, dl, L);
        u256 u0, u1; mont_from_be_bytes_mod_order(&u0, ub, L, FQ); mont_from_be_bytes_mod_order(&u1, ub + L, L, FQ);
        te_aff q0, q1; ell2_map(&q0, &u0, s); ell2_map(&q1, &u1, s);
        te_ext e0, e1; te_from_aff(&e0, &q0, s); te_from_aff(&e1, &q1, s);
        te_add(&e0, &e0, &e1, s); clear_cofactor(&e0, s);
        te_aff r; te_to_aff(&r, &e0, s); te_encode(out, &r, s);
        return 0;
    }
    /* TAI */
    transcript_t prefix; tr_new_mode(&prefix, s->suite_id, s->suite_id_len, s->xof_shake);
    absorb_u8(&prefix, DS_HASH_TO_CURVE); absorb_u64(&prefix, (uint64_t)n); tr_absorb(&prefix, data, n);
    for (int ctr = 0; ctr <= 255; ctr++) {
        transcript_t t = prefix; absorb_u8(&t, (uint8_t)ctr);
        uint8_t buf[32]; tr_squeeze(&t, buf, 32);
        if (s->h2c == ORC_H2C_TAI_SW) {
            /* SWAffine::from_random_bytes on base_len = 32 bytes: Fp::from_random_bytes_with_flags::<SWFlags> copies them into
             * its 33-byte buffer, so the flag byte is always zero and the root is fixed: the LARGER one (pinned by the
             * alpha -> h entries of the reference's bandersnatch_sw vectors: the smaller root gives another point); the
             * bits above MODULUS_BIT_SIZE are cleared; x >= p -> next counter */
            if (s->fq.bits < 256) buf[31] &= (uint8_t)(0xff >> (256 - s->fq.bits));
            u256 x; u256_from_le(&x, buf);
            te_aff p; if (sw_from_x(&p, &x, 1, s)) continue;
            te_ext e; te_from_aff(&e, &p, s); clear_cofactor(&e, s);
            if (te_is_identity_ext(&e, s)) continue;
            te_aff r; te_to_aff(&r, &e, s); te_encode(out, &r, s);
            return 0;
        }
        /* TE Affine::from_random_bytes: top bit = x-sign flag, bits above MODULUS_BIT_SIZE cleared */
        int flag = buf[31] >> 7;
        buf[31] &= (uint8_t)(0xff >> (256 - s->fq.bits));
        uint8_t enc[32]; memcpy(enc, buf, 32);
        u256 y; u256_from_le(&y, enc); if (u256_cmp(&y, &s->fq.p) >= 0) continue;
        enc[31] |= (uint8_t)(flag << 7);
        te_aff p; if (te_decode(&p, enc, s)) continue;
        te_ext e; te_from_aff(&e, &p, s); clear_cofactor(&e, s);
        if (te_is_identity_ext(&e, s)) continue;
        te_aff r; te_to_aff(&r, &e, s); te_encode(out, &r, s);
        return 0;
    }
    return ORC_INVALID_DATA;
}

/* ------------------------------------------------------------------ Thin VRF */

static int io_has_identity(const vrf_io *ios, size_t n, const suite_t *s) {
    for (size_t i = 0; i < n; i++) if (te_is_identity_aff(&ios[i].in, s) || te_is_identity_aff(&ios[i].out, s)) return 1;
    return 0;
}
/* chain_ios (common.rs:231-240): prepend (G, pk) */
static vrf_io *chain_schnorr(const te_aff *pk, const vrf_io *ios, size_t n, const suite_t *s) {
    vrf_io *c = (vrf_io *)malloc((n + 1) * sizeof(vrf_io));
    c[0].in = s->G; c[0].out = *pk; if (n) memcpy(c + 1, ios, n * sizeof(vrf_io));
    return c;
}

/* src/thin.rs:111-129 */
int orc_thin_prove(int suite, const uint8_t sk_b[32], const uint8_t *ios_b, size_t n_ios,
                   const uint8_t *ad, size_t ad_len, uint8_t proof[64]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 sk; if (decode_scalar(&sk, sk_b, s)) return ORC_INVALID_DATA;
    vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    if (decode_ios(ios, ios_b, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    te_ext P; smul_mont(&P, &s->G, &sk, s); te_aff pk; te_to_aff(&pk, &P, s);
    vrf_io *ch = chain_schnorr(&pk, ios, n_ios, s);
    transcript_t t; vrf_io m; transcript_merged(&t, &m, DS_THIN, ch, n_ios + 1, ad, ad_len, s);
    u256 k, c, sres; nonce(&k, &sk, t, s);
    te_ext R; smul_mont(&R, &m.in, &k, s); te_aff r; te_to_aff(&r, &R, s);
    const te_aff *pts[1] = {&r}; challenge(&c, pts, 1, t, s);
    mont_mul(&sres, &c, &sk, FR); mont_add(&sres, &sres, &k, FR);
    te_encode(proof, &r, s); encode_scalar(proof + PL, &sres, s);
    free(ch); free(ios); return ORC_OK;
}

/* src/thin.rs:131-165 */
int orc_thin_verify(int suite, const uint8_t pk_b[32], const uint8_t *ios_b, size_t n_ios,
                    const uint8_t *ad, size_t ad_len, const uint8_t proof[64]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff pk, r; u256 sres, c, negc;
    vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    if (te_decode(&pk, pk_b, s) || decode_ios(ios, ios_b, n_ios, s) || te_decode(&r, proof, s) ||
        decode_scalar(&sres, proof + PL, s)) { free(ios); return ORC_INVALID_DATA; }
    if (te_is_identity_aff(&pk, s) || io_has_identity(ios, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    vrf_io *ch = chain_schnorr(&pk, ios, n_ios, s);
    transcript_t t; vrf_io m; transcript_merged(&t, &m, DS_THIN, ch, n_ios + 1, ad, ad_len, s);
    const te_aff *pts[1] = {&r}; challenge(&c, pts, 1, t, s);
    mont_neg(&negc, &c, FR);
    te_aff P2[2] = {m.in, m.out}; u256 S2[2]; mont_from(&S2[0], &sres, FR); mont_from(&S2[1], &negc, FR);
    te_ext lhs, re; orc_straus(&lhs, P2, S2, 2, 2, s); te_from_aff(&re, &r, s);
    int ok = te_eq_ext(&lhs, &re, s);
    free(ch); free(ios); return ok ? ORC_OK : ORC_VERIFICATION_FAILURE;
}

/* ------------------------------------------------------------------ Tiny VRF (src/tiny.rs)
 * proof = LE16(c) || LE32(s)  (CHALLENGE_LEN = 16, Proof::serialize_with_mode src/tiny.rs:60-78) */

/* src/tiny.rs:163-176 */
int orc_tiny_prove(int suite, const uint8_t sk_b[32], const uint8_t *ios_b, size_t n_ios,
                   const uint8_t *ad, size_t ad_len, uint8_t proof[48]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 sk; if (decode_scalar(&sk, sk_b, s)) return ORC_INVALID_DATA;
    vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    if (decode_ios(ios, ios_b, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    te_ext P; smul_mont(&P, &s->G, &sk, s); te_aff pk; te_to_aff(&pk, &P, s);
    vrf_io *ch = chain_schnorr(&pk, ios, n_ios, s);
    transcript_t t; vrf_io m; transcript_merged(&t, &m, DS_TINY, ch, n_ios + 1, ad, ad_len, s);
    u256 k, c, sres; nonce(&k, &sk, t, s);
    te_ext R; smul_mont(&R, &m.in, &k, s); te_aff r; te_to_aff(&r, &R, s);
    const te_aff *pts[1] = {&r}; challenge(&c, pts, 1, t, s);
    mont_mul(&sres, &c, &sk, FR); mont_add(&sres, &sres, &k, FR);
    uint8_t cb[32]; encode_scalar(cb, &c, s); memcpy(proof, cb, 16);       /* c < 2^128: its first 16 bytes */
    encode_scalar(proof + 16, &sres, s);
    free(ch); free(ios); return ORC_OK;
}

/* src/tiny.rs:178-214 */
int orc_tiny_verify(int suite, const uint8_t pk_b[32], const uint8_t *ios_b, size_t n_ios,
                    const uint8_t *ad, size_t ad_len, const uint8_t proof[48]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff pk; u256 sres, c, c_exp, negc;
    vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    uint8_t cb[32]; memset(cb, 0, 32); memcpy(cb, proof, 16);              /* from_le_bytes_mod_order of 16 bytes */
    if (te_decode(&pk, pk_b, s) || decode_ios(ios, ios_b, n_ios, s) || decode_scalar(&c, cb, s) ||
        decode_scalar(&sres, proof + 16, s)) { free(ios); return ORC_INVALID_DATA; }
    if (te_is_identity_aff(&pk, s) || io_has_identity(ios, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    vrf_io *ch = chain_schnorr(&pk, ios, n_ios, s);
    transcript_t t; vrf_io m; transcript_merged(&t, &m, DS_TINY, ch, n_ios + 1, ad, ad_len, s);
    mont_neg(&negc, &c, FR);                                               /* R = s I_m - c O_m  (short_msm, :207) */
    te_aff P2[2] = {m.in, m.out}; u256 S2[2]; mont_from(&S2[0], &sres, FR); mont_from(&S2[1], &negc, FR);
    te_ext R; orc_straus(&R, P2, S2, 2, 2, s); te_aff r; te_to_aff(&r, &R, s);
    const te_aff *pts[1] = {&r}; challenge(&c_exp, pts, 1, t, s);
    int ok = u256_cmp(&c_exp, &c) == 0;
    free(ch); free(ios); return ok ? ORC_OK : ORC_VERIFICATION_FAILURE;
}

/* src/thin.rs:209-226 + :257-318: builds the (sum(2+2M_j)+1)-term MSM of the batch.
 * bases_xy: n_terms x 64 (LE32 x || LE32 y), scalars: n_terms x 32.  Returns status;
 * when status == ORC_OK the terms are valid.  n_terms_out may be NULL. */
int orc_thin_batch_terms(int suite, size_t n, const uint8_t *pks, const uint8_t *ios_b, const uint32_t *io_counts,
                         const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                         uint8_t *bases_xy, uint8_t *scalars, size_t *n_terms_out) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    if (n_terms_out) *n_terms_out = 0;
    if (n == 0) return ORC_OK;                                     /* thin.rs:262-264 */
    size_t tot_io = 0, max_io = 0;
    for (size_t j = 0; j < n; j++) { tot_io += io_counts[j]; if (io_counts[j] > max_io) max_io = io_counts[j]; }
    te_aff *pk = (te_aff *)malloc(n * sizeof(te_aff)), *R = (te_aff *)malloc(n * sizeof(te_aff));
    vrf_io *ios = (vrf_io *)malloc((tot_io + 1) * sizeof(vrf_io));
    u256 *c = (u256 *)malloc(n * sizeof(u256)), *sv = (u256 *)malloc(n * sizeof(u256));
    u256 *zs = (u256 *)malloc((tot_io + n) * sizeof(u256));
    vrf_io *ch = (vrf_io *)malloc((max_io + 1) * sizeof(vrf_io));
    int st = ORC_OK;
    /* decode everything first (the reference's typed API has done so before push) */
    for (size_t j = 0; j < n && !st; j++)
        if (pt_dec(&pk[j], pks + PSZ * j, s) || pt_dec(&R[j], proofs + (PSZ + 32) * j, s) ||
            decode_scalar(&sv[j], proofs + (PSZ + 32) * j + PSZ, s)) st = ORC_INVALID_DATA;
    if (!st && decode_ios(ios, ios_b, tot_io, s)) st = ORC_INVALID_DATA;
    if (!st) {
        /* prepare: thin.rs:209-226 */
        size_t io_off = 0, ad_off = 0, z_off = 0;
        for (size_t j = 0; j < n; j++) {
            size_t m = io_counts[j];
            ch[0].in = s->G; ch[0].out = pk[j]; if (m) memcpy(ch + 1, ios + io_off, m * sizeof(vrf_io));
            transcript_t t, delin; transcript_base(&t, &delin, DS_THIN, ch, m + 1, ads + ad_off, ad_lens[j], s);
            delin_take(zs + z_off, m + 1, &delin, s);
            const te_aff *pts[1] = {&R[j]}; challenge(&c[j], pts, 1, t, s);
            io_off += m; ad_off += ad_lens[j]; z_off += m + 1;
        }
        /* verify: identity checks thin.rs:266-271 */
        for (size_t j = 0; j < n && !st; j++) if (te_is_identity_aff(&pk[j], s)) st = ORC_INVALID_DATA;
        if (!st && io_has_identity(ios, tot_io, s)) st = ORC_INVALID_DATA;
    }
    if (!st) {
        transcript_t tw; tr_new_mode(&tw, s->suite_id, s->suite_id_len, s->xof_shake); absorb_u8(&tw, DS_BATCH_VERIFY); /* :274-279 */
        for (size_t j = 0; j < n; j++) { absorb_scalar(&tw, &c[j], s); absorb_scalar(&tw, &sv[j], s); }
        size_t k = 0, io_off = 0, z_off = 0; u256 g = {{0, 0, 0, 0}};
        for (size_t j = 0; j < n; j++) {                           /* :287-313 */
            size_t m = io_counts[j];
            u256 w, wc, ws, t0;
            challenge_scalar(&w, &tw, s);
            mont_mul(&wc, &w, &c[j], FR); mont_mul(&ws, &w, &sv[j], FR);
            te_encode_xy(bases_xy + 64 * k, &R[j], s); encode_scalar(scalars + 32 * k, &w, s); k++;
            mont_mul(&t0, &wc, &zs[z_off], FR);
            te_encode_xy(bases_xy + 64 * k, &pk[j], s); encode_scalar(scalars + 32 * k, &t0, s); k++;
            mont_mul(&t0, &ws, &zs[z_off], FR); mont_sub(&g, &g, &t0, FR);
            for (size_t i = 0; i < m; i++) {
                mont_mul(&t0, &wc, &zs[z_off + i + 1], FR);
                te_encode_xy(bases_xy + 64 * k, &ios[io_off + i].out, s); encode_scalar(scalars + 32 * k, &t0, s); k++;
                mont_mul(&t0, &ws, &zs[z_off + i + 1], FR); mont_neg(&t0, &t0, FR);
                te_encode_xy(bases_xy + 64 * k, &ios[io_off + i].in, s); encode_scalar(scalars + 32 * k, &t0, s); k++;
            }
            io_off += m; z_off += m + 1;
        }
        te_encode_xy(bases_xy + 64 * k, &s->G, s); encode_scalar(scalars + 32 * k, &g, s); k++; /* :316-317 */
        if (n_terms_out) *n_terms_out = k;
    }
    free(pk); free(R); free(ios); free(c); free(sv); free(zs); free(ch);
    return st;
}

int orc_msm(int suite, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t out_xy[64], int algo);

/* src/thin.rs:257-325 */
int orc_thin_batch_verify(int suite, size_t n, const uint8_t *pks, const uint8_t *ios_b, const uint32_t *io_counts,
                          const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    if (n == 0) return ORC_OK;
    size_t tot_io = 0; for (size_t j = 0; j < n; j++) tot_io += io_counts[j];
    size_t cap = 2 * n + 2 * tot_io + 1, k = 0;
    uint8_t *bases = (uint8_t *)malloc(cap * 64), *sc = (uint8_t *)malloc(cap * 32), out[64];
    int st = orc_thin_batch_terms(suite, n, pks, ios_b, io_counts, ads, ad_lens, proofs, bases, sc, &k);
    if (!st) {
        orc_msm(suite, k, bases, sc, out, 1);
        te_aff r; te_decode_xy(&r, out, s);
        st = te_is_identity_aff(&r, s) ? ORC_OK : ORC_VERIFICATION_FAILURE;  /* :319-322 */
    }
    free(bases); free(sc); return st;
}

/* ------------------------------------------------------------------ Pedersen VRF */

/* src/pedersen.rs:136-186 */
int orc_pedersen_prove(int suite, const uint8_t sk_b[32], const uint8_t *ios_b, size_t n_ios,
                       const uint8_t *ad, size_t ad_len, uint8_t proof[160], uint8_t blinding_out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    u256 sk; if (decode_scalar(&sk, sk_b, s)) return ORC_INVALID_DATA;
    vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    if (decode_ios(ios, ios_b, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    transcript_t t; vrf_io io; transcript_merged(&t, &io, DS_PEDERSEN, ios, n_ios, ad, ad_len, s);
    u256 b, k, kb, c, sres, sbres;
    { transcript_t tb = t; absorb_u8(&tb, DS_PEDERSEN_BLINDING); nonce(&b, &sk, tb, s); }   /* :51-54,145 */
    te_ext PK, BB, YB; smul_mont(&PK, &s->G, &sk, s); smul_mont(&BB, &s->B, &b, s); te_add(&YB, &PK, &BB, s);
    te_aff yb; te_to_aff(&yb, &YB, s);
    absorb_point(&t, &yb, s);                                       /* :152 */
    nonce(&k, &sk, t, s); nonce(&kb, &b, t, s);                     /* :155-156 */
    te_ext KG, KBB, RO[2]; smul_mont(&KG, &s->G, &k, s); smul_mont(&KBB, &s->B, &kb, s); te_add(&RO[0], &KG, &KBB, s);
    smul_mont(&RO[1], &io.in, &k, s);
    te_aff ro[2]; te_batch_to_aff(ro, RO, 2, s);
    const te_aff *pts[2] = {&ro[0], &ro[1]}; challenge(&c, pts, 2, t, s);
    mont_mul(&sres, &c, &sk, FR); mont_add(&sres, &sres, &k, FR);
    mont_mul(&sbres, &c, &b, FR); mont_add(&sbres, &sbres, &kb, FR);
    te_encode(proof, &yb, s); te_encode(proof + PL, &ro[0], s); te_encode(proof + 2 * PL, &ro[1], s);
    encode_scalar(proof + 3 * PL, &sres, s); encode_scalar(proof + 3 * PL + 32, &sbres, s);
    if (blinding_out) encode_scalar(blinding_out, &b, s);
    free(ios); return ORC_OK;
}

typedef struct { te_aff yb, r, ok; u256 s, sb; } ped_proof;
static int decode_ped(ped_proof *p, const uint8_t *b, const suite_t *s) {
    if (pt_dec(&p->yb, b, s) || pt_dec(&p->r, b + PSZ, s) || pt_dec(&p->ok, b + 2 * PSZ, s) ||
        decode_scalar(&p->s, b + 3 * PSZ, s) || decode_scalar(&p->sb, b + 3 * PSZ + 32, s)) return ORC_INVALID_DATA;
    return ORC_OK;
}

/* src/pedersen.rs:188-249 */
int orc_pedersen_verify(int suite, const uint8_t *ios_b, size_t n_ios, const uint8_t *ad, size_t ad_len,
                        const uint8_t proof[160]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    ped_proof p; vrf_io *ios = (vrf_io *)malloc((n_ios + 1) * sizeof(vrf_io));
    if (decode_ped(&p, proof, s) || decode_ios(ios, ios_b, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    if (te_is_identity_aff(&p.yb, s) || io_has_identity(ios, n_ios, s)) { free(ios); return ORC_INVALID_DATA; }
    transcript_t t; vrf_io io; transcript_merged(&t, &io, DS_PEDERSEN, ios, n_ios, ad, ad_len, s);
    absorb_point(&t, &p.yb, s);
    u256 c, negc; const te_aff *pts[2] = {&p.r, &p.ok}; challenge(&c, pts, 2, t, s);
    mont_neg(&negc, &c, FR);
    int st = ORC_OK;
    { te_aff P2[2] = {io.in, io.out}; u256 S2[2]; mont_from(&S2[0], &p.s, FR); mont_from(&S2[1], &negc, FR);
      te_ext lhs, e; orc_straus(&lhs, P2, S2, 2, 2, s); te_from_aff(&e, &p.ok, s);
      if (!te_eq_ext(&lhs, &e, s)) st = ORC_VERIFICATION_FAILURE; }
    if (!st) {
      te_aff P3[3] = {s->G, s->B, p.yb}; u256 S3[3];
      mont_from(&S3[0], &p.s, FR); mont_from(&S3[1], &p.sb, FR); mont_from(&S3[2], &negc, FR);
      te_ext lhs, e; orc_straus(&lhs, P3, S3, 3, 2, s); te_from_aff(&e, &p.r, s);
      if (!te_eq_ext(&lhs, &e, s)) st = ORC_VERIFICATION_FAILURE; }
    free(ios); return st;
}

/* src/pedersen.rs:276-293 + :341-418: the (5N+2)-term MSM */
int orc_pedersen_batch_terms(int suite, size_t n, const uint8_t *ios_b, const uint32_t *io_counts,
                             const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                             uint8_t *bases_xy, uint8_t *scalars, size_t *n_terms_out) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    if (n_terms_out) *n_terms_out = 0;
    if (n == 0) return ORC_OK;                                     /* :343-345 */
    size_t tot_io = 0; for (size_t j = 0; j < n; j++) tot_io += io_counts[j];
    ped_proof *pp = (ped_proof *)malloc(n * sizeof(ped_proof));
    vrf_io *ios = (vrf_io *)malloc((tot_io + 1) * sizeof(vrf_io)), *merged = (vrf_io *)malloc(n * sizeof(vrf_io));
    u256 *c = (u256 *)malloc(n * sizeof(u256));
    int st = ORC_OK, any_io_identity = 0;
    for (size_t j = 0; j < n && !st; j++) if (decode_ped(&pp[j], proofs + (3 * PSZ + 64) * j, s)) st = ORC_INVALID_DATA;
    if (!st && decode_ios(ios, ios_b, tot_io, s)) st = ORC_INVALID_DATA;
    if (!st) {
        size_t io_off = 0, ad_off = 0;
        for (size_t j = 0; j < n; j++) {                           /* BatchItem::new :276-293 */
            size_t m = io_counts[j];
            if (io_has_identity(ios + io_off, m, s)) any_io_identity = 1;
            transcript_t t; transcript_merged(&t, &merged[j], DS_PEDERSEN, ios + io_off, m, ads + ad_off, ad_lens[j], s);
            absorb_point(&t, &pp[j].yb, s);
            const te_aff *pts[2] = {&pp[j].r, &pp[j].ok}; challenge(&c[j], pts, 2, t, s);
            io_off += m; ad_off += ad_lens[j];
        }
        for (size_t j = 0; j < n; j++) if (te_is_identity_aff(&pp[j].yb, s)) st = ORC_INVALID_DATA; /* :348-353 */
        if (any_io_identity) st = ORC_INVALID_DATA;
    }
    if (!st) {
        transcript_t tw; tr_new_mode(&tw, s->suite_id, s->suite_id_len, s->xof_shake); absorb_u8(&tw, DS_BATCH_VERIFY); /* :361-367 */
        for (size_t j = 0; j < n; j++) { absorb_scalar(&tw, &c[j], s); absorb_scalar(&tw, &pp[j].s, s); absorb_scalar(&tw, &pp[j].sb, s); }
        u256 g = {{0, 0, 0, 0}}, bsc = {{0, 0, 0, 0}}; size_t k = 0;
        for (size_t j = 0; j < n; j++) {
            uint8_t buf[32]; tr_squeeze(&tw, buf, 32);             /* :373-381 */
            u256 tt, uu, t0;
            mont_from_le_bytes_mod_order(&tt, buf, 16, FR); mont_from_le_bytes_mod_order(&uu, buf + 16, 16, FR);
            mont_mul(&t0, &tt, &c[j], FR);
            te_encode_xy(bases_xy + 64 * k, &merged[j].out, s); encode_scalar(scalars + 32 * k, &t0, s); k++;
            te_encode_xy(bases_xy + 64 * k, &pp[j].ok, s); encode_scalar(scalars + 32 * k, &tt, s); k++;
            mont_mul(&t0, &tt, &pp[j].s, FR); mont_neg(&t0, &t0, FR);
            te_encode_xy(bases_xy + 64 * k, &merged[j].in, s); encode_scalar(scalars + 32 * k, &t0, s); k++;
            mont_mul(&t0, &uu, &c[j], FR);
            te_encode_xy(bases_xy + 64 * k, &pp[j].yb, s); encode_scalar(scalars + 32 * k, &t0, s); k++;
            te_encode_xy(bases_xy + 64 * k, &pp[j].r, s); encode_scalar(scalars + 32 * k, &uu, s); k++;
            mont_mul(&t0, &uu, &pp[j].s, FR); mont_add(&g, &g, &t0, FR);
            mont_mul(&t0, &uu, &pp[j].sb, FR); mont_add(&bsc, &bsc, &t0, FR);
        }
        mont_neg(&g, &g, FR); mont_neg(&bsc, &bsc, FR);           /* :412-418 */
        te_encode_xy(bases_xy + 64 * k, &s->G, s); encode_scalar(scalars + 32 * k, &g, s); k++;
        te_encode_xy(bases_xy + 64 * k, &s->B, s); encode_scalar(scalars + 32 * k, &bsc, s); k++;
        if (n_terms_out) *n_terms_out = k;
    }
    free(pp); free(ios); free(merged); free(c);
    return st;
}

/* src/pedersen.rs:341-426 */
int orc_pedersen_batch_verify(int suite, size_t n, const uint8_t *ios_b, const uint32_t *io_counts,
                              const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    if (n == 0) return ORC_OK;
    size_t cap = 5 * n + 2, k = 0;
    uint8_t *bases = (uint8_t *)malloc(cap * 64), *sc = (uint8_t *)malloc(cap * 32), out[64];
    int st = orc_pedersen_batch_terms(suite, n, ios_b, io_counts, ads, ad_lens, proofs, bases, sc, &k);
    if (!st) {
        orc_msm(suite, k, bases, sc, out, 1);
        te_aff r; te_decode_xy(&r, out, s);
        st = te_is_identity_aff(&r, s) ? ORC_OK : ORC_VERIFICATION_FAILURE;
    }
    free(bases); free(sc); return st;
}

/* Same verifiers on the C-ABI layouts of include/avrf.h (points as 64-byte xy): used as the
 * CPU baseline in bench.py, where -- as in benches/thin.rs:46-90 -- deserialisation is not timed. */
int orc_thin_batch_verify_xy(int suite, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                             const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
    g_xy = 1; int st = orc_thin_batch_verify(suite, n, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs); g_xy = 0; return st;
}
int orc_pedersen_batch_verify_xy(int suite, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                                 const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
    g_xy = 1; int st = orc_pedersen_batch_verify(suite, n, ios_xy, io_counts, ads, ad_lens, proofs); g_xy = 0; return st;
}
/* the MSM terms of the two batch verifiers on the same xy layouts (full-size GPU parity tests) */
int orc_thin_batch_terms_xy(int suite, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                            const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                            uint8_t *bases_xy, uint8_t *scalars, size_t *n_terms_out) {
    g_xy = 1; int st = orc_thin_batch_terms(suite, n, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs, bases_xy, scalars, n_terms_out); g_xy = 0; return st;
}
int orc_pedersen_batch_terms_xy(int suite, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                                const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                                uint8_t *bases_xy, uint8_t *scalars, size_t *n_terms_out) {
    g_xy = 1; int st = orc_pedersen_batch_terms(suite, n, ios_xy, io_counts, ads, ad_lens, proofs, bases_xy, scalars, n_terms_out); g_xy = 0; return st;
}

/* ------------------------------------------------------------------ raw MSM / codec helpers */

/* `msm_unchecked` on canonical encodings; algo 0 = naive, 1 = Pippenger, 2 = Straus(w=2, n<=5) */
int orc_msm(int suite, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t out_xy[64], int algo) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff *b = (te_aff *)malloc((n + 1) * sizeof(te_aff)); u256 *k = (u256 *)malloc((n + 1) * sizeof(u256));
    int st = ORC_OK;
    for (size_t i = 0; i < n && !st; i++) {
        if (te_decode_xy(&b[i], bases_xy + 64 * i, s)) st = ORC_INVALID_DATA;
        u256_from_le(&k[i], scalars + 32 * i);
        if (u256_cmp(&k[i], &s->fr.p) >= 0) st = ORC_INVALID_DATA;
    }
    if (!st) {
        te_ext r;
        if (algo == 0) orc_msm_naive(&r, b, k, n, s);
        else if (algo == 2) orc_straus(&r, b, k, n, 2, s);
        else orc_msm_pippenger(&r, b, k, n, s);
        te_aff ra; te_to_aff(&ra, &r, s); te_encode_xy(out_xy, &ra, s);
    }
    free(b); free(k); return st;
}

int orc_point_decompress(int suite, const uint8_t in[32], uint8_t out_xy[64], int validate) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff p; if (te_decode(&p, in, s)) return ORC_INVALID_DATA;
    if (validate && (!te_on_curve(&p, s) || !te_in_subgroup(&p, s))) return ORC_INVALID_DATA;
    te_encode_xy(out_xy, &p, s); return ORC_OK;
}
int orc_point_compress(int suite, const uint8_t in_xy[64], uint8_t out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff p; if (te_decode_xy(&p, in_xy, s)) return ORC_INVALID_DATA;
    te_encode(out, &p, s); return ORC_OK;
}
/* suite constants, compressed: which = 0 G, 1 BLINDING_BASE, 2 ACCUMULATOR_BASE, 3 PADDING */
int orc_suite_point(int suite, int which, uint8_t out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    const te_aff *p = which == 0 ? &s->G : which == 1 ? &s->B : which == 2 ? &s->ACC : &s->PAD;
    te_encode(out, p, s); return 0;
}
/* k*P on compressed encodings (test helper) */
int orc_smul(int suite, const uint8_t k_b[32], const uint8_t pt[32], uint8_t out[32]) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    te_aff p; u256 k; if (te_decode(&p, pt, s)) return ORC_INVALID_DATA;
    u256_from_le(&k, k_b);
    te_ext r; te_smul(&r, &p, &k, s); te_aff ra; te_to_aff(&ra, &r, s); te_encode(out, &ra, s);
    return 0;
}
void orc_sha512(const uint8_t *d, size_t n, uint8_t out[64]) {
    sha512_t h; sha512_init(&h); sha512_update(&h, d, n); sha512_final(&h, out);
}

/* ------------------------------------------------------------------ synthetic batches
 * Deterministic benchmark/test inputs derived from the item index j and a 32-byte run seed, through
 * the suite's own hash functions (recipe modelled on benches/thin.rs:46-63, README.md:161-182):
 *   sk_j    = Secret::from_seed(run_seed XOR (LE64(j) || 0...))        (src/lib.rs:346-369)
 *   input_j = Input::new("avrf-bench-input" || LE64(j))                (src/lib.rs:503-505)
 *   ad_j    = "ad-<j>"
 * Outputs use the C-ABI layouts of include/avrf.h ("xy" = LE32(x)||LE32(y)).
 * kind 0: thin   -> pks_xy n*64, ios_xy n*128, proofs n*96  (R_xy || s)
 * kind 1: pedersen -> ios_xy n*128, proofs n*256 (Yb_xy || R_xy || Ok_xy || s || sb), pks_xy unused (may be NULL)
 * sks (n*32) is always written.  ads: concatenated, ad_lens[n]; caller provides >= 16 bytes per item. */
int orc_gen_batch(int suite, int kind, const uint8_t run_seed[32], uint64_t start, size_t count,
                  uint8_t *sks, uint8_t *pks_xy, uint8_t *ios_xy, uint8_t *ads, uint32_t *ad_lens, uint8_t *proofs) {
    const suite_t *s = orc_suite(suite); if (!s) return -1;
    size_t ad_off = 0;
    for (size_t q = 0; q < count; q++) {
        uint64_t j = start + q;
        uint8_t seed[32], sk[32], pk[33], msg[24], in_c[33], out_c[33], io_c[66];
        memcpy(seed, run_seed, 32);
        for (int i = 0; i < 8; i++) seed[i] ^= (uint8_t)(j >> (8 * i));
        if (orc_from_seed(suite, seed, sk, pk)) return -1;
        memcpy(msg, "avrf-bench-input", 16);
        for (int i = 0; i < 8; i++) msg[16 + i] = (uint8_t)(j >> (8 * i));
        if (orc_hash_to_curve(suite, msg, 24, in_c)) return -1;
        if (orc_vrf_output(suite, sk, in_c, out_c)) return -1;
        memcpy(io_c, in_c, PL); memcpy(io_c + PL, out_c, PL);
        /* "ad-<j>" */
        char adb[24]; int al = 0; { char tmp[24]; int tl = 0; uint64_t v = j; do { tmp[tl++] = (char)('0' + v % 10); v /= 10; } while (v);
            adb[al++] = 'a'; adb[al++] = 'd'; adb[al++] = '-'; while (tl) adb[al++] = tmp[--tl]; }
        memcpy(ads + ad_off, adb, (size_t)al); ad_lens[q] = (uint32_t)al; ad_off += (size_t)al;
        memcpy(sks + 32 * q, sk, 32);
        te_aff p;
        te_decode(&p, in_c, s); te_encode_xy(ios_xy + 128 * q, &p, s);
        te_decode(&p, out_c, s); te_encode_xy(ios_xy + 128 * q + 64, &p, s);
        if (kind == 0) {
            uint8_t pr[65];
            if (orc_thin_prove(suite, sk, io_c, 1, (const uint8_t *)adb, (size_t)al, pr)) return -1;
            te_decode(&p, pk, s); te_encode_xy(pks_xy + 64 * q, &p, s);
            te_decode(&p, pr, s); te_encode_xy(proofs + 96 * q, &p, s);
            memcpy(proofs + 96 * q + 64, pr + PL, 32);
        } else {
            uint8_t pr[163];
            if (orc_pedersen_prove(suite, sk, io_c, 1, (const uint8_t *)adb, (size_t)al, pr, NULL)) return -1;
            if (pks_xy) { te_decode(&p, pk, s); te_encode_xy(pks_xy + 64 * q, &p, s); }
            for (int k = 0; k < 3; k++) { te_decode(&p, pr + PL * k, s); te_encode_xy(proofs + 256 * q + 64 * k, &p, s); }
            memcpy(proofs + 256 * q + 192, pr + 3 * PL, 64);
        }
    }
    return 0;
}
