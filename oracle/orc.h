/*
 * oracle/orc.h -- CPU restatement of the ark-vrf hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle: a plain-C restatement of the reference's
 * algorithms for the path BASELINE.json names (Thin / Pedersen VRF prove, verify
 * and batch verification over Bandersnatch and Baby-JubJub).  It is imported only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
 * checker.  The product (ark_vrf_amd/, libavrf.so) never links or calls it.
 *
 * The arithmetic the reference delegates to un-vendored crates (arkworks 0.6:
 * ark-ff / ark-ec / ark-serialize; sha2 0.10) is restated from the published
 * algorithms (Montgomery CIOS, twisted-Edwards extended coordinates, FIPS 180-4
 * SHA-512, RFC 9380 expand_message_xmd + Elligator2) and PINNED against the
 * reference's own known-answer vectors (tests/golden/ JSON files, copied from
 * /root/reference/data/vectors): sk->pk, alpha->h, gamma, beta, thin proofs,
 * pedersen proofs, for both suites.  See tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } u256;

/* Montgomery context for an odd modulus < 2^256 (arkworks Fp<MontBackend,4>) */
typedef struct {
    u256 p;        /* modulus */
    uint64_t ninv; /* -p^-1 mod 2^64 */
    u256 r1;       /* R mod p  (Montgomery one) */
    u256 r2;       /* R^2 mod p */
    int bits;      /* MODULUS_BIT_SIZE */
    /* Tonelli-Shanks data */
    int two_adicity;
    u256 t_odd;       /* (p-1)/2^s */
    u256 t_minus1_half; /* (t-1)/2 */
    u256 root_of_unity; /* g^t, Montgomery form, g a non-residue */
    u256 pm1_half;    /* (p-1)/2 plain integer */
} mont_t;

typedef struct { u256 x, y; } te_aff;        /* Montgomery-form coordinates */
typedef struct { u256 x, y, t, z; } te_ext;  /* extended twisted Edwards */

enum { ORC_SUITE_BANDERSNATCH = 0, ORC_SUITE_BABYJUBJUB = 1, ORC_SUITE_JUBJUB = 2, ORC_SUITE_ED25519 = 3, ORC_SUITE_BANDERSNATCH_SW = 4, ORC_SUITE_BANDERSNATCH_SHAKE128 = 5, ORC_SUITE_TESTING_SHA256 = 6, ORC_SUITE_SECP256R1 = 7 };
enum { ORC_H2C_ELL2 = 0, ORC_H2C_TAI = 1, ORC_H2C_TAI_SW = 2 };

typedef struct {
    int id;
    const char *suite_id;   /* S::SUITE_ID */
    size_t suite_id_len;
    mont_t fq;              /* base field */
    mont_t fr;              /* scalar field (prime-order subgroup) */
    u256 a, d;              /* TE coefficients, Montgomery form */
    int a_is_minus5;        /* informational */
    int cofactor;
    int h2c;
    te_aff G;               /* S::generator() */
    te_aff B;               /* PedersenSuite::BLINDING_BASE */
    te_aff ACC;             /* RingSuite::ACCUMULATOR_BASE */
    te_aff PAD;             /* RingSuite::PADDING */
    /* Elligator2 (Montgomery model) constants, Montgomery form */
    u256 ell2_j, ell2_k, ell2_z;
    /* short-Weierstrass presentation of the same curve (src/suites/bandersnatch_sw.rs, src/utils/te_sw_map.rs): the suite's
     * Affine type is SWAffine, so every point that is serialised -- into a transcript, a proof, a hash -- takes the 33-byte
     * ark-serialize SW form; the group arithmetic stays in the twisted-Edwards model through the maps. */
    int xof_shake;          /* Suite::Transcript = Shake128Transcript (src/suites/bandersnatch_shake128.rs) */
    int sw_codec;
    /* a genuinely short-Weierstrass suite (src/suites/secp256r1.rs:49-70: NIST P-256, y^2 = x^3 - 3x + b, cofactor 1): no
     * twisted-Edwards model exists, so the te_* group functions below compute in JACOBIAN coordinates for it (te_ext.x/y/z =
     * X/Y/Z, identity Z = 0; te_aff = the SW affine point, (0, 0) = infinity) and the compressed form of a point is the
     * 33-byte ark-serialize SW encoding.  Every protocol function above the group law is shared with the other suites. */
    int sw_native;
    int pt_len;             /* bytes of a compressed point at the oracle's entry points: 32, or 33 for sw_native */
    u256 mont_b, mont_a3, mont_binv, sw_a, sw_b;   /* Montgomery-model B, A/3, 1/B; SW coefficients; all in Montgomery form */
} suite_t;

/* status codes mirror ark_vrf::Error (src/lib.rs:135-147) */
enum { ORC_OK = 0, ORC_VERIFICATION_FAILURE = 1, ORC_INVALID_DATA = 2,
       ORC_RING_CAPACITY_EXCEEDED = 3, ORC_SRS_LOOKUP_FAILED = 4 };

/* ---- bigint / field (orc_field.c) ---- */
int  u256_cmp(const u256 *a, const u256 *b);
int  u256_is_zero(const u256 *a);
void u256_from_le(u256 *o, const uint8_t b[32]);
void u256_to_le(uint8_t b[32], const u256 *a);
int  u256_from_dec(u256 *o, const char *s);
void mont_init(mont_t *m, const u256 *p);
void mont_mul(u256 *o, const u256 *a, const u256 *b, const mont_t *m);
void mont_sqr(u256 *o, const u256 *a, const mont_t *m);
void mont_add(u256 *o, const u256 *a, const u256 *b, const mont_t *m);
void mont_sub(u256 *o, const u256 *a, const u256 *b, const mont_t *m);
void mont_neg(u256 *o, const u256 *a, const mont_t *m);
void mont_to(u256 *o, const u256 *a, const mont_t *m);    /* plain -> Montgomery */
void mont_from(u256 *o, const u256 *a, const mont_t *m);  /* Montgomery -> plain */
void mont_pow(u256 *o, const u256 *a, const u256 *e, const mont_t *m);
void mont_inv(u256 *o, const u256 *a, const mont_t *m);
int  mont_sqrt(u256 *o, const u256 *a, const mont_t *m);  /* 1 if square */
int  mont_is_square(const u256 *a, const mont_t *m);
/* ark_ff PrimeField::from_le_bytes_mod_order / from_be_bytes_mod_order */
void mont_from_le_bytes_mod_order(u256 *o, const uint8_t *b, size_t n, const mont_t *m);
void mont_from_be_bytes_mod_order(u256 *o, const uint8_t *b, size_t n, const mont_t *m);

/* ---- SHA-512 (orc_sha512.c) ---- */
typedef struct { uint64_t h[8]; uint8_t buf[128]; uint64_t len; } sha512_t;
void sha512_init(sha512_t *c);
void sha512_update(sha512_t *c, const void *data, size_t n);
void sha512_final(const sha512_t *c, uint8_t out[64]); /* does not mutate c */

/* ---- transcript (orc_transcript.c): HashTranscript<Sha512>, src/utils/transcript.rs:103-293 */
typedef struct {
    sha512_t h; int squeezing;
    uint8_t seed[64], block[64]; uint64_t counter; size_t off;
    /* XofTranscript<Shake128> (src/utils/transcript.rs:292-293; suites with xof_shake): the sponge itself */
    int shake; uint64_t ks[25]; size_t kpos;   /* shake: 0 SHA-512 counter mode, 1 SHAKE128 sponge, 2 SHA-256 counter mode */
    uint32_t h2[8]; uint8_t buf2[64]; uint64_t len2;   /* HashTranscript<Sha256> (src/suites/testing.rs) */
} transcript_t;
void tr_new(transcript_t *t, const void *label, size_t n);                  /* HashTranscript<Sha512> */
void tr_new_mode(transcript_t *t, const void *label, size_t n, int shake);  /* shake != 0: Shake128Transcript */
void shake128(uint8_t *out, size_t out_len, const uint8_t *const *parts, const size_t *lens, int n_parts);   /* one-shot XOF */
void tr_absorb(transcript_t *t, const void *d, size_t n);
void tr_squeeze(transcript_t *t, void *out, size_t n);

/* ---- suites (orc_suite.c) ---- */
const suite_t *orc_suite(int id);

/* ---- curve (orc_te.c) ---- */
void te_identity(te_ext *o, const suite_t *s);
void te_from_aff(te_ext *o, const te_aff *a, const suite_t *s);
void te_add(te_ext *o, const te_ext *p, const te_ext *q, const suite_t *s);
void te_madd(te_ext *o, const te_ext *p, const te_aff *q, const suite_t *s);
void te_dbl(te_ext *o, const te_ext *p, const suite_t *s);
void te_neg_aff(te_aff *o, const te_aff *p, const suite_t *s);
void te_to_aff(te_aff *o, const te_ext *p, const suite_t *s);
void te_batch_to_aff(te_aff *o, const te_ext *p, size_t n, const suite_t *s);
int  te_is_identity_ext(const te_ext *p, const suite_t *s);
int  te_is_identity_aff(const te_aff *p, const suite_t *s);
int  te_eq_ext(const te_ext *p, const te_ext *q, const suite_t *s);
int  te_on_curve(const te_aff *p, const suite_t *s);
int  te_in_subgroup(const te_aff *p, const suite_t *s);
void te_smul(te_ext *o, const te_aff *p, const u256 *k_plain, const suite_t *s);
/* ark-serialize compressed TE codec */
void te_encode(uint8_t *out /* pt_len */, const te_aff *p, const suite_t *s);
int  te_decode(te_aff *o, const uint8_t *in /* pt_len */, const suite_t *s);  /* 0 ok, else ORC_INVALID_DATA; no subgroup check */
/* canonical uncompressed x||y (LE32 each), non-Montgomery */
void te_encode_xy(uint8_t out[64], const te_aff *p, const suite_t *s);
int  te_decode_xy(te_aff *o, const uint8_t in[64], const suite_t *s);
/* SW presentation (suites with sw_codec): 33-byte ark-serialize compressed form <-> twisted-Edwards point */
void sw_encode(uint8_t out[33], const te_aff *p, const suite_t *s);
int  sw_decode(te_aff *o, const uint8_t in[33], const suite_t *s);   /* 0 ok; identity and undecodable -> ORC_INVALID_DATA */
int  sw_from_x(te_aff *o, const u256 *x_plain, int greatest, const suite_t *s);   /* SWAffine::get_point_from_x_unchecked + sw_to_te; 0 ok */

/* ---- MSM (orc_msm.c) ---- */
void orc_msm_naive(te_ext *o, const te_aff *bases, const u256 *scalars_plain, size_t n, const suite_t *s);
void orc_msm_pippenger(te_ext *o, const te_aff *bases, const u256 *scalars_plain, size_t n, const suite_t *s);
void orc_straus(te_ext *o, const te_aff *pts, const u256 *scalars_plain, size_t n, int w, const suite_t *s);

#ifdef __cplusplus
}
#endif
#endif
