/*
 * oracle/orc_field.c -- 256-bit integers and Montgomery prime fields (TEST ORACLE).
 *
 * Restates arkworks `ark_ff::Fp<MontBackend<_,4>>` (third-party, ark-ff 0.6, not
 * vendored under /root/reference): 4x64-bit little-endian limbs, R = 2^256,
 * CIOS Montgomery multiplication.  Reached from the reference at every field /
 * scalar operation, e.g. src/thin.rs:125 (`k + c * sk`), src/thin.rs:289-311.
 */
#include "orc.h"
#include <string.h>

typedef unsigned __int128 u128;

int u256_cmp(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] < b->l[i]) return -1;
        if (a->l[i] > b->l[i]) return 1;
    }
    return 0;
}
int u256_is_zero(const u256 *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

void u256_from_le(u256 *o, const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 7; j >= 0; j--) v = (v << 8) | b[8 * i + j];
        o->l[i] = v;
    }
}
void u256_to_le(uint8_t b[32], const u256 *a) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(a->l[i] >> (8 * j));
}

static uint64_t add_raw(u256 *o, const u256 *a, const u256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; o->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t sub_raw(u256 *o, const u256 *a, const u256 *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a->l[i] - b->l[i] - br;
        o->l[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1;
    }
    return br;
}

/* decimal string -> u256 (used for the suite constants, which the reference
 * writes as MontFp!("<decimal>"), e.g. src/suites/bandersnatch.rs:73-78) */
int u256_from_dec(u256 *o, const char *s) {
    memset(o, 0, sizeof *o);
    for (; *s; s++) {
        if (*s < '0' || *s > '9') return -1;
        u128 c = (u128)(*s - '0');
        for (int i = 0; i < 4; i++) { c += (u128)o->l[i] * 10; o->l[i] = (uint64_t)c; c >>= 64; }
        if (c) return -1;
    }
    return 0;
}

void mont_add(u256 *o, const u256 *a, const u256 *b, const mont_t *m) {
    u256 t, u; uint64_t c = add_raw(&t, a, b);
    uint64_t br = sub_raw(&u, &t, &m->p);
    *o = (c || !br) ? u : t;
}
void mont_sub(u256 *o, const u256 *a, const u256 *b, const mont_t *m) {
    u256 t; if (sub_raw(&t, a, b)) add_raw(&t, &t, &m->p);
    *o = t;
}
void mont_neg(u256 *o, const u256 *a, const mont_t *m) {
    if (u256_is_zero(a)) { *o = *a; return; }
    sub_raw(o, &m->p, a);
}

/* CIOS Montgomery multiplication, R = 2^256 */
void mont_mul(u256 *o, const u256 *a, const u256 *b, const mont_t *m) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * m->ninv;
        c = (u128)q * m->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)q * m->p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    u256 r = {{t[0], t[1], t[2], t[3]}}, u;
    uint64_t br = sub_raw(&u, &r, &m->p);
    *o = (t[4] || !br) ? u : r;
}
void mont_sqr(u256 *o, const u256 *a, const mont_t *m) { mont_mul(o, a, a, m); }

void mont_to(u256 *o, const u256 *a, const mont_t *m) { mont_mul(o, a, &m->r2, m); }
void mont_from(u256 *o, const u256 *a, const mont_t *m) {
    u256 one = {{1, 0, 0, 0}}; mont_mul(o, a, &one, m);
}

void mont_pow(u256 *o, const u256 *a, const u256 *e, const mont_t *m) {
    u256 r = m->r1, base = *a;
    for (int i = 255; i >= 0; i--) {
        mont_sqr(&r, &r, m);
        if ((e->l[i / 64] >> (i % 64)) & 1) mont_mul(&r, &r, &base, m);
    }
    *o = r;
}
void mont_inv(u256 *o, const u256 *a, const mont_t *m) {
    u256 e = m->p, two = {{2, 0, 0, 0}}; sub_raw(&e, &e, &two);
    mont_pow(o, a, &e, m);
}
int mont_is_square(const u256 *a, const mont_t *m) {
    if (u256_is_zero(a)) return 1;
    u256 r; mont_pow(&r, a, &m->pm1_half, m);
    return u256_cmp(&r, &m->r1) == 0;
}

static void shr1(u256 *a) {
    for (int i = 0; i < 4; i++) a->l[i] = (a->l[i] >> 1) | (i < 3 ? a->l[i + 1] << 63 : 0);
}

void mont_init(mont_t *m, const u256 *p) {
    memset(m, 0, sizeof *m);
    m->p = *p;
    uint64_t inv = 1; /* Newton: inv = p^-1 mod 2^64 */
    for (int i = 0; i < 7; i++) inv *= 2 - p->l[0] * inv;
    m->ninv = (uint64_t)0 - inv;
    int bits = 256; while (bits > 0 && !((p->l[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) bits--;
    m->bits = bits;
    /* R mod p by 256 modular doublings of 1; R^2 by 256 more */
    u256 x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        mont_add(&x, &x, &x, m);
        if (i == 255) m->r1 = x;
    }
    m->r2 = x;
    /* p-1 = 2^s * t */
    u256 t = *p; t.l[0] -= 1; m->pm1_half = t; shr1(&m->pm1_half);
    int s = 0; while (!(t.l[0] & 1)) { shr1(&t); s++; }
    m->two_adicity = s; m->t_odd = t;
    m->t_minus1_half = t; shr1(&m->t_minus1_half); /* (t-1)/2, t odd */
    /* smallest non-residue g, root_of_unity = g^t */
    for (uint64_t g = 2;; g++) {
        u256 gp = {{g, 0, 0, 0}}, gm; mont_to(&gm, &gp, m);
        if (!mont_is_square(&gm, m)) { mont_pow(&m->root_of_unity, &gm, &t, m); break; }
    }
}

/* Tonelli-Shanks square root (same result set as ark_ff SqrtPrecomputation::TonelliShanks;
 * which root is returned does not matter: callers pick by sign/parity) */
int mont_sqrt(u256 *o, const u256 *a, const mont_t *m) {
    if (u256_is_zero(a)) { *o = *a; return 1; }
    u256 z = m->root_of_unity, w, x, b;
    mont_pow(&w, a, &m->t_minus1_half, m);
    mont_mul(&x, &w, a, m);   /* a^((t+1)/2) */
    mont_mul(&b, &x, &w, m);  /* a^t */
    int v = m->two_adicity;
    while (u256_cmp(&b, &m->r1) != 0) {
        int k = 0; u256 b2k = b;
        while (u256_cmp(&b2k, &m->r1) != 0) { mont_sqr(&b2k, &b2k, m); k++; if (k == v) return 0; }
        u256 ww = z;
        for (int j = 0; j < v - k - 1; j++) mont_sqr(&ww, &ww, m);
        mont_sqr(&z, &ww, m);
        mont_mul(&b, &b, &z, m);
        mont_mul(&x, &x, &ww, m);
        v = k;
    }
    u256 chk; mont_sqr(&chk, &x, m);
    if (u256_cmp(&chk, a) != 0) return 0;
    *o = x; return 1;
}

/* PrimeField::from_be_bytes_mod_order: Horner over bytes, most significant first.
 * (ark-ff processes in chunks; the value -- int(bytes) mod p -- is what matters.) */
void mont_from_be_bytes_mod_order(u256 *o, const uint8_t *b, size_t n, const mont_t *m) {
    u256 acc = {{0, 0, 0, 0}}, c256 = {{256, 0, 0, 0}}, m256;
    mont_to(&m256, &c256, m);
    for (size_t i = 0; i < n; i++) {
        u256 d = {{b[i], 0, 0, 0}}, dm; mont_to(&dm, &d, m);
        mont_mul(&acc, &acc, &m256, m);
        mont_add(&acc, &acc, &dm, m);
    }
    *o = acc;
}
/* src/utils/common.rs:65-76 use from_le_bytes_mod_order on squeezed bytes */
void mont_from_le_bytes_mod_order(u256 *o, const uint8_t *b, size_t n, const mont_t *m) {
    uint8_t tmp[128];
    if (n > sizeof tmp) n = sizeof tmp;
    for (size_t i = 0; i < n; i++) tmp[i] = b[n - 1 - i];
    mont_from_be_bytes_mod_order(o, tmp, n, m);
}
