/*
 * oracle/orc_msm.c -- multi-scalar multiplication on the CPU (TEST ORACLE / cpu_baseline).
 *
 * orc_msm_naive      : sum of independent double-and-add products; the simplest
 *                      statement of `VariableBaseMSM::msm_unchecked` (call sites
 *                      src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411).
 * orc_msm_pippenger  : bucket method with arkworks' window rule (SURVEY.md A.9:
 *                      c = 3 for n < 32, else floor(log2(n)*69/100)+2; signed digits;
 *                      running-sum bucket reduction; windows combined MSB->LSB).
 *                      Used as the single-thread CPU baseline ("port") in bench.py.
 * orc_straus         : src/utils/straus.rs:24-100 (`short_msm`, table of (2^w)^n entries).
 * All three return the same group element; parity is on the normalised result.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

#define FQ (&s->fq)

void orc_msm_naive(te_ext *o, const te_aff *bases, const u256 *sc, size_t n, const suite_t *s) {
    te_ext acc; te_identity(&acc, s);
    for (size_t i = 0; i < n; i++) { te_ext t; te_smul(&t, &bases[i], &sc[i], s); te_add(&acc, &acc, &t, s); }
    *o = acc;
}

typedef struct { u256 x, y, k; } te_pre; /* k = d*x*y */

static void mul_a(u256 *o, const u256 *v, const suite_t *s) {
    if (s->a_is_minus5) { u256 t; mont_add(&t, v, v, FQ); mont_add(&t, &t, &t, FQ); mont_add(&t, &t, v, FQ); mont_neg(o, &t, FQ); }
    else mont_mul(o, &s->a, v, FQ); /* a = 1 (Baby-JubJub), a = -1 (JubJub) */
}
/* madd-2008-hwcd with precomputed k = d*x2*y2: 8M; neg != 0 adds -Q */
static void madd_pre(te_ext *p, const te_pre *q, int neg, const suite_t *s) {
    u256 A, B, C, E, F, G, H, t0, t1, qx, qk;
    if (neg) { mont_neg(&qx, &q->x, FQ); mont_neg(&qk, &q->k, FQ); } else { qx = q->x; qk = q->k; }
    mont_mul(&A, &p->x, &qx, FQ);
    mont_mul(&B, &p->y, &q->y, FQ);
    mont_mul(&C, &p->t, &qk, FQ);
    mont_add(&t0, &p->x, &p->y, FQ); mont_add(&t1, &qx, &q->y, FQ);
    mont_mul(&E, &t0, &t1, FQ); mont_sub(&E, &E, &A, FQ); mont_sub(&E, &E, &B, FQ);
    mont_sub(&F, &p->z, &C, FQ); mont_add(&G, &p->z, &C, FQ);
    mul_a(&t0, &A, s); mont_sub(&H, &B, &t0, FQ);
    mont_mul(&p->x, &E, &F, FQ); mont_mul(&p->y, &G, &H, FQ);
    mont_mul(&p->t, &E, &H, FQ); mont_mul(&p->z, &F, &G, FQ);
}

static int ark_window(size_t n) {
    if (n < 32) return 3;
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
    /* ln_without_floats: log2(n) * 69 / 100 */
    return lg * 69 / 100 + 2;
}

void orc_msm_pippenger(te_ext *o, const te_aff *bases, const u256 *sc, size_t n, const suite_t *s) {
    te_identity(o, s);
    if (!n) return;
    int c = ark_window(n);
    int nbits = s->fr.bits;
    int nwin = (nbits + c) / c + ((nbits + c) % c ? 0 : 0); /* ceil((nbits+1)/c): room for the signed carry */
    nwin = (nbits + 1 + c - 1) / c;
    size_t nb = (size_t)1 << (c - 1);
    te_pre *pre = (te_pre *)malloc(n * sizeof(te_pre));
    for (size_t i = 0; i < n; i++) {
        pre[i].x = bases[i].x; pre[i].y = bases[i].y;
        if (s->sw_native) continue;                                /* short-Weierstrass suite: te_madd below */
        mont_mul(&pre[i].k, &bases[i].x, &bases[i].y, FQ); mont_mul(&pre[i].k, &pre[i].k, &s->d, FQ);
    }
    /* signed digits: d in (-2^(c-1), 2^(c-1)] */
    int32_t *dig = (int32_t *)malloc(n * (size_t)nwin * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        int carry = 0;
        for (int w = 0; w < nwin; w++) {
            int bit = w * c; uint64_t v = 0;
            if (bit < 256) {
                v = sc[i].l[bit / 64] >> (bit % 64);
                if (bit % 64 + c > 64 && bit / 64 + 1 < 4) v |= sc[i].l[bit / 64 + 1] << (64 - bit % 64);
                v &= ((uint64_t)1 << c) - 1;
            }
            int64_t d = (int64_t)v + carry;
            if (d > (int64_t)nb) { d -= (int64_t)1 << c; carry = 1; } else carry = 0;
            dig[i * nwin + w] = (int32_t)d;
        }
    }
    te_ext *bk = (te_ext *)malloc(nb * sizeof(te_ext));
    te_ext total; te_identity(&total, s);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) te_dbl(&total, &total, s);
        for (size_t b = 0; b < nb; b++) te_identity(&bk[b], s);
        for (size_t i = 0; i < n; i++) {
            int32_t d = dig[i * nwin + w];
            if (s->sw_native) {
                te_aff q = bases[i]; if (d < 0) te_neg_aff(&q, &q, s);
                if (d) { te_ext *b = &bk[(d > 0 ? d : -d) - 1]; te_madd(b, b, &q, s); }
            }
            else if (d > 0) madd_pre(&bk[d - 1], &pre[i], 0, s);
            else if (d < 0) madd_pre(&bk[-d - 1], &pre[i], 1, s);
        }
        te_ext run, sum; te_identity(&run, s); te_identity(&sum, s);
        for (size_t b = nb; b-- > 0;) { te_add(&run, &run, &bk[b], s); te_add(&sum, &sum, &run, s); }
        te_add(&total, &total, &sum, s);
    }
    *o = total;
    free(bk); free(dig); free(pre);
}

/* src/utils/straus.rs:24-100 */
void orc_straus(te_ext *o, const te_aff *pts, const u256 *sc, size_t n, int w, const suite_t *s) {
    size_t c = (size_t)1 << w, total = 1;
    for (size_t i = 0; i < n; i++) total *= c;
    te_ext *tab = (te_ext *)malloc(total * sizeof(te_ext));
    te_aff *taba = (te_aff *)malloc(total * sizeof(te_aff));
    size_t len = 1; te_identity(&tab[0], s);
    for (size_t i = 0; i < n; i++) {           /* straus.rs:28-42 */
        size_t prev = len;
        for (size_t j = 0; j < prev; j++) te_madd(&tab[len++], &tab[j], &pts[i], s);
        for (size_t k = 2; k < c; k++)
            for (size_t j = 0; j < prev; j++) te_madd(&tab[len++], &tab[(k - 1) * prev + j], &pts[i], s);
    }
    te_batch_to_aff(taba, tab, total, s);      /* straus.rs:43 */
    int ndig = (256 + w - 1) / w;              /* straus.rs:62-63: NUM_LIMBS*64 bits */
    te_ext acc; te_identity(&acc, s);
    int started = 0;
    for (int i = 0; i < ndig; i++) {           /* straus.rs:93-98 */
        int bit = (ndig - 1 - i) * w; size_t idx = 0, pc = 1;
        for (size_t j = 0; j < n; j++) {
            uint64_t v = sc[j].l[bit / 64] >> (bit % 64);
            if (bit % 64 + w > 64 && bit / 64 + 1 < 4) v |= sc[j].l[bit / 64 + 1] << (64 - bit % 64);
            idx += (size_t)(v & (c - 1)) * pc; pc *= c;
        }
        if (!started && idx == 0) continue;    /* skip_while(idx == 0) */
        started = 1;
        for (int k = 0; k < w; k++) te_dbl(&acc, &acc, s);
        te_madd(&acc, &acc, &taba[idx], s);
    }
    *o = acc;
    free(tab); free(taba);
}
