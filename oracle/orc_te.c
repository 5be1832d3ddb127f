/*
 * oracle/orc_te.c -- suites and twisted-Edwards group arithmetic (TEST ORACLE).
 *
 * Restates, from the published formulas, what the reference gets from arkworks
 * `ark_ec::twisted_edwards::{Affine, Projective}` (ark-ec 0.6, third-party):
 * extended coordinates (X:Y:T:Z), add-2008-hwcd / dbl-2008-hwcd for general `a`.
 * Call sites in the reference: src/lib.rs:332,392 (sk*G, sk*I), src/thin.rs:119,158,
 * src/pedersen.rs:148-167,229-245, src/utils/common.rs:400-404,414.
 * Suite constants: src/suites/bandersnatch.rs:13-14,62-105, src/suites/baby_jubjub.rs:12-13,56-95, src/suites/jubjub.rs:12-13,56-95.
 * Any correct group law yields the same group element; parity is defined on the
 * normalised affine result / its encoding.
 */
#include <pthread.h>
#include "orc.h"
#include <stdlib.h>
#include <string.h>

static suite_t g_suites[8];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;      /* gen_batch() calls in from several threads at once */

static void fq_dec(u256 *o, const char *dec, const mont_t *m) {
    u256 t; u256_from_dec(&t, dec); mont_to(o, &t, m);
}
static void fq_small(u256 *o, uint64_t v, const mont_t *m) {
    u256 t = {{v, 0, 0, 0}}; mont_to(o, &t, m);
}

static void init_suites(void) {
    u256 p;
    /* ---- Bandersnatch-SHA512-ELL2-v1 (src/suites/bandersnatch.rs:62-105) ---- */
    suite_t *s = &g_suites[0];
    memset(s, 0, sizeof *s);
    s->id = ORC_SUITE_BANDERSNATCH;
    s->suite_id = "Bandersnatch-SHA512-ELL2-v1"; s->suite_id_len = 27;
    u256_from_dec(&p, "52435875175126190479447740508185965837690552500527637822603658699938581184513");
    mont_init(&s->fq, &p);
    u256_from_dec(&p, "13108968793781547619861935127046491459309155893440570251786403306729687672801");
    mont_init(&s->fr, &p);
    { u256 five; fq_small(&five, 5, &s->fq); mont_neg(&s->a, &five, &s->fq); }
    s->a_is_minus5 = 1;
    fq_dec(&s->d, "45022363124591815672509500913686876175488063829319466900776701791074614335719", &s->fq);
    s->cofactor = 4; s->h2c = ORC_H2C_ELL2;
    fq_dec(&s->G.x, "18886178867200960497001835917649091219057080094937609519140440539760939937304", &s->fq);
    fq_dec(&s->G.y, "19188667384257783945677642223292697773471335439753913231509108946878080696678", &s->fq);
    fq_dec(&s->B.x, "23335687741101763108036518445642207119627658113885888016488710494487028845889", &s->fq);
    fq_dec(&s->B.y, "5552214580375038693022409684979828600325210968745774080859660443337357929963", &s->fq);
    fq_dec(&s->ACC.x, "14056632001415368875257708737821299882600475929746323097150942355715730684350", &s->fq);
    fq_dec(&s->ACC.y, "10322661992765989500407719465917595459409463902187386706652408883505670839210", &s->fq);
    fq_dec(&s->PAD.x, "26913883415342152801331916189968962157924271221160514298872262294143390094043", &s->fq);
    fq_dec(&s->PAD.y, "30874728313203001508631936119690348239461579770372782660098261717479009115354", &s->fq);
    /* Elligator2 Montgomery-model constants: src/suites/bandersnatch_sw.rs:104-111 (J = A, K = B), Z = 5 */
    fq_dec(&s->ell2_j, "29978822694968839326280996386011761570173833766074948509196803838190355340952", &s->fq);
    fq_dec(&s->ell2_k, "25465760566081946422412445027709227188579564747101592991722834452325077642517", &s->fq);
    fq_small(&s->ell2_z, 5, &s->fq);

    /* ---- BabyJubJub-SHA512-TAI-v1 (src/suites/baby_jubjub.rs:56-95) ---- */
    s = &g_suites[1];
    memset(s, 0, sizeof *s);
    s->id = ORC_SUITE_BABYJUBJUB;
    s->suite_id = "BabyJubJub-SHA512-TAI-v1"; s->suite_id_len = 24;
    u256_from_dec(&p, "21888242871839275222246405745257275088548364400416034343698204186575808495617");
    mont_init(&s->fq, &p);
    u256_from_dec(&p, "2736030358979909402780800718157159386076813972158567259200215660948447373041");
    mont_init(&s->fr, &p);
    fq_small(&s->a, 1, &s->fq);
    fq_dec(&s->d, "9706598848417545097372247223557719406784115219466060233080913168975159366771", &s->fq);
    s->cofactor = 8; s->h2c = ORC_H2C_TAI;
    fq_dec(&s->G.x, "19698561148652590122159747500897617769866003486955115824547446575314762165298", &s->fq);
    fq_dec(&s->G.y, "19298250018296453272277890825869354524455968081175474282777126169995084727839", &s->fq);
    fq_dec(&s->B.x, "15549380791300914366206471199568039679131690710803662429646809536753521087193", &s->fq);
    fq_dec(&s->B.y, "15218614024055502695611547593111691164731001864276292210438920202280814188379", &s->fq);
    fq_dec(&s->ACC.x, "6402374321243162085389111671722843560682527921646684137786768606010797479351", &s->fq);
    fq_dec(&s->ACC.y, "9735581299071570006712034490635195155689931359428941496570758703259384062170", &s->fq);
    fq_dec(&s->PAD.x, "11167490195257431015694161063225325511805242064780376648595733691987293447528", &s->fq);
    fq_dec(&s->PAD.y, "18403369502642103292159933062507105566469227524991433735553439433605496057425", &s->fq);

    /* ---- JubJub-SHA512-TAI-v1 (src/suites/jubjub.rs:56-95; curve: ark-ed-on-bls12-381, a = -1, d = -(10240/10241)) ---- */
    s = &g_suites[2];
    memset(s, 0, sizeof *s);
    s->id = ORC_SUITE_JUBJUB;
    s->suite_id = "JubJub-SHA512-TAI-v1"; s->suite_id_len = 20;
    u256_from_dec(&p, "52435875175126190479447740508185965837690552500527637822603658699938581184513");
    mont_init(&s->fq, &p);
    u256_from_dec(&p, "6554484396890773809930967563523245729705921265872317281365359162392183254199");
    mont_init(&s->fr, &p);
    { u256 one; fq_small(&one, 1, &s->fq); mont_neg(&s->a, &one, &s->fq); }
    fq_dec(&s->d, "19257038036680949359750312669786877991949435402254120286184196891950884077233", &s->fq);
    s->cofactor = 8; s->h2c = ORC_H2C_TAI;
    fq_dec(&s->G.x, "8076246640662884909881801758704306714034609987455869804520522091855516602923", &s->fq);
    fq_dec(&s->G.y, "13262374693698910701929044844600465831413122818447359594527400194675274060458", &s->fq);
    fq_dec(&s->B.x, "38206460563694846719174258613922853630278999941532690543235578292520143148532", &s->fq);
    fq_dec(&s->B.y, "34254498978062207918041301829525626783549813531091321004550549786528984401675", &s->fq);
    fq_dec(&s->ACC.x, "48142684311216766702182564801462043940571084233680216669499475549492432046964", &s->fq);
    fq_dec(&s->ACC.y, "34380560660182334518990118617091967209302636551264477863958902286043397647879", &s->fq);
    fq_dec(&s->PAD.x, "17348704025397475127937572481155408456556065464328870407269802701696798733683", &s->fq);
    fq_dec(&s->PAD.y, "24318278422173803457621119807961883607097742387673491974779969503617097905596", &s->fq);

    /* ---- Ed25519-SHA512-TAI-v1 (src/suites/ed25519.rs:44-66; curve: ark-ed25519 = edwards25519, a = -1, d = -121665/121666).
     * Tiny / Thin / Pedersen only: the reference implements no RingSuite for it; ACC / PAD are unused placeholders. ---- */
    s = &g_suites[3];
    memset(s, 0, sizeof *s);
    s->id = ORC_SUITE_ED25519;
    s->suite_id = "Ed25519-SHA512-TAI-v1"; s->suite_id_len = 21;
    u256_from_dec(&p, "57896044618658097711785492504343953926634992332820282019728792003956564819949");
    mont_init(&s->fq, &p);
    u256_from_dec(&p, "7237005577332262213973186563042994240857116359379907606001950938285454250989");
    mont_init(&s->fr, &p);
    { u256 one; fq_small(&one, 1, &s->fq); mont_neg(&s->a, &one, &s->fq); }
    fq_dec(&s->d, "37095705934669439343138083508754565189542113879843219016388785533085940283555", &s->fq);
    s->cofactor = 8; s->h2c = ORC_H2C_TAI;
    fq_dec(&s->G.x, "15112221349535400772501151409588531511454012693041857206046113283949847762202", &s->fq);
    fq_dec(&s->G.y, "46316835694926478169428394003475163141307993866256225615783033603165251855960", &s->fq);
    fq_dec(&s->B.x, "45003173884697328536089278691112838614164406922820087464913813433380838325453", &s->fq);
    fq_dec(&s->B.y, "31256014272390301975555524011230972931324093235775711248505761870355310252869", &s->fq);
    s->ACC = s->G; s->PAD = s->G;

    /* ---- Bandersnatch-SW-SHA512-TAI-v1 (src/suites/bandersnatch_sw.rs:60-112): the SAME curve as suite 0 in its
     * short-Weierstrass presentation (Affine = SWAffine).  The maps of src/utils/te_sw_map.rs are group isomorphisms and take
     * the SW generator to the TE generator, so the arithmetic is suite 0's; what differs is every serialised point (33-byte SW
     * form), try-and-increment on SW x-coordinates, and the suite points, given in the reference as SW coordinates. ---- */
    s = &g_suites[4];
    *s = g_suites[0];
    s->id = ORC_SUITE_BANDERSNATCH_SW;
    s->suite_id = "Bandersnatch-SW-SHA512-TAI-v1"; s->suite_id_len = 29;
    s->h2c = ORC_H2C_TAI_SW; s->sw_codec = 1;
    s->mont_b = s->ell2_k;                                                 /* MontCurveConfig::COEFF_B */
    fq_dec(&s->mont_a3, "9992940898322946442093665462003920523391277922024982836398934612730118446984", &s->fq);   /* :104-111 */
    fq_dec(&s->mont_binv, "41180284393978236561320365279764246793818536543197771097409483252169927600582", &s->fq);
    {   /* y^2 = x^3 + a x + b with a = (3 - A^2) / (3 B^2), b = (2 A^3 - 9 A) / (27 B^3) */
        u256 A, A2, A3c, B2, B3, t, u, three, nine, c27, two;
        A = s->ell2_j; fq_small(&three, 3, &s->fq); fq_small(&nine, 9, &s->fq); fq_small(&c27, 27, &s->fq); fq_small(&two, 2, &s->fq);
        mont_sqr(&A2, &A, &s->fq); mont_mul(&A3c, &A2, &A, &s->fq);
        mont_sqr(&B2, &s->mont_b, &s->fq); mont_mul(&B3, &B2, &s->mont_b, &s->fq);
        mont_sub(&t, &three, &A2, &s->fq); mont_mul(&u, &three, &B2, &s->fq); mont_inv(&u, &u, &s->fq); mont_mul(&s->sw_a, &t, &u, &s->fq);
        mont_mul(&t, &two, &A3c, &s->fq); mont_mul(&u, &nine, &A, &s->fq); mont_sub(&t, &t, &u, &s->fq);
        mont_mul(&u, &c27, &B3, &s->fq); mont_inv(&u, &u, &s->fq); mont_mul(&s->sw_b, &t, &u, &s->fq);
    }
    {   /* suite points: SW coordinates of the reference -> TE */
        static const char *sw_pts[3][2] = {
            {"28115362618644671219696075022370511395136332234538034358311199318506963235315",      /* BLINDING_BASE :71-79 */
             "3900851469868158154936962463930962496000252801946757953905982128670530185313"},
            {"13189182432637108534251278524663360416811744717379968387043749958796254980045",      /* ACCUMULATOR_BASE :87-95 */
             "14483286006782706188671626508232161325054303360192563232232823772738911894793"},
            {"20496180070424734470560955314776462366297546779079302509428101119888111900885",      /* PADDING :97-104 */
             "8839106592405352067483360946162273985142890146060814748321063063028225641813"}};
        te_aff *dst[3] = {&s->B, &s->ACC, &s->PAD};
        for (int i = 0; i < 3; i++) {
            u256 x, y, mx, my, one = s->fq.r1, t, w;
            fq_dec(&x, sw_pts[i][0], &s->fq); fq_dec(&y, sw_pts[i][1], &s->fq);
            mont_mul(&mx, &s->mont_b, &x, &s->fq); mont_sub(&mx, &mx, &s->mont_a3, &s->fq); mont_mul(&my, &s->mont_b, &y, &s->fq);
            mont_inv(&t, &my, &s->fq); mont_mul(&dst[i]->x, &mx, &t, &s->fq);
            mont_add(&t, &mx, &one, &s->fq); mont_inv(&t, &t, &s->fq); mont_sub(&w, &mx, &one, &s->fq); mont_mul(&dst[i]->y, &w, &t, &s->fq);
        }
    }

    /* ---- Bandersnatch-SHAKE128-ELL2-v1 (src/suites/bandersnatch_shake128.rs): suite 0's curve and Elligator2 map with the
     * SHAKE128 sponge as the transcript (XofTranscript<Shake128>) and expand_message_xof in hash-to-curve ---- */
    s = &g_suites[5];
    *s = g_suites[0];
    s->id = ORC_SUITE_BANDERSNATCH_SHAKE128;
    s->suite_id = "Bandersnatch-SHAKE128-ELL2-v1"; s->suite_id_len = 29;
    s->xof_shake = 1;
    fq_dec(&s->B.x, "6153734995852631824944342602386415873379775188383988340041079006556670120775", &s->fq);
    fq_dec(&s->B.y, "27204351599954061630605768787803524395123895650061061132592995395630473050754", &s->fq);
    fq_dec(&s->ACC.x, "27631238720955528589004064829276283990465032040945349648037876197995278250917", &s->fq);
    fq_dec(&s->ACC.y, "37605358688136619817560700742505556266961225274493904038881144193539047100140", &s->fq);
    fq_dec(&s->PAD.x, "1834402953989431481748983728202937234471322740714585873803966488035889514523", &s->fq);
    fq_dec(&s->PAD.y, "52100941849053769665273763352270294131006971127418863694682093199651869272752", &s->fq);

    /* ---- Testing-SHA256-TAI-v1 (src/suites/testing.rs): the crate's own test suite -- edwards25519 with HashTranscript<Sha256> ---- */
    s = &g_suites[6];
    *s = g_suites[3];
    s->id = ORC_SUITE_TESTING_SHA256;
    s->suite_id = "Testing-SHA256-TAI-v1"; s->suite_id_len = 21;
    s->xof_shake = 2;
    fq_dec(&s->B.x, "3310617998588019043596181043598335786888094217571323926547956053100032777190", &s->fq);
    fq_dec(&s->B.y, "16824531136491949759823061604778551593864344614632277377095388820423530178202", &s->fq);

    /* ---- Secp256r1-SHA256-TAI-v1 (src/suites/secp256r1.rs:49-70; curve: ark-secp256r1 = NIST P-256, SP 800-186 3.2.1.3):
     * short Weierstrass a = -3, prime order (cofactor 1), 256-bit base and scalar fields with the top bit set,
     * HashTranscript<Sha256>, try-and-increment on SW x-coordinates, 33-byte points ---- */
    s = &g_suites[7];
    memset(s, 0, sizeof *s);
    s->id = ORC_SUITE_SECP256R1;
    s->suite_id = "Secp256r1-SHA256-TAI-v1"; s->suite_id_len = 23;
    u256_from_dec(&p, "115792089210356248762697446949407573530086143415290314195533631308867097853951");
    mont_init(&s->fq, &p);
    u256_from_dec(&p, "115792089210356248762697446949407573529996955224135760342422259061068512044369");
    mont_init(&s->fr, &p);
    s->cofactor = 1; s->h2c = ORC_H2C_TAI_SW; s->xof_shake = 2; s->sw_native = 1; s->pt_len = 33;
    { u256 three; fq_small(&three, 3, &s->fq); mont_neg(&s->sw_a, &three, &s->fq); }
    fq_dec(&s->sw_b, "41058363725152142129326129780047268409114441015993725554835256314039467401291", &s->fq);
    fq_dec(&s->G.x, "48439561293906451759052585252797914202762949526041747995844080717082404635286", &s->fq);
    fq_dec(&s->G.y, "36134250956749795798585127919587881956611106672985015071877198253568414405109", &s->fq);
    fq_dec(&s->B.x, "100063053743935619201936855760019111820847755970243670581468062459849338000", &s->fq);     /* secp256r1.rs:57-65 */
    fq_dec(&s->B.y, "113675507039234898358330549589155441528265243038226986303017485279501143145422", &s->fq);
    for (int i = 0; i < 7; i++) if (!g_suites[i].pt_len) g_suites[i].pt_len = 32;
}

const suite_t *orc_suite(int id) {
    pthread_once(&g_once, init_suites);
    if (id < 0 || id > 7) return NULL;
    return &g_suites[id];
}

/* ---- short-Weierstrass group law for sw_native suites (a = -3), Jacobian coordinates.  Restates what the reference gets from
 * ark_ec::short_weierstrass::{Affine, Projective} for ark_secp256r1 (third-party); formulas dbl-2001-b / add-2007-bl (EFD).
 * Any correct group law yields the same group element; parity is on normalised encodings. ---- */
#define FQ (&s->fq)
static void swn_dbl(te_ext *o, const te_ext *p, const suite_t *s) {
    u256 delta, gamma, beta, alpha, t0, t1, x3, y3, z3;
    mont_sqr(&delta, &p->z, FQ); mont_sqr(&gamma, &p->y, FQ); mont_mul(&beta, &p->x, &gamma, FQ);
    mont_sub(&t0, &p->x, &delta, FQ); mont_add(&t1, &p->x, &delta, FQ); mont_mul(&alpha, &t0, &t1, FQ);
    mont_add(&t0, &alpha, &alpha, FQ); mont_add(&alpha, &t0, &alpha, FQ);                     /* 3 (X - delta)(X + delta) */
    mont_add(&t0, &beta, &beta, FQ); mont_add(&t0, &t0, &t0, FQ);                             /* 4 beta */
    mont_sqr(&x3, &alpha, FQ); mont_sub(&x3, &x3, &t0, FQ); mont_sub(&x3, &x3, &t0, FQ);
    mont_add(&t1, &p->y, &p->z, FQ); mont_sqr(&z3, &t1, FQ); mont_sub(&z3, &z3, &gamma, FQ); mont_sub(&z3, &z3, &delta, FQ);
    mont_sub(&t0, &t0, &x3, FQ); mont_mul(&y3, &alpha, &t0, FQ);
    mont_sqr(&t1, &gamma, FQ); mont_add(&t1, &t1, &t1, FQ); mont_add(&t1, &t1, &t1, FQ); mont_add(&t1, &t1, &t1, FQ);   /* 8 gamma^2 */
    mont_sub(&y3, &y3, &t1, FQ);
    o->x = x3; o->y = y3; o->z = z3; memset(&o->t, 0, sizeof o->t);
}
static void swn_add(te_ext *o, const te_ext *p, const te_ext *q, const suite_t *s) {
    if (u256_is_zero(&p->z)) { *o = *q; return; }
    if (u256_is_zero(&q->z)) { *o = *p; return; }
    u256 z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t0, x3, y3, z3;
    mont_sqr(&z1z1, &p->z, FQ); mont_sqr(&z2z2, &q->z, FQ);
    mont_mul(&u1, &p->x, &z2z2, FQ); mont_mul(&u2, &q->x, &z1z1, FQ);
    mont_mul(&s1, &p->y, &q->z, FQ); mont_mul(&s1, &s1, &z2z2, FQ);
    mont_mul(&s2, &q->y, &p->z, FQ); mont_mul(&s2, &s2, &z1z1, FQ);
    mont_sub(&h, &u2, &u1, FQ); mont_sub(&r, &s2, &s1, FQ);
    if (u256_is_zero(&h)) {
        if (u256_is_zero(&r)) { swn_dbl(o, p, s); return; }
        memset(o, 0, sizeof *o); o->x = s->fq.r1; o->y = s->fq.r1; return;                   /* P + (-P) */
    }
    mont_add(&r, &r, &r, FQ);
    mont_add(&i, &h, &h, FQ); mont_sqr(&i, &i, FQ); mont_mul(&j, &h, &i, FQ); mont_mul(&v, &u1, &i, FQ);
    mont_sqr(&x3, &r, FQ); mont_sub(&x3, &x3, &j, FQ); mont_sub(&x3, &x3, &v, FQ); mont_sub(&x3, &x3, &v, FQ);
    mont_sub(&t0, &v, &x3, FQ); mont_mul(&y3, &r, &t0, FQ); mont_mul(&t0, &s1, &j, FQ); mont_add(&t0, &t0, &t0, FQ); mont_sub(&y3, &y3, &t0, FQ);
    mont_add(&t0, &p->z, &q->z, FQ); mont_sqr(&t0, &t0, FQ); mont_sub(&t0, &t0, &z1z1, FQ); mont_sub(&t0, &t0, &z2z2, FQ); mont_mul(&z3, &t0, &h, FQ);
    o->x = x3; o->y = y3; o->z = z3; memset(&o->t, 0, sizeof o->t);
}
#undef FQ

/* ---- group law ---- */
#define FQ (&s->fq)

void te_identity(te_ext *o, const suite_t *s) {
    memset(o, 0, sizeof *o); o->y = s->fq.r1;
    if (s->sw_native) { o->x = s->fq.r1; return; }                 /* (1 : 1 : 0) */
    o->z = s->fq.r1;
}
void te_from_aff(te_ext *o, const te_aff *a, const suite_t *s) {
    if (s->sw_native) {
        if (te_is_identity_aff(a, s)) { te_identity(o, s); return; }
        memset(o, 0, sizeof *o); o->x = a->x; o->y = a->y; o->z = s->fq.r1; return;
    }
    o->x = a->x; o->y = a->y; o->z = s->fq.r1; mont_mul(&o->t, &a->x, &a->y, FQ);
}
/* add-2008-hwcd (general a): 9M + 1*a + 1*d */
void te_add(te_ext *o, const te_ext *p, const te_ext *q, const suite_t *s) {
    if (s->sw_native) { swn_add(o, p, q, s); return; }
    u256 A, B, C, D, E, F, G, H, t0, t1;
    mont_mul(&A, &p->x, &q->x, FQ);
    mont_mul(&B, &p->y, &q->y, FQ);
    mont_mul(&C, &p->t, &q->t, FQ); mont_mul(&C, &C, &s->d, FQ);
    mont_mul(&D, &p->z, &q->z, FQ);
    mont_add(&t0, &p->x, &p->y, FQ); mont_add(&t1, &q->x, &q->y, FQ);
    mont_mul(&E, &t0, &t1, FQ); mont_sub(&E, &E, &A, FQ); mont_sub(&E, &E, &B, FQ);
    mont_sub(&F, &D, &C, FQ); mont_add(&G, &D, &C, FQ);
    mont_mul(&t0, &s->a, &A, FQ); mont_sub(&H, &B, &t0, FQ);
    mont_mul(&o->x, &E, &F, FQ); mont_mul(&o->y, &G, &H, FQ);
    mont_mul(&o->t, &E, &H, FQ); mont_mul(&o->z, &F, &G, FQ);
}
void te_madd(te_ext *o, const te_ext *p, const te_aff *q, const suite_t *s) {
    te_ext e; te_from_aff(&e, q, s); te_add(o, p, &e, s);
}
/* dbl-2008-hwcd */
void te_dbl(te_ext *o, const te_ext *p, const suite_t *s) {
    if (s->sw_native) { swn_dbl(o, p, s); return; }
    u256 A, B, C, D, E, F, G, H, t0;
    mont_sqr(&A, &p->x, FQ); mont_sqr(&B, &p->y, FQ);
    mont_sqr(&C, &p->z, FQ); mont_add(&C, &C, &C, FQ);
    mont_mul(&D, &s->a, &A, FQ);
    mont_add(&t0, &p->x, &p->y, FQ); mont_sqr(&E, &t0, FQ); mont_sub(&E, &E, &A, FQ); mont_sub(&E, &E, &B, FQ);
    mont_add(&G, &D, &B, FQ); mont_sub(&F, &G, &C, FQ); mont_sub(&H, &D, &B, FQ);
    mont_mul(&o->x, &E, &F, FQ); mont_mul(&o->y, &G, &H, FQ);
    mont_mul(&o->t, &E, &H, FQ); mont_mul(&o->z, &F, &G, FQ);
}
void te_neg_aff(te_aff *o, const te_aff *p, const suite_t *s) {
    if (s->sw_native) { o->x = p->x; mont_neg(&o->y, &p->y, FQ); return; }
    mont_neg(&o->x, &p->x, FQ); o->y = p->y;
}

void te_to_aff(te_aff *o, const te_ext *p, const suite_t *s) {
    if (s->sw_native) {
        if (u256_is_zero(&p->z)) { memset(o, 0, sizeof *o); return; }
        u256 zi, zi2; mont_inv(&zi, &p->z, FQ); mont_sqr(&zi2, &zi, FQ);
        mont_mul(&o->x, &p->x, &zi2, FQ); mont_mul(&zi2, &zi2, &zi, FQ); mont_mul(&o->y, &p->y, &zi2, FQ); return;
    }
    u256 zi; mont_inv(&zi, &p->z, FQ);
    mont_mul(&o->x, &p->x, &zi, FQ); mont_mul(&o->y, &p->y, &zi, FQ);
}
/* CurveGroup::normalize_batch (Montgomery's trick), src/utils/common.rs:414 */
void te_batch_to_aff(te_aff *o, const te_ext *p, size_t n, const suite_t *s) {
    if (!n) return;
    if (s->sw_native) { for (size_t i = 0; i < n; i++) te_to_aff(&o[i], &p[i], s); return; }   /* (the oracle is not timed on this suite) */
    u256 *pre = (u256 *)malloc(n * sizeof(u256));
    u256 acc = s->fq.r1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; mont_mul(&acc, &acc, &p[i].z, FQ); }
    u256 inv; mont_inv(&inv, &acc, FQ);
    for (size_t i = n; i-- > 0;) {
        u256 zi; mont_mul(&zi, &inv, &pre[i], FQ);
        mont_mul(&inv, &inv, &p[i].z, FQ);
        mont_mul(&o[i].x, &p[i].x, &zi, FQ); mont_mul(&o[i].y, &p[i].y, &zi, FQ);
    }
    free(pre);
}
int te_is_identity_ext(const te_ext *p, const suite_t *s) {
    if (s->sw_native) return u256_is_zero(&p->z);
    return u256_is_zero(&p->x) && u256_cmp(&p->y, &p->z) == 0;
}
int te_is_identity_aff(const te_aff *p, const suite_t *s) {
    if (s->sw_native) return u256_is_zero(&p->x) && u256_is_zero(&p->y);
    return u256_is_zero(&p->x) && u256_cmp(&p->y, &s->fq.r1) == 0;
}
int te_eq_ext(const te_ext *p, const te_ext *q, const suite_t *s) {
    u256 a, b;
    if (s->sw_native) {
        int pi = u256_is_zero(&p->z), qi = u256_is_zero(&q->z);
        if (pi || qi) return pi && qi;
        u256 z1z1, z2z2; mont_sqr(&z1z1, &p->z, FQ); mont_sqr(&z2z2, &q->z, FQ);
        mont_mul(&a, &p->x, &z2z2, FQ); mont_mul(&b, &q->x, &z1z1, FQ);
        if (u256_cmp(&a, &b)) return 0;
        mont_mul(&a, &p->y, &z2z2, FQ); mont_mul(&a, &a, &q->z, FQ); mont_mul(&b, &q->y, &z1z1, FQ); mont_mul(&b, &b, &p->z, FQ);
        return u256_cmp(&a, &b) == 0;
    }
    mont_mul(&a, &p->x, &q->z, FQ); mont_mul(&b, &q->x, &p->z, FQ);
    if (u256_cmp(&a, &b)) return 0;
    mont_mul(&a, &p->y, &q->z, FQ); mont_mul(&b, &q->y, &p->z, FQ);
    return u256_cmp(&a, &b) == 0;
}
int te_on_curve(const te_aff *p, const suite_t *s) {
    u256 x2, y2, l, r;
    if (s->sw_native) {                                            /* y^2 = x^3 + a x + b */
        mont_sqr(&y2, &p->y, FQ); mont_sqr(&x2, &p->x, FQ); mont_add(&x2, &x2, &s->sw_a, FQ); mont_mul(&r, &x2, &p->x, FQ); mont_add(&r, &r, &s->sw_b, FQ);
        return u256_cmp(&y2, &r) == 0;
    }
    mont_sqr(&x2, &p->x, FQ); mont_sqr(&y2, &p->y, FQ);
    mont_mul(&l, &s->a, &x2, FQ); mont_add(&l, &l, &y2, FQ);
    mont_mul(&r, &x2, &y2, FQ); mont_mul(&r, &r, &s->d, FQ); mont_add(&r, &r, &s->fq.r1, FQ);
    return u256_cmp(&l, &r) == 0;
}
/* double-and-add, MSB first; scalar is a plain (non-Montgomery) integer */
void te_smul(te_ext *o, const te_aff *p, const u256 *k, const suite_t *s) {
    te_ext acc, pe; te_identity(&acc, s); te_from_aff(&pe, p, s);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) te_dbl(&acc, &acc, s);
        if ((k->l[i / 64] >> (i % 64)) & 1) { te_add(&acc, &acc, &pe, s); started = 1; }
    }
    *o = acc;
}
/* is_in_correct_subgroup_assuming_on_curve: r * P == 0 */
int te_in_subgroup(const te_aff *p, const suite_t *s) {
    te_ext e; te_smul(&e, p, &s->fr.p, s); return te_is_identity_ext(&e, s);
}

/* ---- codecs (ark-serialize CanonicalSerialize for TE affine) ---- */
static int fq_is_negative(const u256 *x_mont, const suite_t *s) { /* x > (q-1)/2 */
    u256 x; mont_from(&x, x_mont, FQ); return u256_cmp(&x, &s->fq.pm1_half) > 0;
}
void te_encode(uint8_t *out, const te_aff *p, const suite_t *s) {
    if (s->sw_native) { sw_encode(out, p, s); return; }
    u256 y; mont_from(&y, &p->y, FQ); u256_to_le(out, &y);
    if (fq_is_negative(&p->x, s)) out[31] |= 0x80;
}
int te_decode(te_aff *o, const uint8_t *in, const suite_t *s) {
    if (s->sw_native) return sw_decode(o, in, s);
    uint8_t b[32]; memcpy(b, in, 32);
    int neg = b[31] >> 7; b[31] &= 0x7f;
    u256 y; u256_from_le(&y, b);
    if (u256_cmp(&y, &s->fq.p) >= 0) return ORC_INVALID_DATA;
    mont_to(&o->y, &y, FQ);
    /* x^2 = (1 - y^2) / (a - d y^2) */
    u256 y2, num, den, x2, x;
    mont_sqr(&y2, &o->y, FQ);
    mont_sub(&num, &s->fq.r1, &y2, FQ);
    mont_mul(&den, &s->d, &y2, FQ); mont_sub(&den, &s->a, &den, FQ);
    if (u256_is_zero(&den)) return ORC_INVALID_DATA;
    mont_inv(&den, &den, FQ); mont_mul(&x2, &num, &den, FQ);
    if (!mont_sqrt(&x, &x2, FQ)) return ORC_INVALID_DATA;
    if (fq_is_negative(&x, s) != neg) mont_neg(&x, &x, FQ);
    /* x == 0 with the negative flag set is a non-canonical encoding */
    if (u256_is_zero(&x) && neg) return ORC_INVALID_DATA;
    o->x = x;
    return ORC_OK;
}
void te_encode_xy(uint8_t out[64], const te_aff *p, const suite_t *s) {
    u256 t; mont_from(&t, &p->x, FQ); u256_to_le(out, &t);
    mont_from(&t, &p->y, FQ); u256_to_le(out + 32, &t);
}
/* ---- SW presentation (src/utils/te_sw_map.rs:32-68; ark-serialize SWFlags: the flag byte follows the 32-byte x because
 * a 255-bit modulus leaves one spare bit and the two flags need two: bit 7 = y is the larger root, bit 6 = infinity) ---- */
static void sw_xy_from_te(u256 *x, u256 *y, const te_aff *p, const suite_t *s) {   /* te_to_sw; p not the identity, x != 0 */
    u256 one = s->fq.r1, vd, wd, num, v, w;
    mont_sub(&vd, &one, &p->y, FQ);                       /* 1 - y */
    mont_mul(&wd, &p->x, &vd, FQ);                        /* x (1 - y) */
    mont_inv(&vd, &vd, FQ); mont_inv(&wd, &wd, FQ);
    mont_add(&num, &one, &p->y, FQ);
    mont_mul(&v, &num, &vd, FQ); mont_mul(&w, &num, &wd, FQ);
    mont_add(&v, &v, &s->mont_a3, FQ); mont_mul(x, &s->mont_binv, &v, FQ);
    mont_mul(y, &s->mont_binv, &w, FQ);
}
static int te_from_sw_xy(te_aff *o, const u256 *x, const u256 *y, const suite_t *s) {   /* sw_to_te; 0 ok */
    u256 one = s->fq.r1, mx, my, t, w;
    mont_mul(&mx, &s->mont_b, x, FQ); mont_sub(&mx, &mx, &s->mont_a3, FQ); mont_mul(&my, &s->mont_b, y, FQ);
    mont_add(&t, &mx, &one, FQ);
    if (u256_is_zero(&my) || u256_is_zero(&t)) return ORC_INVALID_DATA;
    mont_inv(&my, &my, FQ); mont_mul(&o->x, &mx, &my, FQ);
    mont_inv(&t, &t, FQ); mont_sub(&w, &mx, &one, FQ); mont_mul(&o->y, &w, &t, FQ);
    return ORC_OK;
}
void sw_encode(uint8_t out[33], const te_aff *p, const suite_t *s) {
    memset(out, 0, 33);
    if (s->sw_native) {                                            /* the point IS the SW point: LE32(x) || flags */
        if (te_is_identity_aff(p, s)) { out[32] = 0x40; return; }
        u256 t; mont_from(&t, &p->x, FQ); u256_to_le(out, &t);
        if (fq_is_negative(&p->y, s)) out[32] = 0x80;
        return;
    }
    if (te_is_identity_aff(p, s) || u256_is_zero(&p->x)) { out[32] = 0x40; return; }     /* infinity (the maps are undefined there) */
    u256 x, y, t; sw_xy_from_te(&x, &y, p, s);
    mont_from(&t, &x, FQ); u256_to_le(out, &t);
    if (fq_is_negative(&y, s)) out[32] = 0x80;
}
int sw_from_x(te_aff *o, const u256 *x_plain, int greatest, const suite_t *s) {
    u256 x, rhs, t, y;
    if (u256_cmp(x_plain, &s->fq.p) >= 0) return ORC_INVALID_DATA;
    mont_to(&x, x_plain, FQ);
    mont_sqr(&t, &x, FQ); mont_add(&t, &t, &s->sw_a, FQ); mont_mul(&rhs, &t, &x, FQ); mont_add(&rhs, &rhs, &s->sw_b, FQ);
    if (!mont_sqrt(&y, &rhs, FQ)) return ORC_INVALID_DATA;
    if (fq_is_negative(&y, s) != (greatest != 0)) mont_neg(&y, &y, FQ);
    if (s->sw_native) { o->x = x; o->y = y; return ORC_OK; }
    return te_from_sw_xy(o, &x, &y, s);
}
int sw_decode(te_aff *o, const uint8_t in[33], const suite_t *s) {
    if (in[32] & 0x3f) return ORC_INVALID_DATA;
    if ((in[32] & 0x40) && s->sw_native) {              /* SWFlags::PointAtInfinity: x must be zero, no sign bit */
        for (int i = 0; i < 32; i++) if (in[i]) return ORC_INVALID_DATA;
        if (in[32] & 0x80) return ORC_INVALID_DATA;
        memset(o, 0, sizeof *o); return ORC_OK;
    }
    if (in[32] & 0x40) return ORC_INVALID_DATA;          /* infinity: no twisted-Edwards image (sw_to_te -> None) */
    u256 x; u256_from_le(&x, in);
    return sw_from_x(o, &x, (in[32] & 0x80) != 0, s);
}
int te_decode_xy(te_aff *o, const uint8_t in[64], const suite_t *s) {
    u256 x, y; u256_from_le(&x, in); u256_from_le(&y, in + 32);
    if (u256_cmp(&x, &s->fq.p) >= 0 || u256_cmp(&y, &s->fq.p) >= 0) return ORC_INVALID_DATA;
    mont_to(&o->x, &x, FQ); mont_to(&o->y, &y, FQ);
    return ORC_OK;
}
