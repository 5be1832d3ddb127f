// glv.h -- GLV scalar multiplication for the per-item kernels (vrf_single.hip) on suites whose curve carries an efficient
// endomorphism (S::HAS_GLV; Bandersnatch: psi of degree 2 with psi^2 = [-2], Masson-Sanso-Zhang 2021).
//
//   k P = k1 P + k2 psi(P),   k = k1 + k2 lambda (mod r),  |k1|, |k2| < 2^127
//
// so a 253-bit multiplication is ONE chain of 128 doublings over a joint 2-bit window table {i P + j psi(P)}.  Any correct
// evaluation gives the same group element, so the provers stay byte-exact and the verifiers' verdicts unchanged
// (src/thin.rs:119,158-161, src/pedersen.rs:164,229-232 only fix WHICH element is computed).  Constants (lattice basis,
// rounding multipliers, the two coefficients of psi in twisted-Edwards coordinates) come from tools/gen_consts.py, which
// re-derives and checks them (psi(G) = [lambda] G, |k_i| < 2^127 on 20 000 scalars).
// The Pippenger path (msm.hip) does not use this: a split doubles the terms and halves the windows, the bucket additions stay.
#pragma once
#include "proto_dev.h"

namespace avrf {

// o = a * b (operand scanning; a few hundred instructions per decomposition, off the hot loop)
template <int NA, int NB> AVRF_DI void mp_mul(uint32_t (&o)[NA + NB], const uint32_t (&a)[NA], const uint32_t (&b)[NB]) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) o[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) { uint64_t t = (uint64_t)a[i] * b[j] + o[i + j] + carry; o[i + j] = (uint32_t)t; carry = t >> 32; }
    o[i + NB] = (uint32_t)carry;
  }
}
template <int N> AVRF_DI void mp_sub(uint32_t (&o)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {   // mod 2^(32 N)
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { c += (int64_t)a[i] - (int64_t)b[i]; o[i] = (uint32_t)c; c >>= 32; }
}

struct glv_scalars { fp k1, k2; bool n1, n2; };   // magnitudes (plain integers < 2^127) and signs

// c1 = (k g1 + 2^255) >> 256, c2 = (k g2 + 2^255) >> 256, k1 = k - c1 a1 - c2 a2, k2 = c1 |b1| - c2 b2  (tools/gen_consts.py)
template <class S> AVRF_DI glv_scalars glv_decompose(const fp &k) {
  uint32_t kk[8], g1[4], g2[5], a1[4], a2[4], b1[4], b2[4];
#pragma unroll
  for (int i = 0; i < 8; i++) kk[i] = k.v[i];
#pragma unroll
  for (int i = 0; i < 4; i++) { g1[i] = S::GLV_G1[i]; a1[i] = S::GLV_A1[i]; a2[i] = S::GLV_A2[i]; b1[i] = S::GLV_B1N[i]; b2[i] = S::GLV_B2[i]; }
#pragma unroll
  for (int i = 0; i < 5; i++) g2[i] = S::GLV_G2[i];
  uint32_t p1[12], p2[13], c1[4], c2[5];
  mp_mul<8, 4>(p1, kk, g1);
  mp_mul<8, 5>(p2, kk, g2);
  { uint64_t c = (uint64_t)p1[7] + 0x80000000u; c >>= 32;
#pragma unroll
    for (int i = 0; i < 4; i++) { c += p1[8 + i]; c1[i] = (uint32_t)c; c >>= 32; } }
  { uint64_t c = (uint64_t)p2[7] + 0x80000000u; c >>= 32;
#pragma unroll
    for (int i = 0; i < 5; i++) { c += p2[8 + i]; c2[i] = (uint32_t)c; c >>= 32; } }
  uint32_t c1a1[8], c2a2[9], c1b1[8], c2b2[9];
  mp_mul<4, 4>(c1a1, c1, a1); mp_mul<5, 4>(c2a2, c2, a2);
  mp_mul<4, 4>(c1b1, c1, b1); mp_mul<5, 4>(c2b2, c2, b2);
  uint32_t x[9], y[9], t[9], k1[9], k2[9];
#pragma unroll
  for (int i = 0; i < 8; i++) { x[i] = kk[i]; y[i] = c1a1[i]; }
  x[8] = 0; y[8] = 0;
  mp_sub<9>(t, x, y); mp_sub<9>(k1, t, c2a2);
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = c1b1[i];
  mp_sub<9>(k2, x, c2b2);
  glv_scalars r;
  r.n1 = (k1[8] >> 31) != 0; r.n2 = (k2[8] >> 31) != 0;
  uint32_t z[9];
#pragma unroll
  for (int i = 0; i < 9; i++) z[i] = 0;
  if (r.n1) { mp_sub<9>(t, z, k1);
#pragma unroll
    for (int i = 0; i < 9; i++) k1[i] = t[i]; }
  if (r.n2) { mp_sub<9>(t, z, k2);
#pragma unroll
    for (int i = 0; i < 9; i++) k2[i] = t[i]; }
#pragma unroll
  for (int i = 0; i < 8; i++) { r.k1.v[i] = i < 4 ? k1[i] : 0u; r.k2.v[i] = i < 4 ? k2[i] : 0u; }
  return r;
}

// psi(P) for affine P = (x, y): (c (1 - y^2) / (x y), b (y^2 + b) / (y^2 - b)), as an extended point over the common
// denominator x y (y^2 - b): 1S + 7M.  false for the handful of points the map has no finite image for in these coordinates
// (x y = 0: order <= 4; y^2 = b) -- the callers then take the plain path.
template <class S> AVRF_DI bool te_endo(const te_pre &p, te_ext &r) {
  using Fq = typename S::Fq;
  const fp b = fp_const<Fq>(S::ENDO_B), c = fp_const<Fq>(S::ENDO_C);
  const fp u = fp_sqr<Fq>(p.y), xy = fp_mul<Fq>(p.x, p.y);
  const fp bq = fp_sub<Fq>(u, b);
  if (fp_is_zero(xy) || fp_is_zero(bq)) return false;
  const fp ca = fp_mul<Fq>(c, fp_sub<Fq>(fp_one<Fq>(), u)), cq = fp_mul<Fq>(b, fp_add<Fq>(u, b));
  r.x = fp_mul<Fq>(ca, bq); r.y = fp_mul<Fq>(cq, xy); r.z = fp_mul<Fq>(xy, bq); r.t = fp_mul<Fq>(ca, cq);
  return true;
}
// Prime-order-subgroup membership of a point ON THE CURVE from its y alone (Montgomery form), for curves with
// E(Fq) = Z2 x Z2 x Zr (S::HAS_2DESCENT; Bandersnatch): the subgroup is 2 E(Fq), and a point is a double iff the three values
// B u, B (u - alpha), B (u - beta) of the complete 2-descent are squares (u = (1 + y) / (1 - y) on the Montgomery model
// B v^2 = u (u - alpha)(u - beta); their product is a square on the curve, so two symbols decide): modulo squares
//   chi(1 - y^2) = chi(B)   and   chi((c0 + c1 y)(1 - y)) = chi(B),   c0 = 1 - alpha, c1 = 1 + alpha
// (tools/gen_consts.py two_descent derives the constants and checks the criterion against r P on every coset).  Two Jacobi
// symbols in one loop (fp256.h fp_jacobi2_nf), ~16 k instructions, instead of the 253-bit scalar multiplication r P (~540 k) that src/lib.rs:410-433's
// is_in_correct_subgroup_assuming_on_curve amounts to.  The identity (y = 1) is in the subgroup; (0, -1) gives chi = 0: rejected.
template <class S> AVRF_DI bool te_in_subgroup_2descent(const fp &ym) {
  using Fq = typename S::Fq;
  const fp one = fp_one<Fq>();
  if (fp_eq(ym, one)) return true;
  const fp omy = fp_sub<Fq>(one, ym);
  const fp n1 = fp_mul_nf<Fq>(omy, fp_add<Fq>(one, ym));
  const fp n2 = fp_mul_nf<Fq>(fp_add<Fq>(fp_const<Fq>(S::TD_C0), fp_mul_nf<Fq>(fp_const<Fq>(S::TD_C1), ym)), omy);
  int j1, j2;
  fp_jacobi2_nf<Fq>(n1, n2, &j1, &j2);
  return j1 == S::TD_WANT && j2 == S::TD_WANT;
}

template <class S> AVRF_DI te_ext te_ext_neg(const te_ext &p) {
  using Fq = typename S::Fq;
  te_ext r; r.x = fp_neg<Fq>(p.x); r.y = p.y; r.t = fp_neg<Fq>(p.t); r.z = p.z; return r;
}

// tab[4 j + i] = i P + j Q  (i, j < 4), P affine, Q extended: 12 mixed + 2 general additions + 1 doubling
template <class S> AVRF_DI void glv_table(te_ext (&tab)[16], const te_pre &p, const te_ext &q) {
  tab[0] = te_identity<S>(); tab[4] = q; tab[8] = te_dbl<S>(q); tab[12] = te_add<S>(tab[8], q);
  for (int j = 0; j < 4; j++) for (int i = 1; i < 4; i++) tab[4 * j + i] = te_madd<S>(tab[4 * j + i - 1], p);
}
AVRF_DI uint32_t digit2(const fp &k, int w) { return (k.v[w >> 4] >> (2 * (w & 15))) & 3u; }

// the same table written to the item's workspace slots (proto_dev.h te_smul_ws): tab[4 j + i] = i P + j Q
template <class S> AVRF_DI void glv_table_ws(te_ext *tab, const te_pre &p, const te_ext &q) {
  te_ext row = q;
  for (int j = 0; j < 4; j++) {
    if (j == 2) row = te_dbl<S>(q); else if (j == 3) row = te_add<S>(row, q);
    te_ext cur = j ? row : te_identity<S>();
    if (j) store_ext(tab + 4 * j, cur);
    for (int i = 1; i < 4; i++) { cur = te_madd<S>(cur, p); store_ext(tab + 4 * j + i, cur); }
  }
}
// k * P with the table in the workspace (ws: the ITEM_TAB_SLOTS entries of this item; never null -- every caller sizes BatchDev::tabs first)
template <class S> AVRF_DN te_ext te_smul_glv_ws_nf(te_ext *ws, te_pre p, fp k) {
  using Fr = typename S::Fr;
  te_ext q;
  if (!te_endo<S>(p, q)) return te_smul_ws<S>(ws, p, k, Fr::BITS);
  const glv_scalars g = glv_decompose<S>(k);
  if (g.n1) p = te_pre_neg<S>(p);
  if (g.n2) q = te_ext_neg<S>(q);
  glv_table_ws<S>(ws, p, q);
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = 63; w >= 0; w--) {
    const uint32_t d = 4 * digit2(g.k2, w) + digit2(g.k1, w);
    te_ext e; if (d) e = load_ext(ws + d);
    acc = CH::template dbln<2>(acc);
    if (d) acc = CH::add(acc, e);
  }
  return CH::finish(acc);
}
template <class S, bool HAVE_Q> AVRF_DN te_ext te_smul_multi_glv_ws_nf(te_ext *ws, te_pre p, fp a, te_pre q, fp b, te_pre r, fp c) {
  using Fr = typename S::Fr;
  te_ext pe, qe;
  bool ok = te_endo<S>(p, pe);
  if (HAVE_Q) ok = te_endo<S>(q, qe) && ok;
  if (!ok) {                                                        // a degenerate point: the plain forms
    te_ext acc = HAVE_Q ? te_smul2_ws<S>(ws, p, a, q, b, Fr::BITS) : te_smul_ws<S>(ws, p, a, Fr::BITS);
    return te_add<S>(acc, te_smul_ws<S>(ws + 16, r, c, 128));
  }
  const glv_scalars ga = glv_decompose<S>(a);
  if (ga.n1) p = te_pre_neg<S>(p);
  if (ga.n2) pe = te_ext_neg<S>(pe);
  te_ext *tp = ws, *tq = ws + 16, *tr = ws + 32;
  glv_table_ws<S>(tp, p, pe);
  glv_scalars gb = ga;
  if (HAVE_Q) {
    gb = glv_decompose<S>(b);
    if (gb.n1) q = te_pre_neg<S>(q);
    if (gb.n2) qe = te_ext_neg<S>(qe);
    glv_table_ws<S>(tq, q, qe);
  }
  { te_ext cur = te_from_pre<S>(r); store_ext(tr + 1, cur); cur = te_madd<S>(cur, r); store_ext(tr + 2, cur); cur = te_madd<S>(cur, r); store_ext(tr + 3, cur); }
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = 63; w >= 0; w--) {
    const uint32_t d = 4 * digit2(ga.k2, w) + digit2(ga.k1, w), e = HAVE_Q ? 4 * digit2(gb.k2, w) + digit2(gb.k1, w) : 0u, f = digit2(c, w);
    acc = CH::template dbln<2>(acc);
    // one inlined addition per chain function: the window's (up to) three entries go through a loop of one body
#pragma unroll 1
    for (int which = 0; which < 3; which++) {
      const uint32_t dg = which == 0 ? d : which == 1 ? e : f;
      if (!dg) continue;
      const te_ext en = load_ext((which == 0 ? tp : which == 1 ? tq : tr) + dg);
      acc = CH::add(acc, en);
    }
  }
  return CH::finish(acc);
}

// k * P, k a plain integer < r.  Suites without the endomorphism: the 4-bit window form.
template <class S> AVRF_DN te_ext te_smul_glv_nf(te_pre p, fp k) {
  using Fr = typename S::Fr;
  te_ext q;
  if (!te_endo<S>(p, q)) return te_smul<S>(p, k, Fr::BITS);
  const glv_scalars g = glv_decompose<S>(k);
  if (g.n1) p = te_pre_neg<S>(p);
  if (g.n2) q = te_ext_neg<S>(q);
  te_ext tab[16];
  glv_table<S>(tab, p, q);
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = 63; w >= 0; w--) {
    acc = CH::template dbln<2>(acc);
    const uint32_t d = 4 * digit2(g.k2, w) + digit2(g.k1, w);
    if (d) acc = CH::add(acc, tab[d]);
  }
  return CH::finish(acc);
}
template <class S> AVRF_DI te_ext te_smul_glv(const te_pre &p, const fp &k) {
  if constexpr (S::HAS_GLV) return te_smul_glv_nf<S>(p, k);
  else return te_smul<S>(p, k, S::Fr::BITS);
}
template <class S> AVRF_DI te_ext te_smul_glv(te_ext *ws, const te_pre &p, const fp &k) {
  if constexpr (S::HAS_GLV) return te_smul_glv_ws_nf<S>(ws, p, k);
  else return te_smul_ws<S>(ws, p, k, S::Fr::BITS);
}

// a * P [+ b * Q] + c * R with a, b plain integers < r and c < 2^128: one chain of 128 doublings over the GLV tables of P
// (and Q) and the multiples {R, 2R, 3R}.  The verifiers' equations: Thin / Tiny  (s z) I - (c z) O - c pk,  Pedersen  s I - c O.
template <class S, bool HAVE_Q> AVRF_DN te_ext te_smul_multi_glv_nf(te_pre p, fp a, te_pre q, fp b, te_pre r, fp c) {
  using Fr = typename S::Fr;
  te_ext pe, qe;
  bool ok = te_endo<S>(p, pe);
  if (HAVE_Q) ok = te_endo<S>(q, qe) && ok;
  if (!ok) {                                                        // a degenerate point: the plain forms
    te_ext acc = HAVE_Q ? te_smul2<S>(p, a, q, b, Fr::BITS) : te_smul<S>(p, a, Fr::BITS);
    return te_add<S>(acc, te_smul<S>(r, c, 128));
  }
  const glv_scalars ga = glv_decompose<S>(a);
  if (ga.n1) p = te_pre_neg<S>(p);
  if (ga.n2) pe = te_ext_neg<S>(pe);
  te_ext tp[16], tq[16], tr[4];
  glv_table<S>(tp, p, pe);
  glv_scalars gb = ga;
  if (HAVE_Q) {
    gb = glv_decompose<S>(b);
    if (gb.n1) q = te_pre_neg<S>(q);
    if (gb.n2) qe = te_ext_neg<S>(qe);
    glv_table<S>(tq, q, qe);
  }
  tr[0] = te_identity<S>(); tr[1] = te_from_pre<S>(r); tr[2] = te_madd<S>(tr[1], r); tr[3] = te_madd<S>(tr[2], r);
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = 63; w >= 0; w--) {
    acc = CH::template dbln<2>(acc);
    const uint32_t d = 4 * digit2(ga.k2, w) + digit2(ga.k1, w), e = HAVE_Q ? 4 * digit2(gb.k2, w) + digit2(gb.k1, w) : 0u, f = digit2(c, w);
#pragma unroll 1
    for (int which = 0; which < 3; which++) {
      const uint32_t dg = which == 0 ? d : which == 1 ? e : f;
      if (!dg) continue;
      acc = CH::add(acc, which == 0 ? tp[dg] : which == 1 ? tq[dg] : tr[dg]);
    }
  }
  return CH::finish(acc);
}
template <class S, bool HAVE_Q> AVRF_DI te_ext te_smul_multi_glv(te_ext *ws, const te_pre &p, const fp &a, const te_pre &q, const fp &b, const te_pre &r, const fp &c);
template <class S, bool HAVE_Q> AVRF_DI te_ext te_smul_multi_glv(const te_pre &p, const fp &a, const te_pre &q, const fp &b, const te_pre &r, const fp &c) {
  if constexpr (S::HAS_GLV) return te_smul_multi_glv_nf<S, HAVE_Q>(p, a, q, b, r, c);
  else {                                                            // exactly the pre-GLV forms
    te_ext acc = HAVE_Q ? te_smul2<S>(p, a, q, b, S::Fr::BITS) : te_smul2<S>(p, a, r, c, S::Fr::BITS);
    return HAVE_Q ? te_add<S>(acc, te_smul<S>(r, c, 128)) : acc;
  }
}

template <class S, bool HAVE_Q> AVRF_DI te_ext te_smul_multi_glv(te_ext *ws, const te_pre &p, const fp &a, const te_pre &q, const fp &b, const te_pre &r, const fp &c) {
  if constexpr (S::HAS_GLV) return te_smul_multi_glv_ws_nf<S, HAVE_Q>(ws, p, a, q, b, r, c);
  else {
    te_ext acc = HAVE_Q ? te_smul2_ws<S>(ws, p, a, q, b, S::Fr::BITS) : te_smul2_ws<S>(ws, p, a, r, c, S::Fr::BITS);
    return HAVE_Q ? te_add<S>(acc, te_smul_ws<S>(ws + 16, r, c, 128)) : acc;
  }
}

}  // namespace avrf
