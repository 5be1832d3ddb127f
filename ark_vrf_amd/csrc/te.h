// te.h -- twisted-Edwards group law on gfx950 lanes (extended coordinates, general a in {1,-5}).
//
// Device counterpart of arkworks `twisted_edwards::{Affine,Projective}` arithmetic
// (third-party ark-ec 0.6) as reached from src/thin.rs:119,158,319, src/pedersen.rs:148-167,
// 229-245,420, src/utils/straus.rs:28-42,92-98.  Formulas: add-2008-hwcd / madd-2008-hwcd /
// dbl-2008-hwcd.  A "pre" point is an affine point with k = d*x*y precomputed, so a mixed
// addition is 8 field multiplications; multiplication by a = -5 is three additions.
// (5 is a non-residue mod q_Bandersnatch, so no isomorphic a = -1 model exists over Fq.)
#pragma once
#include "fp256.h"

namespace avrf {

struct te_ext { fp x, y, t, z; };
struct te_pre { fp x, y, k; };  // affine + k = d*x*y  (96 bytes)
struct te_aff { fp x, y; };

template <class S> AVRF_DI fp mul_a(const fp &v) {
  using Fq = typename S::Fq;
  if (S::A_KIND == 1) {  // a = -5
    fp t = fp_dbl<Fq>(v); t = fp_dbl<Fq>(t); t = fp_add<Fq>(t, v); return fp_neg<Fq>(t);
  }
  if (S::A_KIND == 2) return fp_neg<Fq>(v);  // a = -1 (JubJub)
  return v;  // a = 1
}

// ---- S::SW_NATIVE (secp256r1, src/suites/secp256r1.rs:49-70; ark_ec::short_weierstrass arithmetic of ark-secp256r1): a curve
// y^2 = x^3 - 3x + b with no twisted-Edwards model.  The SAME types carry it through every kernel of the engine:
//   te_ext {x, y, t, z} = XYZZ coordinates (X, Y, ZZ, ZZZ): x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ = 0
//                         (so all-zero memory reads as the identity);
//   te_pre {x, y, k}    = the affine point (x, y), k unused; (0, 0) = the point at infinity (not on the curve: b != 0);
// and the te_* functions below dispatch to the sw_* forms.  The twisted-Edwards law is complete; these formulas are not, and the
// kernels rely on completeness (window tables add P to P, sums meet their negatives), so every exceptional case is handled here.
// The multiplications go through the OUT-OF-LINE multiplier (fp_mul_nf): these fields have no asm block, and forty inlined copies
// of the generic 128-statement form inside one non-kernel function (te_smul) produced code that faulted on the device -- the
// same function body inlined into its kernel, or calling one shared multiplier, is correct (tools/secp_probe.hip).
template <class S> AVRF_DI te_ext sw_identity() {
  te_ext r; r.x = fp_one<typename S::Fq>(); r.y = r.x; r.t = fp_zero(); r.z = fp_zero(); return r;
}
// 2 a (dbl-2008-s-1 with a = -3: M = 3 (X - ZZ)(X + ZZ)); the identity and points of order 2 come out with ZZ = 0
template <class S> AVRF_DI te_ext sw_dbl(const te_ext &a) {
  using Fq = typename S::Fq;
  fp U = fp_dbl<Fq>(a.y), V = fp_sqr_nf<Fq>(U), W = fp_mul_nf<Fq>(U, V), Sx = fp_mul_nf<Fq>(a.x, V);
  fp M = fp_mul_nf<Fq>(fp_sub<Fq>(a.x, a.t), fp_add<Fq>(a.x, a.t)); M = fp_add<Fq>(fp_dbl<Fq>(M), M);
  te_ext r;
  r.x = fp_sub<Fq>(fp_sqr_nf<Fq>(M), fp_dbl<Fq>(Sx));
  r.y = fp_sub<Fq>(fp_mul_nf<Fq>(M, fp_sub<Fq>(Sx, r.x)), fp_mul_nf<Fq>(W, a.y));
  r.t = fp_mul_nf<Fq>(V, a.t); r.z = fp_mul_nf<Fq>(W, a.z);
  return r;
}
template <class S> AVRF_DI te_ext sw_from_affine(const fp &x, const fp &y) {
  if (fp_is_zero(x) && fp_is_zero(y)) return sw_identity<S>();
  te_ext r; r.x = x; r.y = y; r.t = fp_one<typename S::Fq>(); r.z = r.t; return r;
}
// a + (x, y) (madd-2008-s: 8M + 2S).  Single exit: the exceptional cases (a or the base at infinity, equal or opposite points)
// overwrite the result of the main sequence, which is computed for every lane and harmless on their inputs.
template <class S> AVRF_DI te_ext sw_madd(const te_ext &a, const fp &qx, const fp &qy) {
  using Fq = typename S::Fq;
  const bool q_inf = fp_is_zero(qx) && fp_is_zero(qy), a_inf = fp_is_zero(a.t);
  fp P = fp_sub<Fq>(fp_mul_nf<Fq>(qx, a.t), a.x), R = fp_sub<Fq>(fp_mul_nf<Fq>(qy, a.z), a.y);
  te_ext r;
  fp PP = fp_sqr_nf<Fq>(P);
  r.t = fp_mul_nf<Fq>(a.t, PP);
  fp Q = fp_mul_nf<Fq>(a.x, PP), PPP = fp_mul_nf<Fq>(P, PP);
  r.z = fp_mul_nf<Fq>(a.z, PPP);
  fp T = fp_mul_nf<Fq>(a.y, PPP);
  r.x = fp_sub<Fq>(fp_sub<Fq>(fp_sqr_nf<Fq>(R), PPP), fp_dbl<Fq>(Q));
  r.y = fp_sub<Fq>(fp_mul_nf<Fq>(R, fp_sub<Fq>(Q, r.x)), T);
  if (q_inf) r = a;
  else if (a_inf) r = sw_from_affine<S>(qx, qy);
  else if (fp_is_zero(P)) { if (fp_is_zero(R)) r = sw_dbl<S>(sw_from_affine<S>(qx, qy)); else r = sw_identity<S>(); }
  return r;
}
// a + b (add-2008-s: 12M + 2S), same structure
template <class S> AVRF_DI te_ext sw_add(const te_ext &a, const te_ext &b) {
  using Fq = typename S::Fq;
  const bool a_inf = fp_is_zero(a.t), b_inf = fp_is_zero(b.t);
  fp U1 = fp_mul_nf<Fq>(a.x, b.t), P = fp_sub<Fq>(fp_mul_nf<Fq>(b.x, a.t), U1);
  fp S1 = fp_mul_nf<Fq>(a.y, b.z), R = fp_sub<Fq>(fp_mul_nf<Fq>(b.y, a.z), S1);
  te_ext r;
  fp PP = fp_sqr_nf<Fq>(P);
  r.t = fp_mul_nf<Fq>(fp_mul_nf<Fq>(a.t, b.t), PP);
  fp Q = fp_mul_nf<Fq>(U1, PP), PPP = fp_mul_nf<Fq>(P, PP);
  r.z = fp_mul_nf<Fq>(fp_mul_nf<Fq>(a.z, b.z), PPP);
  fp T = fp_mul_nf<Fq>(S1, PPP);
  r.x = fp_sub<Fq>(fp_sub<Fq>(fp_sqr_nf<Fq>(R), PPP), fp_dbl<Fq>(Q));
  r.y = fp_sub<Fq>(fp_mul_nf<Fq>(R, fp_sub<Fq>(Q, r.x)), T);
  if (a_inf) r = b;
  else if (b_inf) r = a;
  else if (fp_is_zero(P)) { if (fp_is_zero(R)) r = sw_dbl<S>(a); else r = sw_identity<S>(); }
  return r;
}

template <class S> AVRF_DI te_ext te_identity() {
  if constexpr (S::SW_NATIVE) return sw_identity<S>();
  te_ext r; r.x = fp_zero(); r.y = fp_one<typename S::Fq>(); r.t = fp_zero(); r.z = fp_one<typename S::Fq>(); return r;
}
template <class S> AVRF_DI bool te_is_identity(const te_ext &p) {
  if constexpr (S::SW_NATIVE) return fp_is_zero(p.t);
  return fp_is_zero(p.x) && fp_eq(p.y, p.z);
}

// p + q, q precomputed affine (8M)
template <class S> AVRF_DI te_ext te_madd(const te_ext &p, const te_pre &q) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) return sw_madd<S>(p, q.x, q.y);
  fp A = fp_mul<Fq>(p.x, q.x);
  fp B = fp_mul<Fq>(p.y, q.y);
  fp C = fp_mul<Fq>(p.t, q.k);
  fp E = fp_mul<Fq>(fp_add<Fq>(p.x, p.y), fp_add<Fq>(q.x, q.y));
  E = fp_sub<Fq>(fp_sub<Fq>(E, A), B);
  fp F = fp_sub<Fq>(p.z, C), G = fp_add<Fq>(p.z, C);
  fp H = fp_sub<Fq>(B, mul_a<S>(A));
  te_ext r;
  r.x = fp_mul<Fq>(E, F); r.y = fp_mul<Fq>(G, H); r.t = fp_mul<Fq>(E, H); r.z = fp_mul<Fq>(F, G);
  return r;
}
// p - q
template <class S> AVRF_DI te_pre te_pre_neg(const te_pre &q) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) { te_pre r; r.x = q.x; r.y = fp_neg<Fq>(q.y); r.k = q.k; return r; }
  te_pre r; r.x = fp_neg<Fq>(q.x); r.y = q.y; r.k = fp_neg<Fq>(q.k); return r;
}
// p + q, both extended (9M + 1 mul by d)
template <class S> AVRF_DI te_ext te_add(const te_ext &p, const te_ext &q) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) return sw_add<S>(p, q);
  fp A = fp_mul<Fq>(p.x, q.x);
  fp B = fp_mul<Fq>(p.y, q.y);
  fp C = fp_mul<Fq>(fp_mul<Fq>(p.t, q.t), fp_const<Fq>(S::D));
  fp D = fp_mul<Fq>(p.z, q.z);
  fp E = fp_mul<Fq>(fp_add<Fq>(p.x, p.y), fp_add<Fq>(q.x, q.y));
  E = fp_sub<Fq>(fp_sub<Fq>(E, A), B);
  fp F = fp_sub<Fq>(D, C), G = fp_add<Fq>(D, C);
  fp H = fp_sub<Fq>(B, mul_a<S>(A));
  te_ext r;
  r.x = fp_mul<Fq>(E, F); r.y = fp_mul<Fq>(G, H); r.t = fp_mul<Fq>(E, H); r.z = fp_mul<Fq>(F, G);
  return r;
}
// 2p (4M + 4S)
template <class S> AVRF_DI te_ext te_dbl(const te_ext &p) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) return sw_dbl<S>(p);
  fp A = fp_sqr<Fq>(p.x), B = fp_sqr<Fq>(p.y);
  fp C = fp_dbl<Fq>(fp_sqr<Fq>(p.z));
  fp D = mul_a<S>(A);
  fp E = fp_sub<Fq>(fp_sub<Fq>(fp_sqr<Fq>(fp_add<Fq>(p.x, p.y)), A), B);
  fp G = fp_add<Fq>(D, B), F = fp_sub<Fq>(G, C), H = fp_sub<Fq>(D, B);
  te_ext r;
  r.x = fp_mul<Fq>(E, F); r.y = fp_mul<Fq>(G, H); r.t = fp_mul<Fq>(E, H); r.z = fp_mul<Fq>(F, G);
  return r;
}
template <class S> AVRF_DI te_ext te_from_pre(const te_pre &q) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) return sw_from_affine<S>(q.x, q.y);
  te_ext r; r.x = q.x; r.y = q.y; r.t = fp_mul<Fq>(q.x, q.y); r.z = fp_one<Fq>(); return r;
}
template <class S> AVRF_DI te_pre te_make_pre(const fp &x_mont, const fp &y_mont) {
  using Fq = typename S::Fq;
  te_pre r; r.x = x_mont; r.y = y_mont;
  if constexpr (S::SW_NATIVE) { r.k = fp_zero(); return r; }
  r.k = fp_mul<Fq>(fp_mul<Fq>(x_mont, y_mont), fp_const<Fq>(S::D));
  return r;
}
template <class S> AVRF_DI te_aff te_to_aff(const te_ext &p) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) {                        // 1 / (ZZ ZZZ) gives both 1 / ZZ and 1 / ZZZ; the identity -> (0, 0)
    fp i = fp_inv<Fq>(fp_mul<Fq>(p.t, p.z));           // (0^(p-2) = 0)
    te_aff r; r.x = fp_mul<Fq>(p.x, fp_mul<Fq>(i, p.z)); r.y = fp_mul<Fq>(p.y, fp_mul<Fq>(i, p.t)); return r;
  }
  fp zi = fp_inv<Fq>(p.z);
  te_aff r; r.x = fp_mul<Fq>(p.x, zi); r.y = fp_mul<Fq>(p.y, zi); return r;
}
// a*x^2 + y^2 == 1 + d*x^2*y^2
template <class S> AVRF_DI bool te_on_curve(const fp &x, const fp &y) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE)                          // y^2 = x^3 + a x + b, or the point at infinity (0, 0)
    return (fp_is_zero(x) && fp_is_zero(y)) || fp_eq(fp_sqr<Fq>(y), fp_add<Fq>(fp_mul<Fq>(fp_add<Fq>(fp_sqr<Fq>(x), fp_const<Fq>(S::SW_A)), x), fp_const<Fq>(S::SW_B)));
  fp x2 = fp_sqr<Fq>(x), y2 = fp_sqr<Fq>(y);
  fp l = fp_add<Fq>(mul_a<S>(x2), y2);
  fp r = fp_add<Fq>(fp_one<Fq>(), fp_mul<Fq>(fp_mul<Fq>(x2, y2), fp_const<Fq>(S::D)));
  return fp_eq(l, r);
}

// out-of-line forms (per-item protocol kernels)
template <class S> AVRF_DN te_ext te_madd_nf(te_ext p, te_pre q) { return te_madd<S>(p, q); }
template <class S> AVRF_DN te_ext te_add_nf(te_ext p, te_ext q) { return te_add<S>(p, q); }
template <class S> AVRF_DN te_ext te_dbl_nf(te_ext p) { return te_dbl<S>(p); }
template <class S> AVRF_DN te_aff te_to_aff_nf(te_ext p) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) {
    fp i = fp_inv_nf<Fq>(fp_mul_nf<Fq>(p.t, p.z));
    te_aff r; r.x = fp_mul_nf<Fq>(p.x, fp_mul_nf<Fq>(i, p.z)); r.y = fp_mul_nf<Fq>(p.y, fp_mul_nf<Fq>(i, p.t)); return r;
  }
  fp zi = fp_inv_nf<Fq>(p.z);
  te_aff r; r.x = fp_mul_nf<Fq>(p.x, zi); r.y = fp_mul_nf<Fq>(p.y, zi); return r;
}
template <class S> AVRF_DN te_pre te_make_pre_nf(fp x_mont, fp y_mont) { return te_make_pre<S>(x_mont, y_mont); }

// raw load/store of points as 32-bit words (global memory, 16-byte vectorised)
AVRF_DI void load_words(uint32_t *dst, const uint32_t *src, int nwords) {
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  for (int i = 0; i < nwords / 4; i++) { uint4 v = s4[i]; dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w; }
}
AVRF_DI void store_words(uint32_t *dst, const uint32_t *src, int nwords) {
  uint4 *d4 = reinterpret_cast<uint4 *>(dst);
  for (int i = 0; i < nwords / 4; i++) d4[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}
AVRF_DI te_pre load_pre(const te_pre *p) {
  te_pre r; const uint4 *s = reinterpret_cast<const uint4 *>(p);
  uint4 a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3], a4 = s[4], a5 = s[5];
  r.x.v[0] = a0.x; r.x.v[1] = a0.y; r.x.v[2] = a0.z; r.x.v[3] = a0.w; r.x.v[4] = a1.x; r.x.v[5] = a1.y; r.x.v[6] = a1.z; r.x.v[7] = a1.w;
  r.y.v[0] = a2.x; r.y.v[1] = a2.y; r.y.v[2] = a2.z; r.y.v[3] = a2.w; r.y.v[4] = a3.x; r.y.v[5] = a3.y; r.y.v[6] = a3.z; r.y.v[7] = a3.w;
  r.k.v[0] = a4.x; r.k.v[1] = a4.y; r.k.v[2] = a4.z; r.k.v[3] = a4.w; r.k.v[4] = a5.x; r.k.v[5] = a5.y; r.k.v[6] = a5.z; r.k.v[7] = a5.w;
  return r;
}
AVRF_DI void store_fp(uint32_t *d, const fp &a) {
  uint4 *d4 = reinterpret_cast<uint4 *>(d);
  d4[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]); d4[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
AVRF_DI fp load_fp(const uint32_t *s) {
  const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
  uint4 a = s4[0], b = s4[1]; fp r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
AVRF_DI void store_pre(te_pre *p, const te_pre &v) {
  uint32_t *d = reinterpret_cast<uint32_t *>(p);
  store_fp(d, v.x); store_fp(d + 8, v.y); store_fp(d + 16, v.k);
}
AVRF_DI void store_ext(te_ext *p, const te_ext &v) {
  uint32_t *d = reinterpret_cast<uint32_t *>(p);
  store_fp(d, v.x); store_fp(d + 8, v.y); store_fp(d + 16, v.t); store_fp(d + 24, v.z);
}
AVRF_DI te_ext load_ext(const te_ext *p) {
  const uint32_t *s = reinterpret_cast<const uint32_t *>(p);
  te_ext r; r.x = load_fp(s); r.y = load_fp(s + 8); r.t = load_fp(s + 16); r.z = load_fp(s + 24); return r;
}

}  // namespace avrf
