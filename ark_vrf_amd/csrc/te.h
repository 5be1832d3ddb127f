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

template <class S> AVRF_DI te_ext te_identity() {
  te_ext r; r.x = fp_zero(); r.y = fp_one<typename S::Fq>(); r.t = fp_zero(); r.z = fp_one<typename S::Fq>(); return r;
}
template <class S> AVRF_DI bool te_is_identity(const te_ext &p) { return fp_is_zero(p.x) && fp_eq(p.y, p.z); }

// p + q, q precomputed affine (8M)
template <class S> AVRF_DI te_ext te_madd(const te_ext &p, const te_pre &q) {
  using Fq = typename S::Fq;
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
  te_pre r; r.x = fp_neg<Fq>(q.x); r.y = q.y; r.k = fp_neg<Fq>(q.k); return r;
}
// p + q, both extended (9M + 1 mul by d)
template <class S> AVRF_DI te_ext te_add(const te_ext &p, const te_ext &q) {
  using Fq = typename S::Fq;
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
  te_ext r; r.x = q.x; r.y = q.y; r.t = fp_mul<Fq>(q.x, q.y); r.z = fp_one<Fq>(); return r;
}
template <class S> AVRF_DI te_pre te_make_pre(const fp &x_mont, const fp &y_mont) {
  using Fq = typename S::Fq;
  te_pre r; r.x = x_mont; r.y = y_mont;
  r.k = fp_mul<Fq>(fp_mul<Fq>(x_mont, y_mont), fp_const<Fq>(S::D));
  return r;
}
template <class S> AVRF_DI te_aff te_to_aff(const te_ext &p) {
  using Fq = typename S::Fq;
  fp zi = fp_inv<Fq>(p.z);
  te_aff r; r.x = fp_mul<Fq>(p.x, zi); r.y = fp_mul<Fq>(p.y, zi); return r;
}
// a*x^2 + y^2 == 1 + d*x^2*y^2
template <class S> AVRF_DI bool te_on_curve(const fp &x, const fp &y) {
  using Fq = typename S::Fq;
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
