// fpu_g1.h -- the short-Weierstrass (a = 0, XYZZ) mixed addition of k_accumulate<G1Curve> on unsaturated limbs (fpu.h):
// 14 x 28-bit limbs for the BLS12-381 base field, 9 x 29 for BN254's.  Formulas: madd-2008-s (8M + 2S), as curves.h.
//
// Montgomery domains.  fu_mul divides by R' = 2^(W L) = R 2^SH (R = 2^(32 N), the radix of every table in HBM).  XYZZ
// coordinates are only WEIGHTED-homogeneous -- (X, Y, ZZ, ZZZ) ~ (l^2 X, l^3 Y, l^2 ZZ, l^3 ZZZ) -- so the uniform 2^-SH of
// fpu_te.h cannot be absorbed as it is.  Instead every coordinate carries its own constant: the accumulator holds
//      (A X, B Y, G ZZ, D ZZZ),   A = R 2^a, B = R 2^b, G = R 2^g, D = R 2^d,
// a base is sliced as x R 2^sx, y R 2^sy, and the formulas close on the same class of points iff
//      G 2^sx R / R' = A,   D 2^sy R / R' = B   (U2 = x ZZ and S2 = y ZZZ meet X and Y),     B^2 = A^3 / R'   (R^2 meets P^3)
// <=>  g + sx = a + SH,     d + sy = b + SH,     3a - 2b = SH:   each mixed addition then multiplies the representative by
// l = A / R'.  Taking g = 2m, d = 3m (m = 2) makes ZZ and ZZZ exactly those of the representative (2^2m X', 2^3m Y', ...)
// whose X and Y need the constants 2^(2m - a), 2^(3m - b) -- applied once per PARTIAL SUM by its reader (one constant
// multiplication for Y; for X too when a is odd), never per addition.   BLS12-381: SH = 8, (a, b) = (4, 2), shifts (8, 4).
// BN254: SH = 5, (a, b) = (3, 2), shifts (4, 1).   An affine point enters as plain slices (x R 2^a, y R 2^b, G, D).
//
// Exceptional cases (the law is not complete): the identity is a flag; a base at infinity is all-zero words; P = +-Q shows
// as PP = P^2 / R' = 0 mod p, and a Montgomery product of operands with |P|^2 < R' p is 0 mod p iff its limbs are all zero
// or exactly p's (unique representation: limbs 0 .. L-2 in [0, 2^W)) -- tested on limb 0 first, in full only when a lane of
// the wave passes that; the doubling itself runs in the saturated form (G1Curve::dbl_affine) and is converted.
//
// Bounds: tools/fpu_model.py (inductive |X|, |Y|, |ZZ|, |ZZZ| < 4.6p, 2.6p, 1.5p, 1.5p; entry states fall into it).
#pragma once
#include "fpu.h"

namespace avrf {

template <class C> struct G1U {
  using Fq = typename C::Fq; using U = UL<Fq>;
  static constexpr int SH = U::SH;
  static constexpr int a = SH == 8 ? 4 : 3, b = 2, m = 2, g = 2 * m, d = 3 * m;
  static constexpr int sx = a + SH - g, sy = b + SH - d;
  static constexpr int kx = 2 * m - a, ky = 3 * m - b;              // X 2^kx, Y 2^ky at the reader
  static_assert(3 * a - 2 * b == SH && sx >= 0 && sx <= SH && sy >= 0 && sy <= SH && kx >= 0 && ky >= 0, "fpu_g1.h: scaling constants");
  static constexpr ulimbs<U::L> ZZ1 = U::shl_mod(Fq::ONE, g);       // R 2^g mod p: ZZ of an affine point
  static constexpr ulimbs<U::L> ZZZ1 = U::shl_mod(Fq::ONE, d);
  static constexpr ulimbs<U::L> CX = U::shl_mod(Fq::ONE, SH + kx);  // R' 2^kx mod p: fu_mul by it multiplies by 2^kx
  static constexpr ulimbs<U::L> CY = U::shl_mod(Fq::ONE, SH + ky);
  static constexpr ulimbs<U::L> CA = U::shl_mod(Fq::ONE, SH + a);   // saturated X -> accumulator X (doubling branch)
  static constexpr ulimbs<U::L> CB = U::shl_mod(Fq::ONE, SH + b);
  static constexpr ulimbs<U::L> CG = U::shl_mod(Fq::ONE, SH + g);
  static constexpr ulimbs<U::L> CD = U::shl_mod(Fq::ONE, SH + d);
};

template <class C> struct g1_acc_u {
  fuF<typename C::Fq> x, y, zz, zzz;
  uint32_t inf;                                 // != 0: the identity (coordinates meaningless)
};

template <class C> AVRF_DI g1_acc_u<C> g1u_identity() {
  constexpr int L = UL<typename C::Fq>::L;
  g1_acc_u<C> r; r.x = fu_zero<L>(); r.y = r.x; r.zz = r.x; r.zzz = r.x; r.inf = 1u; return r;
}
template <int N> AVRF_DI bool words_zero(const uint32_t (&w)[N]) { uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= w[i];
  return o == 0; }
// the affine base (x R, y R) itself; (0, 0) = infinity
template <class C> AVRF_DI g1_acc_u<C> g1u_from_affine(const uint32_t (&qx)[C::Fq::N], const uint32_t (&qy)[C::Fq::N], bool neg) {
  using Fq = typename C::Fq; using K = G1U<C>; constexpr int L = UL<Fq>::L;
  g1_acc_u<C> r;
  r.x = fu_slice<Fq, K::a>(qx);
  r.y = fu_cneg<L>(fu_slice<Fq, K::b>(qy), neg ? -1 : 0);
  r.zz = fu_const<Fq>(K::ZZ1); r.zzz = fu_const<Fq>(K::ZZZ1);
  r.inf = (words_zero(qx) && words_zero(qy)) ? 1u : 0u;
  return r;
}
// a canonical saturated XYZZ point in the accumulator's scaling (rare: the doubling branch)
template <class C> AVRF_DI g1_acc_u<C> g1u_from_xyzz(const uint32_t (&X)[C::Fq::N], const uint32_t (&Y)[C::Fq::N], const uint32_t (&ZZ)[C::Fq::N], const uint32_t (&ZZZ)[C::Fq::N]) {
  using Fq = typename C::Fq; using K = G1U<C>;
  g1_acc_u<C> r;
  r.x = fu_mul<Fq>(fu_slice<Fq, 0>(X), fu_const<Fq>(K::CA)); r.y = fu_mul<Fq>(fu_slice<Fq, 0>(Y), fu_const<Fq>(K::CB));
  r.zz = fu_mul<Fq>(fu_slice<Fq, 0>(ZZ), fu_const<Fq>(K::CG)); r.zzz = fu_mul<Fq>(fu_slice<Fq, 0>(ZZZ), fu_const<Fq>(K::CD));
  r.inf = words_zero(ZZ) ? 1u : 0u;
  return r;
}
// limbs of a Montgomery product == 0 mod p (see the header): all zero, or p's
template <class F> AVRF_DI bool fu_is_zero_mod_p(const fuF<F> &v) {
  using U = UL<F>;
  uint32_t z = 0, zp = 0;
#pragma unroll
  for (int i = 0; i < U::L; i++) { z |= (uint32_t)v.v[i]; zp |= (uint32_t)v.v[i] ^ U::P1.v[i]; }
  return z == 0 || zp == 0;
}
template <class F> AVRF_DI bool fu_maybe_zero_mod_p(const fuF<F> &v) {
  return v.v[0] == 0 || (uint32_t)v.v[0] == UL<F>::P1.v[0];
}

// acc + (neg ? -q : q).  DBL(qx, qy) -> the saturated XYZZ words of 2 q, called only when acc = q.
template <class C, class DBL> AVRF_DI g1_acc_u<C> g1u_madd(const g1_acc_u<C> &p, const uint32_t (&qx)[C::Fq::N], const uint32_t (&qy)[C::Fq::N], bool neg, DBL dbl) {
  using Fq = typename C::Fq; using K = G1U<C>; constexpr int L = UL<Fq>::L;
  if (words_zero(qx) && words_zero(qy)) return p;
  if (p.inf) return g1u_from_affine<C>(qx, qy, neg);
  const fu<L> sy = fu_cneg<L>(fu_slice<Fq, K::sy>(qy), neg ? -1 : 0);
  const fu<L> P = fu_sub<L>(fu_mul<Fq>(p.zz, fu_slice<Fq, K::sx>(qx)), p.x);
  const fu<L> R = fu_sub<L>(fu_mul<Fq>(p.zzz, sy), p.y);
  const fu<L> PP = fu_sqr<Fq>(P);
  if (__builtin_expect(__any(fu_maybe_zero_mod_p<Fq>(PP)), 0)) {
    if (fu_is_zero_mod_p<Fq>(PP)) {                                       // P = +-Q
      if (fu_is_zero_mod_p<Fq>(fu_sqr<Fq>(R))) return dbl();
      return g1u_identity<C>();
    }
  }
  g1_acc_u<C> r;
  const fu<L> PPP = fu_mul<Fq>(P, PP), Q = fu_mul<Fq>(p.x, PP);
  r.zz = fu_mul<Fq>(p.zz, PP); r.zzz = fu_mul<Fq>(p.zzz, PPP);
  const fu<L> T = fu_mul<Fq>(p.y, PPP), RR = fu_sqr<Fq>(R);
  fu<L> X3;
#pragma unroll
  for (int i = 0; i < L; i++) X3.v[i] = RR.v[i] - PPP.v[i] - 2 * Q.v[i];
  const fu<L> QX = fu_carry<Fq>(fu_sub<L>(Q, X3));
  r.y = fu_sub<L>(fu_mul<Fq>(R, QX), T);
  r.x = fu_carry<Fq>(X3);
  r.inf = 0;
  return r;
}

// ---- the general addition of the fixed-base reduction kernels (k_bucket_sum, k_heavy_sum, k_wsum, k_wsum_blk over G1RedCurve,
// curves.h): XYZZ in the Montgomery domain R' = 2^(W L) itself -- every coordinate is value * R', lazily reduced, so fu_mul is
// a plain Montgomery product and the formulas are add-2008-s / dbl-2008-s-1 as written, with no scaling to track.  A saturated
// coordinate v R enters as the limbs of (v R) * 2^SH = v R' (fu_slice with the shift: an integer below 2^SH p) and leaves through
// fu_mul(x R', R) = x R and fu_to_packed.  The additions are 12 + 2 asm blocks of 461 / 383 instructions (14 x 28) against
// ~650 of the saturated 12-word multiplier.  tools/fpu_model.py "g1r_add / g1r_dbl" proves: limbs within the multipliers'
// operand ranges, |values| < 128 p everywhere (the doubling of a freshly loaded point is the widest), the exceptional-case test
// below exact for every pair of loaded / computed operands.
template <class C> struct g1_red {
  fuF<typename C::Fq> x, y, zz, zzz;
  uint32_t inf;                                 // != 0: the identity (coordinates meaningless)
};
// limbs of a SQUARE's Montgomery output == 0 mod p: all zero, p's or 2 p's (exact while operand^2 / R' < 2 p: two freshly loaded
// coordinates differ by up to ~1.06 sqrt(R' p))
template <class F> AVRF_DI bool fu_is_zero_mod_p2(const fuF<F> &v) {
  using U = UL<F>;
  uint32_t z = 0, zp = 0, z2 = 0;
#pragma unroll
  for (int i = 0; i < U::L; i++) { z |= (uint32_t)v.v[i]; zp |= (uint32_t)v.v[i] ^ U::P1.v[i]; z2 |= (uint32_t)v.v[i] ^ U::template PK<1>.v[i]; }
  return z == 0 || zp == 0 || z2 == 0;
}
template <class F> AVRF_DI bool fu_maybe_zero_mod_p2(const fuF<F> &v) {
  const uint32_t l = (uint32_t)v.v[0];
  return l == 0 || l == UL<F>::P1.v[0] || l == UL<F>::template PK<1>.v[0];
}
template <class C> AVRF_DI g1_red<C> g1r_identity() {
  constexpr int L = UL<typename C::Fq>::L;
  g1_red<C> r; r.x = fu_zero<L>(); r.y = r.x; r.zz = r.x; r.zzz = r.x; r.inf = 1u; return r;
}
// canonical saturated XYZZ words (zz = 0: the identity)
template <class C> AVRF_DI g1_red<C> g1r_from_sat(const uint32_t (&X)[C::Fq::N], const uint32_t (&Y)[C::Fq::N], const uint32_t (&ZZ)[C::Fq::N], const uint32_t (&ZZZ)[C::Fq::N]) {
  using Fq = typename C::Fq; constexpr int SH = UL<Fq>::SH;
  g1_red<C> r;
  r.x = fu_slice<Fq, SH>(X); r.y = fu_slice<Fq, SH>(Y); r.zz = fu_slice<Fq, SH>(ZZ); r.zzz = fu_slice<Fq, SH>(ZZZ);
  r.inf = words_zero(ZZ) ? 1u : 0u;
  return r;
}
template <class C> AVRF_DI void g1r_to_sat(const g1_red<C> &a, uint32_t (&X)[C::Fq::N], uint32_t (&Y)[C::Fq::N], uint32_t (&ZZ)[C::Fq::N], uint32_t (&ZZZ)[C::Fq::N]) {
  using Fq = typename C::Fq; constexpr int N = Fq::N;
  const fuF<Fq> one = fu_const<Fq>(UL<Fq>::ONE);                    // R mod p: fu_mul(x R', R) = x R
  fu_to_packed<Fq, 2>(X, fu_mul<Fq>(a.x, one)); fu_to_packed<Fq, 2>(Y, fu_mul<Fq>(a.y, one));
  fu_to_packed<Fq, 2>(ZZ, fu_mul<Fq>(a.zz, one)); fu_to_packed<Fq, 2>(ZZZ, fu_mul<Fq>(a.zzz, one));
  if (a.inf) {
#pragma unroll
    for (int i = 0; i < N; i++) { X[i] = Fq::ONE[i]; Y[i] = Fq::ONE[i]; ZZ[i] = 0; ZZZ[i] = 0; }
  }
}
// 2 a (dbl-2008-s-1, a = 0): 6M + 3S
template <class C> AVRF_DI g1_red<C> g1r_dbl(const g1_red<C> &a) {
  using Fq = typename C::Fq; constexpr int L = UL<Fq>::L;
  if (a.inf) return a;
  fu<L> V = fu_sqr<Fq>(a.y), U, M = fu_sqr<Fq>(a.x);
#pragma unroll
  for (int i = 0; i < L; i++) { V.v[i] *= 4; U.v[i] = 2 * a.y.v[i]; M.v[i] *= 3; }
  V = fu_carry<Fq>(V); M = fu_carry<Fq>(M);
  const fu<L> W = fu_mul<Fq>(U, V), S = fu_mul<Fq>(a.x, V), MM = fu_sqr<Fq>(M);
  fu<L> X3;
#pragma unroll
  for (int i = 0; i < L; i++) X3.v[i] = MM.v[i] - 2 * S.v[i];
  const fu<L> SX = fu_carry<Fq>(fu_sub<L>(S, X3));
  g1_red<C> r;
  r.y = fu_sub<L>(fu_mul<Fq>(M, SX), fu_mul<Fq>(W, a.y));
  r.x = fu_carry<Fq>(X3);
  r.zz = fu_mul<Fq>(V, a.zz); r.zzz = fu_mul<Fq>(W, a.zzz);
  r.inf = 0;
  return r;
}
// a + b (add-2008-s): 12M + 2S; the exceptional cases leave early
template <class C> AVRF_DI g1_red<C> g1r_add(const g1_red<C> &a, const g1_red<C> &b) {
  using Fq = typename C::Fq; constexpr int L = UL<Fq>::L;
  if (a.inf) return b;
  if (b.inf) return a;
  const fu<L> U1 = fu_mul<Fq>(a.x, b.zz), S1 = fu_mul<Fq>(a.y, b.zzz);
  const fu<L> P = fu_sub<L>(fu_mul<Fq>(b.x, a.zz), U1), R = fu_sub<L>(fu_mul<Fq>(b.y, a.zzz), S1);
  const fu<L> PP = fu_sqr<Fq>(P);
  if (__builtin_expect(__any(fu_maybe_zero_mod_p2<Fq>(PP)), 0)) {
    if (fu_is_zero_mod_p2<Fq>(PP)) {                                      // b = +-a
      if (fu_is_zero_mod_p2<Fq>(fu_sqr<Fq>(R))) return g1r_dbl<C>(a);
      return g1r_identity<C>();
    }
  }
  g1_red<C> r;
  const fu<L> PPP = fu_mul<Fq>(P, PP), Q = fu_mul<Fq>(U1, PP);
  r.zz = fu_mul<Fq>(fu_mul<Fq>(a.zz, b.zz), PP); r.zzz = fu_mul<Fq>(fu_mul<Fq>(a.zzz, b.zzz), PPP);
  const fu<L> T = fu_mul<Fq>(S1, PPP), RR = fu_sqr<Fq>(R);
  fu<L> X3;
#pragma unroll
  for (int i = 0; i < L; i++) X3.v[i] = RR.v[i] - PPP.v[i] - 2 * Q.v[i];
  const fu<L> QX = fu_carry<Fq>(fu_sub<L>(Q, X3));
  r.y = fu_sub<L>(fu_mul<Fq>(R, QX), T);
  r.x = fu_carry<Fq>(X3);
  r.inf = 0;
  return r;
}

// A partial sum as k_accumulate leaves it (raw limbs + the identity flag; see fpu_te.h teu_store_part for why the conversion
// belongs to the reader): 4 L limbs, the flag, padding to a multiple of four words.
template <class C> struct G1UPart { static constexpr int L = UL<typename C::Fq>::L, WORDS = (4 * L + 1 + 3) / 4 * 4; };
template <class C> AVRF_DI void g1u_store_part(uint32_t *p, const g1_acc_u<C> &a) {
  constexpr int L = G1UPart<C>::L, WORDS = G1UPart<C>::WORDS;
  uint32_t w[WORDS];
#pragma unroll
  for (int i = 0; i < L; i++) { w[i] = (uint32_t)a.x.v[i]; w[L + i] = (uint32_t)a.y.v[i]; w[2 * L + i] = (uint32_t)a.zz.v[i]; w[3 * L + i] = (uint32_t)a.zzz.v[i]; }
  w[4 * L] = a.inf;
#pragma unroll
  for (int i = 4 * L + 1; i < WORDS; i++) w[i] = 0;
  uint4 *d = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < WORDS / 4; i++) d[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// -> the canonical saturated XYZZ words (x, y, zz, zzz: N words each; the identity: zz = zzz = 0)
template <class C> AVRF_DI void g1u_load_part(const uint32_t *p, uint32_t (&X)[C::Fq::N], uint32_t (&Y)[C::Fq::N], uint32_t (&ZZ)[C::Fq::N], uint32_t (&ZZZ)[C::Fq::N]) {
  using Fq = typename C::Fq; using K = G1U<C>;
  constexpr int L = G1UPart<C>::L, WORDS = G1UPart<C>::WORDS, N = Fq::N;
  uint32_t w[WORDS];
  const uint4 *s = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < WORDS / 4; i++) { const uint4 q = s[i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
  fu<L> x, y, zz, zzz;
#pragma unroll
  for (int i = 0; i < L; i++) { x.v[i] = (int32_t)w[i]; y.v[i] = (int32_t)w[L + i]; zz.v[i] = (int32_t)w[2 * L + i]; zzz.v[i] = (int32_t)w[3 * L + i]; }
  if constexpr (K::kx == 0) fu_to_packed<Fq, 4>(X, x);                    // |X| < 2^a p (an affine point stored as it entered)
  else fu_to_packed<Fq, 2>(X, fu_mul<Fq>(x, fu_const<Fq>(K::CX)));
  fu_to_packed<Fq, 2>(Y, fu_mul<Fq>(y, fu_const<Fq>(K::CY)));
  fu_to_packed<Fq, 2>(ZZ, zz); fu_to_packed<Fq, 2>(ZZZ, zzz);
  if (w[4 * L]) {
#pragma unroll
    for (int i = 0; i < N; i++) { ZZ[i] = 0; ZZZ[i] = 0; }
  }
}

// A reduction point in memory (buckets, LDS): the raw limbs and the flag in the partial sums' layout; all-zero words (a bucket
// nothing was written to) read as the identity
template <class C> AVRF_DI void g1r_store(uint32_t *p, const g1_red<C> &a) {
  g1_acc_u<C> t; t.x = a.x; t.y = a.y; t.zz = a.zz; t.zzz = a.zzz; t.inf = a.inf;
  g1u_store_part<C>(p, t);
}
template <class C> AVRF_DI g1_red<C> g1r_load(const uint32_t *p) {
  constexpr int L = G1UPart<C>::L, WORDS = G1UPart<C>::WORDS;
  uint32_t w[WORDS];
  const uint4 *s = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < WORDS / 4; i++) { const uint4 q = s[i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
  g1_red<C> r;
  uint32_t zzor = 0;
#pragma unroll
  for (int i = 0; i < L; i++) { r.x.v[i] = (int32_t)w[i]; r.y.v[i] = (int32_t)w[L + i]; r.zz.v[i] = (int32_t)w[2 * L + i]; r.zzz.v[i] = (int32_t)w[3 * L + i]; zzor |= w[2 * L + i]; }
  r.inf = (w[4 * L] != 0 || zzor == 0) ? 1u : 0u;
  return r;
}
template <class C> AVRF_DI g1_red<C> g1r_shfl_down(const g1_red<C> &a, int delta) {
  constexpr int L = G1UPart<C>::L;
  g1_red<C> r;
#pragma unroll
  for (int i = 0; i < L; i++) {
    r.x.v[i] = __shfl_down(a.x.v[i], delta); r.y.v[i] = __shfl_down(a.y.v[i], delta);
    r.zz.v[i] = __shfl_down(a.zz.v[i], delta); r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta);
  }
  r.inf = __shfl_down(a.inf, delta);
  return r;
}

}  // namespace avrf
