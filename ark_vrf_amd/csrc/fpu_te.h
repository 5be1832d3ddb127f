// fpu_te.h -- the twisted-Edwards mixed addition of k_accumulate on unsaturated limbs (fpu.h): the accumulator of a lane lives
// in 4 x 9 signed 29-bit limbs between the bucket boundaries of its share; bases are read from the same te_pre table as
// before (x | y | k = d x y, saturated Montgomery words) and sliced into limbs in registers; a partial sum leaves through
// fu_to_packed as the canonical te_ext every downstream kernel reads.  (madd-2008-hwcd, as te.h's te_madd.)
//
// Montgomery domains.  A product by a base coordinate is exact in the saturated form's domain (fpu.h: the base is sliced as
// b R 2^5).  The four closing products E F, G H, E H, F G multiply two accumulator-domain values and come out times 2^-5 --
// ALL FOUR, so the new accumulator is 2^-5 (X3, Y3, T3, Z3): the same projective point (extended coordinates are homogeneous:
// x = X / Z, y = Y / Z, T Z = X Y).  No correction is ever applied.
//
// Sign.  The accumulator holds s (X, Y, T, Z) -> the point (s X, Y, s T, Z), s = +-1 in `neg`: adding e Q (e = +-1) to it is
// e ((s e) acc + Q), so the BASE is never negated (its x + y is formed once, in saturated words, before slicing); the
// accumulator's X and T change sign when s e = -1 (two instructions per limb) and the new s is e.
//
// Bounds (tools/fpu_model.py): with |X| < 1.5p, |Y| < 1.7p, |T| < 2.4p, |Z| < 1.3p going in, the same holds coming out for
// every a in {1, -1, -5} and every field of 251..255 bits; limb magnitudes at every product are <= (2^30, 2^29 + 4).
#pragma once
#include "fpu.h"
#include "te.h"

namespace avrf {

template <class S> struct te_acc_u {
  fuF<typename S::Fq> x, y, t, z;
  uint32_t neg;                               // 0 / 0xffffffff: the point is (-x, y, -t, z)
};

template <class S> AVRF_DI te_acc_u<S> teu_identity() {
  using Fq = typename S::Fq;
  te_acc_u<S> r;
  r.x = fu_zero<UL<Fq>::L>(); r.t = r.x; r.y = fu_const<Fq>(UL<Fq>::ONE); r.z = r.y; r.neg = 0;
  return r;
}
// the base itself (first entry of a lane's share): t = x y
template <class S> AVRF_DI te_acc_u<S> teu_from_pre(const te_pre &q, bool neg) {
  using Fq = typename S::Fq;
  te_acc_u<S> r;
  r.x = fu_slice<Fq, 0>(q.x.v); r.y = fu_slice<Fq, 0>(q.y.v);
  r.t = fu_mul<Fq>(r.x, fu_slice<Fq, UL<Fq>::SH>(q.y.v));
  r.z = fu_const<Fq>(UL<Fq>::ONE);
  r.neg = neg ? 0xffffffffu : 0u;
  return r;
}
// acc + (neg ? -q : q)
template <class S> AVRF_DI te_acc_u<S> teu_madd(const te_acc_u<S> &p, const te_pre &q, bool neg) {
  using Fq = typename S::Fq;
  constexpr int SH = UL<Fq>::SH, L = UL<Fq>::L;
  const uint32_t e = neg ? 0xffffffffu : 0u;
  const int32_t flip = (int32_t)(p.neg ^ e);
  const fu<L> X1 = fu_cneg<L>(p.x, flip), T1 = fu_cneg<L>(p.t, flip);
  fp xy; add8(xy, q.x, q.y);                                                 // x + y < 2p < 2^256 (top bit of p clear)
  const fu<L> A = fu_mul<Fq>(X1, fu_slice<Fq, SH>(q.x.v));
  const fu<L> B = fu_mul<Fq>(p.y, fu_slice<Fq, SH>(q.y.v));
  const fu<L> C = fu_mul<Fq>(T1, fu_slice<Fq, SH>(q.k.v));
  fu<L> E = fu_mul<Fq>(fu_add<L>(X1, p.y), fu_slice<Fq, SH>(xy.v));
  E = fu_sub<L>(fu_sub<L>(E, A), B);
  const fu<L> F = fu_sub<L>(p.z, C), G = fu_add<L>(p.z, C);
  fu<L> H;                                                                   // B - a A
  if (S::A_KIND == 1) {                                                      // a = -5: B + 5 A, limbs < 6 * 2^29 (unsigned: more than an int32 holds)
    uint32_t h[L - 1];
#pragma unroll
    for (int i = 0; i < L - 1; i++) h[i] = (uint32_t)B.v[i] + 5u * (uint32_t)A.v[i];
    H = fu_carry_u<Fq>(h, B.v[L - 1] + 5 * A.v[L - 1]);
  } else if (S::A_KIND == 2) H = fu_carry<Fq>(fu_add<L>(B, A));              // a = -1
  else H = fu_sub<L>(B, A);                                                  // a = 1
  te_acc_u<S> r;
  r.x = fu_mul<Fq>(E, F); r.y = fu_mul<Fq>(G, H); r.t = fu_mul<Fq>(E, H); r.z = fu_mul<Fq>(F, G);
  r.neg = e;
  return r;
}
// the canonical te_ext (what te.h's kernels read)
template <class S> AVRF_DI te_ext teu_to_ext(const te_acc_u<S> &p) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L;
  te_ext r;
  fu_to_packed<Fq>(r.x.v, fu_cneg<L>(p.x, (int32_t)p.neg)); fu_to_packed<Fq>(r.y.v, p.y);
  fu_to_packed<Fq>(r.t.v, fu_cneg<L>(p.t, (int32_t)p.neg)); fu_to_packed<Fq>(r.z.v, p.z);
  return r;
}

// A partial sum as a lane of k_accumulate leaves it at a bucket boundary: the 4 x 9 limbs and the sign, forty words, ten 16-byte
// stores under the lane mask -- NOT the canonical te_ext.  The boundary falls at a different iteration in every lane, so
// whatever runs there runs for the whole wave once per iteration with ANY boundary in it (64 % of the iterations at C2's
// 40-entry shares of 64-entry buckets); the four conditional-subtraction chains of fu_to_packed (~450 instructions) belong to
// the readers (k_bucket_sum, k_heavy_sum: one lane or one quad per partial, uniform control flow).
constexpr int TEU_PART_WORDS = 40;
template <class S> AVRF_DI void teu_store_part(uint32_t *p, const te_acc_u<S> &a) {
  uint4 *d = reinterpret_cast<uint4 *>(p);
  const int32_t *x = a.x.v, *y = a.y.v, *t = a.t.v, *z = a.z.v;
  d[0] = make_uint4(x[0], x[1], x[2], x[3]); d[1] = make_uint4(x[4], x[5], x[6], x[7]);
  d[2] = make_uint4(x[8], y[0], y[1], y[2]); d[3] = make_uint4(y[3], y[4], y[5], y[6]);
  d[4] = make_uint4(y[7], y[8], t[0], t[1]); d[5] = make_uint4(t[2], t[3], t[4], t[5]);
  d[6] = make_uint4(t[6], t[7], t[8], z[0]); d[7] = make_uint4(z[1], z[2], z[3], z[4]);
  d[8] = make_uint4(z[5], z[6], z[7], z[8]); d[9] = make_uint4(a.neg, 0u, 0u, 0u);
}
// coordinate j (0 x, 1 y, 2 t, 3 z) of a stored partial as canonical saturated words
template <class S> AVRF_DI fp teu_load_part_coord(const uint32_t *p, int j) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L;
  fu<L> v;
#pragma unroll
  for (int i = 0; i < L; i++) v.v[i] = (int32_t)p[j * L + i];
  const int32_t s = (j & 1) ? 0 : (int32_t)p[4 * L];
  fp r; fu_to_packed<Fq>(r.v, fu_cneg<L>(v, s));
  return r;
}
template <class S> AVRF_DI te_ext teu_load_part(const uint32_t *p) {
  const uint4 *d = reinterpret_cast<const uint4 *>(p);
  uint32_t w[40];
#pragma unroll
  for (int i = 0; i < 10; i++) { const uint4 q = d[i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
  te_acc_u<S> a;
#pragma unroll
  for (int i = 0; i < 9; i++) { a.x.v[i] = (int32_t)w[i]; a.y.v[i] = (int32_t)w[9 + i]; a.t.v[i] = (int32_t)w[18 + i]; a.z.v[i] = (int32_t)w[27 + i]; }
  a.neg = w[36];
  return teu_to_ext<S>(a);
}

}  // namespace avrf
