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

// ---- the doubling chains of the per-item kernels (proto_dev.h te_smul_*, glv.h): the running point of a scalar multiplication in
// unsaturated limbs, the window-table entries read from memory in their saturated form and sliced as they arrive
// (dbl-2008-hwcd / add-2008-hwcd as te.h).  Montgomery domains as above: products by a table coordinate are exact (the entry is
// sliced with the shift), products of two running values come out times 2^-SH -- uniformly in all four coordinates of a result.
// Bounds (tools/fpu_model.py, "per-item doubling chains"): |X|, |Y|, |T|, |Z| < 1.6p, 1.8p, 2.5p, 1.9p are closed under both
// operations; operands of every product are within (2^30, 2^29 + 4) after the carry passes below.  The products are the asm blocks
// of fpu_asm_gen.h (206 / 178 vector instructions against the saturated multiplier's ~250, which also squares by multiplying).
template <class S> struct teu4 { fuF<typename S::Fq> x, y, t, z; };

template <class S> AVRF_DI teu4<S> teu4_identity() {
  using Fq = typename S::Fq;
  teu4<S> r; r.x = fu_zero<UL<Fq>::L>(); r.t = r.x; r.y = fu_const<Fq>(UL<Fq>::ONE); r.z = r.y; return r;
}
template <class S> AVRF_DI teu4<S> teu4_from_ext(const te_ext &e) {
  using Fq = typename S::Fq;
  teu4<S> r; r.x = fu_slice<Fq, 0>(e.x.v); r.y = fu_slice<Fq, 0>(e.y.v); r.t = fu_slice<Fq, 0>(e.t.v); r.z = fu_slice<Fq, 0>(e.z.v); return r;
}
template <class S> AVRF_DI te_ext teu4_to_ext(const teu4<S> &p) {
  using Fq = typename S::Fq;
  te_ext r; fu_to_packed<Fq>(r.x.v, p.x); fu_to_packed<Fq>(r.y.v, p.y); fu_to_packed<Fq>(r.t.v, p.t); fu_to_packed<Fq>(r.z.v, p.z); return r;
}
// B - a A for the closing products: carried to limbs in [0, 2^W + 4) where the sum can exceed 2^30
template <class S, int L> AVRF_DI fu<L> teu_h(const fu<L> &A, const fu<L> &B) {
  using Fq = typename S::Fq;
  if (S::A_KIND == 1) {
    uint32_t h[L - 1];
#pragma unroll
    for (int i = 0; i < L - 1; i++) h[i] = (uint32_t)B.v[i] + 5u * (uint32_t)A.v[i];
    return fu_carry_u<Fq>(h, B.v[L - 1] + 5 * A.v[L - 1]);
  }
  if (S::A_KIND == 2) return fu_carry<Fq>(fu_add<L>(B, A));
  return fu_sub<L>(B, A);
}
// 2 P (4M + 4S)
template <class S> AVRF_DI teu4<S> teu4_dbl(const teu4<S> &p) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L;
  const fu<L> A = fu_sqr<Fq>(p.x), B = fu_sqr<Fq>(p.y), Zs = fu_sqr<Fq>(p.z);
  const fu<L> Sq = fu_sqr<Fq>(fu_carry<Fq>(fu_add<L>(p.x, p.y)));
  const fu<L> E = fu_sub<L>(fu_sub<L>(Sq, A), B);                  // limbs in (-2^30, 2^29)
  // G = D + B, H = D - B, F = G - C with D = a A, C = 2 Zs.  a = -5: 5 A is formed and carried in unsigned arithmetic first (its
  // limbs need 32 bits); then |G_i| <= 2^29 + 4, |H_i| <= 2^30 + 4, |F_i| <= 3 * 2^29 + 4 all fit an int32 and H, F take a carry pass.
  fu<L> G, H, F;
  if (S::A_KIND == 1) {
    const fu<L> A5 = fu_times5<Fq>(A);
#pragma unroll
    for (int i = 0; i < L; i++) { G.v[i] = B.v[i] - A5.v[i]; H.v[i] = -A5.v[i] - B.v[i]; F.v[i] = G.v[i] - 2 * Zs.v[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int32_t D = S::A_KIND == 2 ? -A.v[i] : A.v[i];
      G.v[i] = D + B.v[i]; H.v[i] = D - B.v[i]; F.v[i] = G.v[i] - 2 * Zs.v[i];
    }
  }
  H = fu_carry<Fq>(H); F = fu_carry<Fq>(F);
  teu4<S> r;
  r.x = fu_mul<Fq>(E, F); r.y = fu_mul<Fq>(G, H); r.t = fu_mul<Fq>(E, H); r.z = fu_mul<Fq>(F, G);
  return r;
}
// P + e, e an extended point in saturated canonical words (a window-table entry): 9M + 1 by d
template <class S> AVRF_DI teu4<S> teu4_add_sat(const teu4<S> &p, const te_ext &e) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L, SH = UL<Fq>::SH;
  fp xy; add8(xy, e.x, e.y);
  const fu<L> A = fu_mul<Fq>(p.x, fu_slice<Fq, SH>(e.x.v)), B = fu_mul<Fq>(p.y, fu_slice<Fq, SH>(e.y.v));
  const fu<L> C = fu_mul<Fq>(fu_mul<Fq>(p.t, fu_slice<Fq, SH>(e.t.v)), fu_slice<Fq, SH>(S::D));
  const fu<L> D = fu_mul<Fq>(p.z, fu_slice<Fq, SH>(e.z.v));
  fu<L> E = fu_mul<Fq>(fu_add<L>(p.x, p.y), fu_slice<Fq, SH>(xy.v));
  E = fu_sub<L>(fu_sub<L>(E, A), B);
  const fu<L> F = fu_sub<L>(D, C), G = fu_add<L>(D, C), H = teu_h<S, L>(A, B);
  teu4<S> r;
  r.x = fu_mul<Fq>(E, F); r.y = fu_mul<Fq>(G, H); r.t = fu_mul<Fq>(E, H); r.z = fu_mul<Fq>(F, G);
  return r;
}
// P + q, q affine with k = d x y (a fixed-base table entry): 8M
template <class S> AVRF_DI teu4<S> teu4_madd_pre(const teu4<S> &p, const te_pre &q) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L, SH = UL<Fq>::SH;
  fp xy; add8(xy, q.x, q.y);
  const fu<L> A = fu_mul<Fq>(p.x, fu_slice<Fq, SH>(q.x.v)), B = fu_mul<Fq>(p.y, fu_slice<Fq, SH>(q.y.v));
  const fu<L> C = fu_mul<Fq>(p.t, fu_slice<Fq, SH>(q.k.v));
  fu<L> E = fu_mul<Fq>(fu_add<L>(p.x, p.y), fu_slice<Fq, SH>(xy.v));
  E = fu_sub<L>(fu_sub<L>(E, A), B);
  const fu<L> F = fu_sub<L>(p.z, C), G = fu_add<L>(p.z, C), H = teu_h<S, L>(A, B);
  teu4<S> r;
  r.x = fu_mul<Fq>(E, F); r.y = fu_mul<Fq>(G, H); r.t = fu_mul<Fq>(E, H); r.z = fu_mul<Fq>(F, G);
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
// the same coordinate as its raw limbs with the point's sign applied (the unsaturated quad addition of te_quad.h reads it as it is)
template <class S> AVRF_DI fuF<typename S::Fq> teu_load_part_coord_raw(const uint32_t *p, int j) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L;
  fu<L> v;
#pragma unroll
  for (int i = 0; i < L; i++) v.v[i] = (int32_t)p[j * L + i];
  const int32_t s = (j & 1) ? 0 : (int32_t)p[4 * L];
  return fu_cneg<L>(v, s);
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
