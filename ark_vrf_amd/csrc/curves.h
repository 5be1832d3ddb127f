// curves.h -- curve policies for the generic Pippenger kernels (msm_kernels.h):
//   TeCurve<Suite>   twisted Edwards, extended coordinates (te.h)           -> Thin / Pedersen batch MSMs
//   G1Curve<G1>      short Weierstrass y^2 = x^3 + b (a = 0), XYZZ coordinates -> KZG commit / open MSMs
//                    of the ring SNARK (ark-bls12-381 / ark-bn254 G1; w3f-pcs KZG; src/ring.rs:220,404,416,731)
// A policy provides: base_t (precomputed affine base as stored in HBM), acc_t (accumulator),
// identity / madd(acc, base, negate) / add(acc, acc), raw load/store, and a lane shuffle.
#pragma once
#include <type_traits>
#include "fpn.h"
#include "te.h"
#include "fpu_te.h"
#include "fpu_g1.h"

#ifndef AVRF_TE_ACC_WAVES
#define AVRF_TE_ACC_WAVES 2
#endif
// k_accumulate<TeCurve> needs 141 VGPRs: three waves per SIMD fit.  Alone the kernel is 2 % faster with three (0.429 ms against
// 0.440 ms per 262 145-term batch), with 20 contexts in flight the batch rate is 2-4 % higher with two (103.6-107.5 M/s against
// 101.4-103.5 M/s, same box, tools/r3_run11.sh): the third wave takes the registers the other contexts' kernels overlap in.
#ifndef AVRF_TE_ACC_MAX_WAVES
#define AVRF_TE_ACC_MAX_WAVES 2
#endif
// 12-limb G1 accumulate: 2 waves per SIMD = 192 VGPRs (201 with the carry-chain field additions), no scratch; 3 = 168 VGPRs + 44
// spilled registers.  A/B on one box
// (tools/r3_run4.sh, ring 1024): 4 contexts 11.49 k proofs/s against 11.05-11.11 k, one context 8.0-9.3 k against 7.4-7.8 k,
// and the window-table build of a setup 82 ms against 96-118 ms.
#ifndef AVRF_G1_ACC_WAVES
#define AVRF_G1_ACC_WAVES 2
#endif
// reduction kernels of the 12-limb curve (general additions, two accumulators live): the whole register file, no scratch
// (at 2 waves per SIMD k_wsum spilled 94 registers and k_wsum_blk 458; same A/B: 11.48-11.63 k proofs/s against 11.19-11.51 k.
// With the carry-chain field additions the spills are 40 / 71 registers and the A/B is a tie: 12.4-12.9 k against 12.7 k,
// tools/r3_run13.sh)
#ifndef AVRF_G1_RED_WAVES
#define AVRF_G1_RED_WAVES 1
#endif

#ifndef AVRF_G1_RED_INLINE
#define AVRF_G1_RED_INLINE 1     // the reductions' general additions inlined at each call site (0: one out-of-line copy per kernel)
#endif

namespace avrf {

// what k_accumulate keeps in registers between the bucket boundaries of a lane's share: the policy's own accumulator, or -- the
// twisted-Edwards suites with an asm-free path -- the unsaturated-limb form of fpu_te.h (teu_madd 18.3 G/s against te_madd
// 12.7 G/s at two waves per SIMD, profiles/r5_ubench_fpu_gate.txt).  Bases are read and partial sums are written in the
// policy's HBM formats either way.
template <class CV> struct AccumSame {
  using acc_t = typename CV::acc_t; using base_t = typename CV::base_t;
  static AVRF_DI acc_t identity() { return CV::identity(); }
  static AVRF_DI acc_t madd(const acc_t &a, const base_t &q, bool neg) { return CV::madd(a, q, neg); }
  static AVRF_DI acc_t from_base(const base_t &q, bool neg) { return CV::from_base(q, neg); }
  // a lane's partial sums in HBM (the array k_bucket_sum / k_heavy_sum read): words per slot, store, load as the policy's accumulator
  static constexpr int PART_WORDS = CV::ACC_WORDS;
  static AVRF_DI void store_part(uint32_t *p, const acc_t &a) { CV::store_acc(p, a); }
  static AVRF_DI typename CV::acc_t load_part(const uint32_t *p) { return CV::load_acc(p); }
};
template <class S> struct AccumTeU {
  using acc_t = te_acc_u<S>; using base_t = te_pre;
  static AVRF_DI acc_t identity() { return teu_identity<S>(); }
  static AVRF_DI acc_t madd(const acc_t &a, const base_t &q, bool neg) { return teu_madd<S>(a, q, neg); }
  static AVRF_DI acc_t from_base(const base_t &q, bool neg) { return teu_from_pre<S>(q, neg); }
  static constexpr int PART_WORDS = TEU_PART_WORDS;           // raw limbs + sign: fpu_te.h teu_store_part
  static AVRF_DI void store_part(uint32_t *p, const acc_t &a) { teu_store_part<S>(p, a); }
  static AVRF_DI te_ext load_part(const uint32_t *p) { return teu_load_part<S>(p); }
};

template <class C> struct G1Curve;
// XYZZ on unsaturated limbs (fpu_g1.h); the doubling of the P = Q case runs in the saturated form and is converted
template <class C> struct AccumG1U {
  using CV = G1Curve<C>; using Fq = typename C::Fq;
  using acc_t = g1_acc_u<C>; using base_t = typename CV::base_t;
  static AVRF_DI acc_t identity() { return g1u_identity<C>(); }
  static AVRF_DI acc_t madd(const acc_t &a, const base_t &q, bool neg) {
    return g1u_madd<C>(a, q.x.v, q.y.v, neg, [&]() {
      base_t t = q; if (neg) t.y = fn_neg<Fq>(t.y);
      const typename CV::acc_t r = CV::dbl_affine(t);
      return g1u_from_xyzz<C>(r.x.v, r.y.v, r.zz.v, r.zzz.v);
    });
  }
  static AVRF_DI acc_t from_base(const base_t &q, bool neg) { return g1u_from_affine<C>(q.x.v, q.y.v, neg); }
  static constexpr int PART_WORDS = G1UPart<C>::WORDS;
  static AVRF_DI void store_part(uint32_t *p, const acc_t &a) { g1u_store_part<C>(p, a); }
  static AVRF_DI typename CV::acc_t load_part(const uint32_t *p) {
    typename CV::acc_t r; g1u_load_part<C>(p, r.x.v, r.y.v, r.zz.v, r.zzz.v); return r;
  }
};

template <class S> struct TeCurve {
  using base_t = te_pre; using acc_t = te_ext; using suite = S;
#ifndef AVRF_NO_UNSAT
  using accum = typename std::conditional<!S::SW_NATIVE, AccumTeU<S>, AccumSame<TeCurve<S>>>::type;
#else
  using accum = AccumSame<TeCurve<S>>;
#endif
  static constexpr int BASE_WORDS = 24, ACC_WORDS = 32;
  // (S::SW_NATIVE, secp256r1: the policy's types hold XYZZ coordinates, te.h; the twisted-Edwards-only reduction kernels
  // -- te_quad.h, window triples -- are switched off and the generic row / column + bit-sum reduction of the G1 MSMs runs)
  static constexpr bool QUAD = !S::SW_NATIVE;         // reduction tails use the four-lanes-per-point addition (te_quad.h)
  static constexpr bool PREFETCH = true;
  static constexpr bool ZERO_IS_IDENTITY = S::SW_NATIVE;   // twisted Edwards: (0, 1, 0, 1); XYZZ: ZZ = 0
  static constexpr bool FIXED_TABLE = false;          // no fixed-base window-table mode (bases change per batch)
  // waves per SIMD asked of the register allocator in k_accumulate (the XYZZ mixed addition over the GENERIC multiplier -- secp256r1, whose
  // modulus fills its top bit -- needs more than 256 registers: one wave, the whole file, instead of 60 spilled registers)
  static constexpr int MIN_WAVES = (S::SW_NATIVE && S::Fq::FULL) ? 1 : AVRF_TE_ACC_WAVES;
  static constexpr int MAX_WAVES = AVRF_TE_ACC_MAX_WAVES;   // resident k_accumulate waves per SIMD at most (0 = what the registers allow), msm.hip accumulate_shape
  static constexpr int RED_WAVES = 1;                 // reduction kernels (general additions, several points live): latency-bound, full register file
  static constexpr bool INLINE_REDUCE_OPS = true;
  static constexpr bool WINDOW_SUMS = !S::SW_NATIVE;  // single MSMs: one weighted bucket sum per window (k_wsum_q1/q2) instead of row/column + bit sums
  static AVRF_DI acc_t identity() { return te_identity<S>(); }
  static AVRF_DI acc_t madd(const acc_t &a, base_t q, bool neg) {
    if (neg) q = te_pre_neg<S>(q);
    return te_madd<S>(a, q);
  }
  static AVRF_DI acc_t from_base(base_t q, bool neg) {
    if (neg) q = te_pre_neg<S>(q);
    return te_from_pre<S>(q);
  }
  static AVRF_DI acc_t add(const acc_t &a, const acc_t &b) { return te_add<S>(a, b); }
  static AVRF_DI acc_t dbl(const acc_t &a) { return te_dbl<S>(a); }
  static AVRF_DI base_t load_base(const uint32_t *p) { return load_pre(reinterpret_cast<const te_pre *>(p)); }
  static AVRF_DI base_t base_from_words(const uint32_t (&w)[BASE_WORDS]) {
    base_t r;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.x.v[i] = w[i]; r.y.v[i] = w[8 + i]; r.k.v[i] = w[16 + i]; }
    return r;
  }
  static AVRF_DI acc_t load_acc(const uint32_t *p) { return load_ext(reinterpret_cast<const te_ext *>(p)); }
  static AVRF_DI void store_acc(uint32_t *p, const acc_t &a) { store_ext(reinterpret_cast<te_ext *>(p), a); }
  using red = TeCurve<S>;                             // the policy of the fixed-base reduction kernels (G1 only has another one)
  static constexpr int OUT_WORDS = ACC_WORDS;
  static AVRF_DI void store_out(uint32_t *p, const acc_t &a) { store_acc(p, a); }
  static AVRF_DI acc_t shfl_down(const acc_t &p, int delta) {
    acc_t r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      r.x.v[i] = __shfl_down(p.x.v[i], delta); r.y.v[i] = __shfl_down(p.y.v[i], delta);
      r.t.v[i] = __shfl_down(p.t.v[i], delta); r.z.v[i] = __shfl_down(p.z.v[i], delta);
    }
    return r;
  }
  static AVRF_DI acc_t shfl_from(const acc_t &p, int lane) {
    acc_t r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      r.x.v[i] = __shfl(p.x.v[i], lane); r.y.v[i] = __shfl(p.y.v[i], lane);
      r.t.v[i] = __shfl(p.t.v[i], lane); r.z.v[i] = __shfl(p.z.v[i], lane);
    }
    return r;
  }
};

// XYZZ: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ = 0.  Affine bases: (0, 0) = infinity.
template <class C> struct G1RedCurve;
template <class C> struct G1Curve {
  using Fq = typename C::Fq;
  static constexpr int N = Fq::N;
  using el = fpn<N>;
  struct base_t { el x, y; };
  struct acc_t { el x, y, zz, zzz; };
#ifndef AVRF_NO_UNSAT_G1
  using accum = AccumG1U<C>;
#else
  using accum = AccumSame<G1Curve<C>>;
#endif
  static constexpr int BASE_WORDS = 2 * N, ACC_WORDS = 4 * N;
  static constexpr bool QUAD = false;
  static constexpr bool PREFETCH = (N <= 8);
#ifndef AVRF_NO_UNSAT_G1
  static constexpr int MIN_WAVES = AVRF_G1_ACC_WAVES;              // k_accumulate on unsaturated limbs: 256 registers (14 x 28), 166 (9 x 29; at three waves it spilled 57)
#else
  static constexpr int MIN_WAVES = N > 8 ? AVRF_G1_ACC_WAVES : 3;   // k_accumulate holds one accumulator + one base
#endif
  static constexpr int MAX_WAVES = 0;
  static constexpr int RED_WAVES = N > 8 ? AVRF_G1_RED_WAVES : 2;   // the general addition (two accumulators live) needs the 256-register budget
  static constexpr bool INLINE_REDUCE_OPS = true;     // k_wsum*: additions inlined (the asm multiplier keeps the code small)
  static constexpr bool WINDOW_SUMS = false;
  static constexpr bool ZERO_IS_IDENTITY = true;      // zz = 0; all-zero memory reads as the identity
  static constexpr bool FIXED_TABLE = true;           // KZG SRS: msm_g1_fixed_device

  static AVRF_DI acc_t identity() { acc_t r; r.x = fn_one<Fq>(); r.y = fn_one<Fq>(); r.zz = fn_zero<N>(); r.zzz = fn_zero<N>(); return r; }
  static AVRF_DI bool is_identity(const acc_t &a) { return fn_is_zero(a.zz); }
  static AVRF_DI acc_t from_affine(const base_t &q) {
    acc_t r; r.x = q.x; r.y = q.y; r.zz = fn_one<Fq>(); r.zzz = fn_one<Fq>();
    if (fn_is_zero(q.x) && fn_is_zero(q.y)) { r.zz = fn_zero<N>(); r.zzz = fn_zero<N>(); }
    return r;
  }
  static AVRF_DI acc_t from_base(base_t q, bool neg) { if (neg) q.y = fn_neg<Fq>(q.y); return from_affine(q); }
  // 2 * (affine q)  (mdbl-2008-s-1, a = 0)
  static AVRF_DI acc_t dbl_affine(const base_t &q) {
    el U = fn_dbl<Fq>(q.y), V = fn_sqr<Fq>(U), W = fn_mul<Fq>(U, V), S = fn_mul<Fq>(q.x, V);
    el X2 = fn_sqr<Fq>(q.x), M = fn_add<Fq>(fn_dbl<Fq>(X2), X2);
    acc_t r;
    r.x = fn_sub<Fq>(fn_sqr<Fq>(M), fn_dbl<Fq>(S));
    r.y = fn_sub<Fq>(fn_mul<Fq>(M, fn_sub<Fq>(S, r.x)), fn_mul<Fq>(W, q.y));
    r.zz = V; r.zzz = W;
    if (fn_is_zero(q.y)) return identity();
    return r;
  }
  // 2 * a  (dbl-2008-s-1, a = 0)
  static AVRF_DI acc_t dbl(const acc_t &a) {
    el U = fn_dbl<Fq>(a.y), V = fn_sqr<Fq>(U), W = fn_mul<Fq>(U, V), S = fn_mul<Fq>(a.x, V);
    el X2 = fn_sqr<Fq>(a.x), M = fn_add<Fq>(fn_dbl<Fq>(X2), X2);
    acc_t r;
    r.x = fn_sub<Fq>(fn_sqr<Fq>(M), fn_dbl<Fq>(S));
    r.y = fn_sub<Fq>(fn_mul<Fq>(M, fn_sub<Fq>(S, r.x)), fn_mul<Fq>(W, a.y));
    r.zz = fn_mul<Fq>(V, a.zz); r.zzz = fn_mul<Fq>(W, a.zzz);
    return r;                                     // a identity (zz = 0) or y = 0 -> zz = 0: identity
  }
  // a + (neg ? -q : q), q affine  (madd-2008-s: 8M + 2S).  The exceptional cases leave early (wave-divergent but rare:
  // the first addition into an empty accumulator, a point at infinity in the table, P = +-Q), so that neither the old
  // accumulator nor q stays live across the main sequence -- that is what lets the 381-bit version fit its registers.
  static AVRF_DI acc_t madd(const acc_t &a, base_t q, bool neg) {
    if (neg) q.y = fn_neg<Fq>(q.y);
    if (fn_is_zero(q.x) && fn_is_zero(q.y)) return a;
    if (is_identity(a)) return from_affine(q);
    el P = fn_sub<Fq>(fn_mul<Fq>(q.x, a.zz), a.x), R = fn_sub<Fq>(fn_mul<Fq>(q.y, a.zzz), a.y);
    if (fn_is_zero(P)) return fn_is_zero(R) ? dbl_affine(q) : identity();
    acc_t r;
    el PP = fn_sqr<Fq>(P);
    r.zz = fn_mul<Fq>(a.zz, PP);
    el Q = fn_mul<Fq>(a.x, PP);
    el PPP = fn_mul<Fq>(P, PP);
    r.zzz = fn_mul<Fq>(a.zzz, PPP);
    el T = fn_mul<Fq>(a.y, PPP);
    r.x = fn_sub<Fq>(fn_sub<Fq>(fn_sqr<Fq>(R), PPP), fn_dbl<Fq>(Q));
    r.y = fn_sub<Fq>(fn_mul<Fq>(R, fn_sub<Fq>(Q, r.x)), T);
    return r;
  }
  // a + b  (add-2008-s: 12M + 2S); exceptional cases leave early so that a and b die as the sequence consumes them
  static AVRF_DI acc_t add(const acc_t &a, const acc_t &b) {
    if (is_identity(a)) return b;
    if (is_identity(b)) return a;
    el U1 = fn_mul<Fq>(a.x, b.zz), P = fn_sub<Fq>(fn_mul<Fq>(b.x, a.zz), U1);
    el S1 = fn_mul<Fq>(a.y, b.zzz), R = fn_sub<Fq>(fn_mul<Fq>(b.y, a.zzz), S1);
    if (fn_is_zero(P)) return fn_is_zero(R) ? dbl(a) : identity();
    acc_t r;
    el PP = fn_sqr<Fq>(P);
    r.zz = fn_mul<Fq>(fn_mul<Fq>(a.zz, b.zz), PP);
    el Q = fn_mul<Fq>(U1, PP);
    el PPP = fn_mul<Fq>(P, PP);
    r.zzz = fn_mul<Fq>(fn_mul<Fq>(a.zzz, b.zzz), PPP);
    el T = fn_mul<Fq>(S1, PPP);
    r.x = fn_sub<Fq>(fn_sub<Fq>(fn_sqr<Fq>(R), PPP), fn_dbl<Fq>(Q));
    r.y = fn_sub<Fq>(fn_mul<Fq>(R, fn_sub<Fq>(Q, r.x)), T);
    return r;
  }
  static AVRF_DI base_t load_base(const uint32_t *p) { base_t r; r.x = fn_load<N>(p); r.y = fn_load<N>(p + N); return r; }
  static AVRF_DI base_t base_from_words(const uint32_t (&w)[BASE_WORDS]) {
    base_t r;
#pragma unroll
    for (int i = 0; i < N; i++) { r.x.v[i] = w[i]; r.y.v[i] = w[N + i]; }
    return r;
  }
  static AVRF_DI acc_t load_acc(const uint32_t *p) {
    acc_t r; r.x = fn_load<N>(p); r.y = fn_load<N>(p + N); r.zz = fn_load<N>(p + 2 * N); r.zzz = fn_load<N>(p + 3 * N); return r;
  }
  static AVRF_DI void store_acc(uint32_t *p, const acc_t &a) {
    fn_store<N>(p, a.x); fn_store<N>(p + N, a.y); fn_store<N>(p + 2 * N, a.zz); fn_store<N>(p + 3 * N, a.zzz);
  }
  // the reduction kernels of a fixed-base MSM (bucket sums, weighted sums): general additions on the asm multipliers
#if !defined(AVRF_NO_UNSAT_G1) && !defined(AVRF_NO_UNSAT_G1_RED)
  using red = G1RedCurve<C>;
#else
  using red = G1Curve<C>;
#endif
  static constexpr int OUT_WORDS = ACC_WORDS;
  static AVRF_DI void store_out(uint32_t *p, const acc_t &a) { store_acc(p, a); }
  static AVRF_DI acc_t shfl_down(const acc_t &a, int delta) {
    acc_t r; r.x = fn_shfl_down<N>(a.x, delta); r.y = fn_shfl_down<N>(a.y, delta);
    r.zz = fn_shfl_down<N>(a.zz, delta); r.zzz = fn_shfl_down<N>(a.zzz, delta); return r;
  }
  static AVRF_DI acc_t shfl_from(const acc_t &a, int lane) {
    acc_t r;
#pragma unroll
    for (int i = 0; i < N; i++) {
      r.x.v[i] = __shfl(a.x.v[i], lane); r.y.v[i] = __shfl(a.y.v[i], lane);
      r.zz.v[i] = __shfl(a.zz.v[i], lane); r.zzz.v[i] = __shfl(a.zzz.v[i], lane);
    }
    return r;
  }
};

// G1 in the reduction kernels of the fixed-base MSMs (k_bucket_sum, k_heavy_sum, k_wsum, k_wsum_blk): the same interface as
// G1Curve over g1_red (fpu_g1.h: XYZZ in the Montgomery domain of the unsaturated limbs, general addition = 14 asm blocks).
// Buckets and LDS hold the raw limbs (the partial sums' layout, G1UPart words per point); what leaves for the host (store_out)
// is the canonical saturated XYZZ the callers have always read.
template <class C> struct G1RedCurve {
  using Fq = typename C::Fq; using SAT = G1Curve<C>;
  static constexpr int N = Fq::N;
  using acc_t = g1_red<C>;
  static constexpr int ACC_WORDS = G1UPart<C>::WORDS, OUT_WORDS = SAT::ACC_WORDS;
  static constexpr bool QUAD = false, ZERO_IS_IDENTITY = true, INLINE_REDUCE_OPS = AVRF_G1_RED_INLINE != 0;
  static constexpr int RED_WAVES = SAT::RED_WAVES;
  struct accum {                                      // the partial sums k_accumulate<G1Curve<C>> left
    static constexpr int PART_WORDS = SAT::accum::PART_WORDS;
    static AVRF_DI acc_t load_part(const uint32_t *p) {
      const typename SAT::acc_t t = SAT::accum::load_part(p);
      return g1r_from_sat<C>(t.x.v, t.y.v, t.zz.v, t.zzz.v);
    }
  };
  static AVRF_DI acc_t identity() { return g1r_identity<C>(); }
  static AVRF_DI acc_t add(const acc_t &a, const acc_t &b) { return g1r_add<C>(a, b); }
  static AVRF_DI acc_t dbl(const acc_t &a) { return g1r_dbl<C>(a); }
  static AVRF_DI acc_t load_acc(const uint32_t *p) { return g1r_load<C>(p); }
  static AVRF_DI void store_acc(uint32_t *p, const acc_t &a) { g1r_store<C>(p, a); }
  static AVRF_DI void store_out(uint32_t *p, const acc_t &a) {
    typename SAT::acc_t t; g1r_to_sat<C>(a, t.x.v, t.y.v, t.zz.v, t.zzz.v); SAT::store_acc(p, t);
  }
  static AVRF_DI acc_t shfl_down(const acc_t &a, int delta) { return g1r_shfl_down<C>(a, delta); }
};

}  // namespace avrf
