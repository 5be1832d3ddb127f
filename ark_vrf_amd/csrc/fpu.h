// fpu.h -- unsaturated-limb ("carry-free") prime-field arithmetic for the bucket-accumulation kernels of the MSMs
// (reference call sites: the MSM of src/thin.rs:319, src/pedersen.rs:420, src/ring.rs:220).
//
// fp256.h / fpn.h keep a field element in N saturated 32-bit limbs: every limb product of a Montgomery multiplication is a
// v_mad_u64_u32 that can carry out of its 64-bit accumulator, so each one drags a v_addc_co_u32 behind it (the pair issues at
// 17.9 T/s against 31.7 T/s for the multiply-add alone, profiles/r4_ubench.txt).  Here an element is L SIGNED limbs of W bits
// (9 x 29 for the 251..255-bit fields, 14 x 28 for the 381-bit one): a column of the schoolbook product is at most L products
// of |a_i| <= 2^30 by |b_j| <= 2^29 plus L products m_i p_j < 2^(2W), which fits the signed 64-bit accumulator of
// v_mad_i64_i32 -- one instruction per limb product, no carry instruction, and the column's carry into the next one is one
// 64-bit arithmetic shift.  Additions and subtractions are limb-wise (no carry chain, no conditional subtraction of p): the
// spare bits above W hold them until the next multiplication, whose bounds (tools/fpu_model.py checks them by interval
// arithmetic and on random operands) say where a carry pass (fu_carry) is needed.
//
// Montgomery radix: R' = 2^(W L) = 2^261 (2^392), not the 2^256 (2^384) of the saturated form every table in HBM is written
// in.  A base coordinate b R (R = 2^(32 N)) is sliced into limbs of (b R) 2^SH, SH = W L - 32 N: the same shifts and masks as
// slicing b R itself, and mont'(a R, b R 2^SH) = a b R -- products by a base stay in the saturated form's Montgomery domain.
// Products of two accumulator coordinates come out as a b R 2^-SH; fpu_te.h / fpu_g1.h say how the point formulas absorb
// that constant.
#pragma once
#include "fp256.h"
#include "fpn.h"

namespace avrf {

template <int L> struct fu { int32_t v[L]; };
template <int L> struct ulimbs { uint32_t v[L]; };

// layout of field F in unsaturated limbs + its constants (evaluated at compile time from consts_gen.h's 32-bit limbs)
template <class F> struct UL {
  static constexpr int N = (int)(sizeof(F::P) / sizeof(uint32_t));
  static constexpr int W = N == 8 ? 29 : 28;
  static constexpr int L = N == 8 ? 9 : 14;
  static constexpr int SH = W * L - 32 * N;                       // 5 (8 limbs), 8 (12 limbs)
  static constexpr uint32_t MASK = (1u << W) - 1u;
  static constexpr uint32_t NINV = F::NINV & MASK;                // -p^-1 mod 2^W
  // p = 1 mod 2^W (Bandersnatch's base field): the reduction SUBTRACTS m_k p with m_k = the column's low W bits as they are -- no multiply
  // and no negation for m_k, and the column's own term m_k p_0 = m_k is what the arithmetic shift drops (floor).  Results differ from the
  // additive form's by exactly p (|value| < |a b| / 2^(W L) + p either way); tools/fpu_model.py and gen_fpu_asm.py take the same branch.
  static constexpr bool SUBTRACTIVE = (F::P[0] & MASK) == 1u && NINV == MASK;
  // limbs of K * value(c) (c: N words, K small): limb i = bits [W i, W i + W), the top limb takes what is left
  static constexpr ulimbs<L> slice_const(const uint32_t (&c)[N], uint32_t K) {
    uint32_t w[N + 2] = {};
    uint64_t cy = 0;
    for (int j = 0; j < N; j++) { cy += (uint64_t)c[j] * K; w[j] = (uint32_t)cy; cy >>= 32; }
    w[N] = (uint32_t)cy;
    ulimbs<L> r = {};
    for (int i = 0; i < L; i++) {
      const int o = W * i, j = o >> 5, sh = o & 31;
      const uint64_t two = (uint64_t)w[j] | ((uint64_t)w[j + 1] << 32);
      const uint32_t v = (uint32_t)(two >> sh);
      r.v[i] = i == L - 1 ? v : (v & MASK);
    }
    return r;
  }
  // limbs of value(c) * 2^k mod p (c < p): k modular doublings
  static constexpr ulimbs<L> shl_mod(const uint32_t (&c)[N], int k) {
    uint32_t w[N] = {};
    for (int j = 0; j < N; j++) w[j] = c[j];
    for (int s = 0; s < k; s++) {
      uint32_t cy = 0;
      for (int j = 0; j < N; j++) { const uint32_t t = (w[j] << 1) | cy; cy = w[j] >> 31; w[j] = t; }   // < 2p < 2^(32 N): the top bit of p is clear
      uint32_t d[N] = {}; uint64_t br = 0;
      for (int j = 0; j < N; j++) { const uint64_t t = (uint64_t)w[j] - F::P[j] - br; d[j] = (uint32_t)t; br = (t >> 32) & 1; }
      if (!br) for (int j = 0; j < N; j++) w[j] = d[j];
    }
    return slice_const(w, 1);
  }
  static constexpr ulimbs<L> P1 = slice_const(F::P, 1);           // p
  template <int KB> static constexpr ulimbs<L> PK = slice_const(F::P, 1u << KB);   // 2^KB p: makes a lazily reduced value positive (fu_to_packed)
  static constexpr ulimbs<L> ONE = slice_const(F::ONE, 1);        // R mod p: the saturated form's Montgomery one
};

template <class F> using fuF = fu<UL<F>::L>;
}  // namespace avrf
#include "fpu_asm_gen.h"     // fu_mul_asm / fu_sqr_asm: the products below as single asm blocks for the 8-word fields (tools/gen_fpu_asm.py)
namespace avrf {

// limbs of value(w) * 2^S (S = 0, or UL::SH for a base coordinate that is going to be multiplied): one v_alignbit_b32 (or shift) + one v_and_b32 each
template <class F, int S> AVRF_DI fuF<F> fu_slice(const uint32_t (&w)[UL<F>::N]) {
  using U = UL<F>;
  fuF<F> r;
#pragma unroll
  for (int i = 0; i < U::L; i++) {
    const int o = U::W * i - S;
    uint32_t v;
    if (o < 0) v = w[0] << (-o);
    else {
      const int j = o >> 5, sh = o & 31;
      const uint32_t lo = w[j], hi = (j + 1 < U::N) ? w[(j + 1 < U::N) ? j + 1 : j] : 0u;
      v = sh ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo;
    }
    r.v[i] = (int32_t)(i == U::L - 1 ? v : (v & U::MASK));
  }
  return r;
}
template <class F> AVRF_DI fuF<F> fu_const(const ulimbs<UL<F>::L> &c) {
  fuF<F> r;
#pragma unroll
  for (int i = 0; i < UL<F>::L; i++) r.v[i] = (int32_t)c.v[i];
  return r;
}
template <int L> AVRF_DI fu<L> fu_zero() { fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = 0;
  return r; }
template <int L> AVRF_DI fu<L> fu_add(const fu<L> &a, const fu<L> &b) { fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = a.v[i] + b.v[i];
  return r; }
template <int L> AVRF_DI fu<L> fu_sub(const fu<L> &a, const fu<L> &b) { fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = a.v[i] - b.v[i];
  return r; }
// s = 0 / -1 (all ones): a or -a, two instructions per limb
template <int L> AVRF_DI fu<L> fu_cneg(const fu<L> &a, int32_t s) { fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = (a.v[i] ^ s) - s;
  return r; }
// one parallel carry pass: limbs of magnitude < 2^31 -> limbs in [0, 2^W + 4) (top limb: signed, what is left).  The value is unchanged.
template <class F> AVRF_DI fuF<F> fu_carry(const fuF<F> &a) {
  using U = UL<F>;
  fuF<F> r;
  r.v[0] = a.v[0] & (int32_t)U::MASK;
#pragma unroll
  for (int i = 1; i < U::L - 1; i++) r.v[i] = (a.v[i] & (int32_t)U::MASK) + (a.v[i - 1] >> U::W);
  r.v[U::L - 1] = a.v[U::L - 1] + (a.v[U::L - 2] >> U::W);
  return r;
}

// the same pass for sums of NON-NEGATIVE limbs that need all 32 bits (B + 5 A < 6 * 2^29): unsigned arithmetic and logical shifts
// for limbs 0 .. L-2 (a signed int32 would overflow -- undefined behaviour, however the hardware wraps); the top limb stays signed
template <class F> AVRF_DI fuF<F> fu_carry_u(const uint32_t (&a)[UL<F>::L - 1], int32_t top) {
  using U = UL<F>;
  fuF<F> r;
  r.v[0] = (int32_t)(a[0] & U::MASK);
#pragma unroll
  for (int i = 1; i < U::L - 1; i++) r.v[i] = (int32_t)((a[i] & U::MASK) + (a[i - 1] >> U::W));
  r.v[U::L - 1] = top + (int32_t)(a[U::L - 2] >> U::W);
  return r;
}
// 5 a for a with limbs 0 .. L-2 in [0, 2^W + 4) (a product's output, or carried), carried to limbs in [0, 2^W + 4)
template <class F> AVRF_DI fuF<F> fu_times5(const fuF<F> &a) {
  constexpr int L = UL<F>::L;
  uint32_t t[L - 1];
#pragma unroll
  for (int i = 0; i < L - 1; i++) t[i] = 5u * (uint32_t)a.v[i];
  return fu_carry_u<F>(t, 5 * a.v[L - 1]);
}
// a * b / 2^(W L) mod p for limbs with L |a_i| |b_j| + L 2^(2W) < 2^63 (e.g. |a_i| <= 2^30, |b_j| <= 2^29 + 2^4), product scanning with the
// reduction interleaved: column k gets its k + 1 (or fewer) limb products and the products m_i p_(k-i) of the reduction so far, m_k
// is chosen to clear the column's low W bits, and the rest of the accumulator moves down by W bits into column k + 1.
// Result: limbs 0 .. L-2 in [0, 2^W), limb L-1 signed; |value| < |a b| / 2^(W L) + p.
template <class F> AVRF_DI fuF<F> fu_mul(const fuF<F> &a, const fuF<F> &b) {
#if !defined(AVRF_NO_FPU_ASM) && defined(__HIP_DEVICE_COMPILE__)
  if constexpr (FuAsm<F>::value) return fu_mul_asm(F{}, a, b);
#endif
  using U = UL<F>;
  constexpr int L = U::L, W = U::W;
  int64_t acc = 0;
  int32_t m[L];
  fuF<F> r;
#pragma unroll
  for (int k = 0; k < L; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (int64_t)a.v[i] * (int64_t)b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (U::SUBTRACTIVE ? -(int64_t)(int32_t)U::P1.v[k - i] : (int64_t)(int32_t)U::P1.v[k - i]);
    if constexpr (U::SUBTRACTIVE) m[k] = (int32_t)((uint32_t)acc & U::MASK);
    else {
      m[k] = (int32_t)(((uint32_t)acc * U::NINV) & U::MASK);
      acc += (int64_t)m[k] * (int64_t)(int32_t)U::P1.v[0];
    }
    acc >>= W;
  }
#pragma unroll
  for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (int64_t)a.v[i] * (int64_t)b.v[k - i];
#pragma unroll
    for (int i = k - L + 1; i < L; i++) acc += (int64_t)m[i] * (U::SUBTRACTIVE ? -(int64_t)(int32_t)U::P1.v[k - i] : (int64_t)(int32_t)U::P1.v[k - i]);
    r.v[k - L] = (int32_t)((uint32_t)acc & U::MASK);
    acc >>= W;
  }
  r.v[L - 1] = (int32_t)acc;
  return r;
}
// a * a / 2^(W L): L (L + 1) / 2 limb products (the cross products against the doubled limbs); |a_i| <= 2^29 + 2^4
template <class F> AVRF_DI fuF<F> fu_sqr(const fuF<F> &a) {
#if !defined(AVRF_NO_FPU_ASM) && !defined(AVRF_NO_FPU_ASM_SQR) && defined(__HIP_DEVICE_COMPILE__)
  if constexpr (FuAsm<F>::value) return fu_sqr_asm(F{}, a);
#endif
  using U = UL<F>;
  constexpr int L = U::L, W = U::W;
  int64_t acc = 0;
  int32_t m[L], d[L];
  fuF<F> r;
#pragma unroll
  for (int i = 0; i < L; i++) d[i] = a.v[i] << 1;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
    const int i0 = k < L ? 0 : k - L + 1;
#pragma unroll
    for (int i = i0; 2 * i <= k; i++) acc += (int64_t)a.v[i] * (int64_t)(2 * i == k ? a.v[i] : d[k - i]);
#pragma unroll
    for (int i = i0; i < (k < L ? k : L); i++) acc += (int64_t)m[i] * (U::SUBTRACTIVE ? -(int64_t)(int32_t)U::P1.v[k - i] : (int64_t)(int32_t)U::P1.v[k - i]);
    if (k < L) {
      if constexpr (U::SUBTRACTIVE) m[k] = (int32_t)((uint32_t)acc & U::MASK);
      else {
        m[k] = (int32_t)(((uint32_t)acc * U::NINV) & U::MASK);
        acc += (int64_t)m[k] * (int64_t)(int32_t)U::P1.v[0];
      }
    } else r.v[k - L] = (int32_t)((uint32_t)acc & U::MASK);
    acc >>= W;
  }
  r.v[L - 1] = (int32_t)acc;
  return r;
}

// the canonical value (< p) of a lazily reduced element with |value| < 2^KB p, as N saturated words: add 2^KB p, one exact
// carry pass, repack, subtract 2^KB p .. 2p, p where it fits.  Off the hot path: once per partial sum, in its reader.
template <class F, int KB = 2> AVRF_DI void fu_to_packed(uint32_t (&w)[UL<F>::N], const fuF<F> &a) {
  using U = UL<F>;
  constexpr int L = U::L, W = U::W, N = U::N;
  uint32_t u[L];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < L - 1; i++) { const int32_t t = a.v[i] + (int32_t)U::template PK<KB>.v[i] + c; u[i] = (uint32_t)t & U::MASK; c = t >> W; }
  u[L - 1] = (uint32_t)(a.v[L - 1] + (int32_t)U::template PK<KB>.v[L - 1] + c);   // >= 0: the value is in (0, 2^(KB+1) p)
  uint32_t x[N + 1];
#pragma unroll
  for (int j = 0; j <= N; j++) {                                             // word j = bits [32 j, 32 j + 32)
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int lo = W * i - 32 * j;                                         // position of limb i's bit 0 in this word
      if (lo >= 32 || lo + 32 <= 0) continue;                                // (the top limb may be up to 32 bits wide)
      if (lo >= 0) v |= u[i] << lo; else v |= u[i] >> (-lo);
    }
    x[j] = v;
  }
#pragma unroll
  for (int K = 1 << KB; K >= 1; K >>= 1) {
    uint32_t kp[N + 1]; uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < N; j++) { cy += (uint64_t)F::P[j] * (uint32_t)K; kp[j] = (uint32_t)cy; cy >>= 32; }
    kp[N] = (uint32_t)cy;
    uint32_t d[N + 1]; unsigned br = 0;
#pragma unroll
    for (int j = 0; j <= N; j++) d[j] = __builtin_subc(x[j], kp[j], br, &br);
#pragma unroll
    for (int j = 0; j <= N; j++) x[j] = br ? x[j] : d[j];
  }
#pragma unroll
  for (int j = 0; j < N; j++) w[j] = x[j];
}

}  // namespace avrf
