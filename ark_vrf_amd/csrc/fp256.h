// fp256.h -- 256-bit prime-field arithmetic for gfx950 VALU (8 x u32 limbs, Montgomery R = 2^256).
//
// Device counterpart of what the reference gets from arkworks `Fp<MontBackend<_,4>>`
// (third-party ark-ff 0.6; reached from src/thin.rs:289-311, src/pedersen.rs:373-410 for the
// scalar field and from every group operation for the base field).  One field element per
// lane, limbs in VGPRs; products through v_mad_u64_u32.  The twisted-Edwards suites' moduli have their
// top bit clear (251..255 bits), which admits the carry-free forms; the two 256-bit fields of secp256r1
// (F::FULL) keep the 257th bit of a sum or a Montgomery product and fold it into the conditional subtraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts_gen.h"
#include "mac96.h"

namespace avrf {

#define AVRF_DI __device__ __forceinline__

struct fp { uint32_t v[8]; };

template <class F> AVRF_DI fp fp_const(const uint32_t (&c)[8]) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c[i];
  return r;
}
template <class F> AVRF_DI fp fp_one() { return fp_const<F>(F::ONE); }
AVRF_DI fp fp_zero() { fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r; }

AVRF_DI bool fp_is_zero(const fp &a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
AVRF_DI bool fp_eq(const fp &a, const fp &b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a + b, returns carry
// (carry chains through __builtin_addc / __builtin_subc: one v_addc_co_u32 / v_subb_co_u32 per limb.  The uint64_t / int64_t
// accumulator idiom compiled to ~90 instructions per field addition -- 64-bit adds, arithmetic shifts and the moves that build
// their register pairs -- against ~30 in this form; a mixed addition has eleven of them.)
AVRF_DI uint32_t add8(fp &r, const fp &a, const fp &b) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  return c;
}
// r = a - b, returns borrow (0/1)
AVRF_DI uint32_t sub8(fp &r, const fp &a, const fp &b) {
  unsigned br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
  return br;
}
template <class F> AVRF_DI uint32_t sub_p(fp &r, const fp &a) {
  unsigned br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], (unsigned)F::P[i], br, &br);
  return br;
}
// a >= p ?  (plain integer compare)
template <class F> AVRF_DI bool ge_p(const fp &a) { fp t; return sub_p<F>(t, a) == 0; }

template <class F> AVRF_DI fp fp_add(const fp &a, const fp &b) {
  fp t, u; uint32_t c = add8(t, a, b);  // < 2p; a carry out only when the top bit of p is set (F::FULL)
  uint32_t br = sub_p<F>(u, t);
  if constexpr (F::FULL) br = br && !c; else (void)c;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = br ? t.v[i] : u.v[i];
  return t;
}
template <class F> AVRF_DI fp fp_sub(const fp &a, const fp &b) {
  fp t; const uint32_t m = 0u - sub8(t, a, b);           // borrow: add p back
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = __builtin_addc(t.v[i], (unsigned)(F::P[i] & m), c, &c);
  return t;
}
template <class F> AVRF_DI fp fp_neg(const fp &a) {
  fp t; unsigned br = 0; const bool z = fp_is_zero(a);
#pragma unroll
  for (int i = 0; i < 8; i++) { const unsigned d = __builtin_subc((unsigned)F::P[i], a.v[i], br, &br); t.v[i] = z ? 0u : d; }
  return t;
}
template <class F> AVRF_DI fp fp_dbl(const fp &a) { return fp_add<F>(a, a); }

// Montgomery product a*b/R mod p (top bit of p clear), product scanning (mac96.h)
template <class F> AVRF_DI fp fp_mul(const fp &a, const fp &b) {
  fp r, u;
  uint32_t c = mont_mul_ps<8, F>(r.v, a.v, b.v);
  uint32_t br = sub_p<F>(u, r);
  if constexpr (F::FULL) br = br && !c; else (void)c;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}
template <class F> AVRF_DI fp fp_sqr(const fp &a) {
  fp r, u;
  uint32_t c = mont_sqr_ps<8, F>(r.v, a.v);
  uint32_t br = sub_p<F>(u, r);
  if constexpr (F::FULL) br = br && !c; else (void)c;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}

template <class F> AVRF_DI fp fp_to_mont(const fp &a) { return fp_mul<F>(a, fp_const<F>(F::R2)); }
template <class F> AVRF_DI fp fp_from_mont(const fp &a) {
  fp one = fp_zero(); one.v[0] = 1; return fp_mul<F>(a, one);
}

// a^e for a constant exponent (plain 256-bit integer), square-and-multiply MSB first.
template <class F> AVRF_DI fp fp_pow_const(const fp &a, const uint32_t (&e)[8]) {
  fp r = fp_one<F>();
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) r = fp_sqr<F>(r);
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? fp_mul<F>(r, a) : a; started = true; }
  }
  return r;
}
template <class F> AVRF_DI fp fp_inv(const fp &a) { return fp_pow_const<F>(a, F::PM2); }


// Square root by Tonelli-Shanks (p - 1 = 2^s * t, ROOT = g^t for a non-residue g).  Returns
// false when `a` is a non-residue.  Which of the two roots is returned is unspecified; callers
// choose by sign (ark-serialize's x-sign flag).
template <class F> AVRF_DI bool fp_sqrt(const fp &a, fp &out) {
  if (fp_is_zero(a)) { out = a; return true; }
  const fp one = fp_one<F>();
  fp z = fp_const<F>(F::ROOT);
  fp w = fp_pow_const<F>(a, F::T_MINUS1_HALF);
  fp x = fp_mul<F>(w, a);          // a^((t+1)/2)
  fp b = fp_mul<F>(x, w);          // a^t
  int v = F::TWO_ADICITY;
  while (!fp_eq(b, one)) {
    int k = 0; fp b2k = b;
    while (!fp_eq(b2k, one)) { b2k = fp_sqr<F>(b2k); k++; if (k >= v) return false; }
    fp ww = z;
    for (int j = 0; j < v - k - 1; j++) ww = fp_sqr<F>(ww);
    z = fp_sqr<F>(ww);
    b = fp_mul<F>(b, z);
    x = fp_mul<F>(x, ww);
    v = k;
  }
  if (!fp_eq(fp_sqr<F>(x), a)) return false;
  out = x; return true;
}

// x > (p-1)/2 on the canonical value of a Montgomery-form element ("negative" in ark-serialize)
template <class F> AVRF_DI bool fp_is_negative_mont(const fp &a_mont) {
  fp a = fp_from_mont<F>(a_mont), t;
  // a > HALF  <=>  HALF - a borrows
  return sub8(t, fp_const<F>(F::HALF), a) != 0;
}
template <class F> AVRF_DI bool fp_is_negative_plain(const fp &a) {
  fp t; return sub8(t, fp_const<F>(F::HALF), a) != 0;
}

// little-endian bytes <-> limbs (global or local memory)
AVRF_DI fp fp_load_le(const uint8_t *p) {
  fp r;
  if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++)
      r.v[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  }
  return r;
}
AVRF_DI void fp_store_le(uint8_t *p, const fp &a) {
  if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = a.v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) { p[4 * i] = (uint8_t)a.v[i]; p[4 * i + 1] = (uint8_t)(a.v[i] >> 8); p[4 * i + 2] = (uint8_t)(a.v[i] >> 16); p[4 * i + 3] = (uint8_t)(a.v[i] >> 24); }
  }
}

// value of `nbytes` little-endian bytes (nbytes <= 64) reduced mod p, Montgomery form.
// Mirrors PrimeField::from_le_bytes_mod_order as used by src/utils/common.rs:65-76.
template <class F> AVRF_DI fp fp_from_le_bytes_mod_order_16(const uint32_t w[4]) {
  // 128-bit value < p for every field on the path (p > 2^250): just convert.
  fp a = fp_zero();
#pragma unroll
  for (int i = 0; i < 4; i++) a.v[i] = w[i];
  return fp_to_mont<F>(a);
}
// 512-bit (lo + hi*2^256): lo*R2/R + hi*R3/R... computed as mont(lo,R2) + mont(mont(hi,R2),R2)
template <class F> AVRF_DI fp fp_from_wide_mont(const fp &lo, const fp &hi) {
  fp r2 = fp_const<F>(F::R2);
  fp l = fp_mul<F>(lo, r2);              // lo * R        (lo may be >= p: mont mul handles any a < 2^256 with b < p)
  fp h = fp_mul<F>(fp_mul<F>(hi, r2), r2); // hi * R^2 = (hi * 2^256) * R
  return fp_add<F>(l, h);
}


// ---- out-of-line variants for the per-item protocol kernels (code size / compile time);
// the MSM hot loops keep the force-inlined forms above.
#define AVRF_DN __device__ __noinline__ static
template <class F> AVRF_DN fp fp_mul_nf(fp a, fp b) { return fp_mul<F>(a, b); }
template <class F> AVRF_DI fp fp_sqr_nf(const fp &a) { return fp_mul_nf<F>(a, a); }
template <class F> AVRF_DI fp fp_to_mont_nf(const fp &a) { return fp_mul_nf<F>(a, fp_const<F>(F::R2)); }
template <class F> AVRF_DI fp fp_from_mont_nf(const fp &a) { fp one = fp_zero(); one.v[0] = 1; return fp_mul_nf<F>(a, one); }
template <class F> AVRF_DI bool fp_is_negative_mont_nf(const fp &a_mont) {
  fp a = fp_from_mont_nf<F>(a_mont), t; return sub8(t, fp_const<F>(F::HALF), a) != 0;
}
template <class F> AVRF_DI fp fp_from_wide_mont_nf(const fp &lo, const fp &hi) {
  fp r2 = fp_const<F>(F::R2);
  return fp_add<F>(fp_mul_nf<F>(lo, r2), fp_mul_nf<F>(fp_mul_nf<F>(hi, r2), r2));
}
// a^-1 (0 -> 0) for the lane-per-item kernels: the binary GCD with ONE fused step per iteration, written without branches so that the
// 64 different values of a wave walk the same instruction stream (only the trip count differs, ~1.4 x 255 +- a few):
//   u even:          u <- u / 2,        x1 <- x1 / 2
//   u odd, u >= v:   u <- (u - v) / 2,  x1 <- (x1 - x2) / 2
//   u odd, u <  v:   (u, v) <- ((v - u) / 2, u),  (x1, x2) <- ((x2 - x1) / 2, x1)
// with x1 a = u, x2 a = v (mod p), v odd throughout; u = 0 leaves v = 1, x2 = a^-1.  ~120 carry / select instructions per step,
// ~45 k per inversion against ~110 k (two thirds multiply-adds) for the fixed power a^(p-2) it replaces: 0.23 -> ~0.1 ms of a
// prover kernel's single wave per SIMD.  (The few-items kernels use fp_inv_few below: same data in all lanes, plain loops.)
template <class F> AVRF_DN fp fp_inv_nf(fp a) {
  const fp P = fp_const<F>(F::P);
  fp u = a, v = P, x1 = fp_zero(), x2 = fp_zero();
  x1.v[0] = 1;
#pragma unroll 1
  while (!fp_is_zero(u)) {
    const bool odd = (u.v[0] & 1u) != 0;
    fp d1, d2;
    const bool lt = sub8(d1, u, v) != 0;                   // d1 = u - v, d2 = v - u
    sub8(d2, v, u);
    const bool sw = odd && lt;
    fp xa, xb;                                            // minuend / subtrahend of the x update
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t un = odd ? (lt ? d2.v[i] : d1.v[i]) : u.v[i];
      v.v[i] = sw ? u.v[i] : v.v[i];
      u.v[i] = un;
      xa.v[i] = sw ? x2.v[i] : x1.v[i];
      xb.v[i] = odd ? (sw ? x1.v[i] : x2.v[i]) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) u.v[i] = (u.v[i] >> 1) | (u.v[i + 1] << 31);
    u.v[7] >>= 1;
#pragma unroll
    for (int i = 0; i < 8; i++) x2.v[i] = sw ? x1.v[i] : x2.v[i];
    fp t = fp_sub<F>(xa, xb);                             // in [0, p)
    uint32_t c = 0;
    { const uint32_t m = 0u - (t.v[0] & 1u); fp pm;        // t / 2 mod p: (t + p) / 2 when t is odd
#pragma unroll
      for (int i = 0; i < 8; i++) pm.v[i] = F::P[i] & m;
      c = add8(t, t, pm); }
#pragma unroll
    for (int i = 0; i < 7; i++) x1.v[i] = (t.v[i] >> 1) | (t.v[i + 1] << 31);
    x1.v[7] = (t.v[7] >> 1) | (c << 31);
  }
  const fp r2 = fp_const<F>(F::R2);                        // x2 = (a' R)^-1 for a = a' R; times R^3 / R gives a'^-1 R
  return fp_mul_nf<F>(x2, fp_mul_nf<F>(r2, r2));
}
// The Jacobi symbols (a / p), (b / p) = the quadratic characters of a and b (p prime): +1, -1, or 0 for a = 0.  Exact binary algorithm on
// (a, n), n odd, in MACRO steps -- one step removes ALL trailing zeros of a (v_ffbl_b32 + a funnel shift per limb), then subtracts:
//   z = ctz(a) (at most 30 per step):  a <- a / 2^z            sign *= (2 / n)^z,  (2 / n) = -1 iff n = 3, 5 mod 8
//   a odd, a >= n:                     a <- a - n
//   a odd, a <  n:                     (a, n) <- (n - a, a)    sign *= (-1)^((a-1)(n-1)/4)            (quadratic reciprocity)
// ~180 macro steps for a 255-bit modulus (every subtraction leaves two trailing zeros on average) instead of ~360 single-bit steps, and the
// values shrink by ~1.4 bits each per step, so the wave walks the steps in PHASES of K = 8, 7, .. 1 live limbs (K = the longest value any of
// its lanes still holds; a phase is left when limb K-1 of every lane's a and n is zero): 6 K + 11 instructions per step and symbol, ~8 k
// instructions per symbol against the ~22 k of the single-bit form on all eight limbs (tools/jac_model.py counts both) and the ~50 k of
// Euler's criterion.  Two symbols share one loop (the subgroup test needs two: glv.h), which also gives every carry chain an independent
// neighbour.  No branches inside a step; every lane walks the same stream; a lane that is done (a = 0) idles harmlessly: its z is 30 (even:
// no sign change), it is never odd.  The loop ends on any input: a step with a odd shortens len(a) + len(n), a step with a even shifts.
// `a` may be given in Montgomery form: R = (2^128)^2 is a square, so (a R / p) = (a / p).
struct jac_state { uint32_t a[8], n[8], t; };
template <int K> AVRF_DI void jac_step(jac_state &s) {
  uint32_t z = s.a[0] ? (uint32_t)__builtin_ctz(s.a[0]) : 32u;
  z = z < 30u ? z : 30u;
#pragma unroll
  for (int i = 0; i + 1 < K; i++) s.a[i] = __builtin_amdgcn_alignbit(s.a[i + 1], s.a[i], z);
  s.a[K - 1] >>= z;
  s.t ^= (s.n[0] ^ (s.n[0] >> 1)) & (z << 1);                 // bit 1 of t: the sign so far is -1
  const bool odd = (s.a[0] & 1u) != 0;
  uint32_t d[K], e[K];
  unsigned br = 0, br2 = 0;
#pragma unroll
  for (int i = 0; i < K; i++) d[i] = __builtin_subc(s.a[i], s.n[i], br, &br);      // a - n; borrow: a < n
#pragma unroll
  for (int i = 0; i < K; i++) e[i] = __builtin_subc(s.n[i], s.a[i], br2, &br2);    // n - a
  const bool lt = br != 0, sw = odd && lt;
  s.t ^= sw ? (s.a[0] & s.n[0]) : 0u;                         // both 3 mod 4
#pragma unroll
  for (int i = 0; i < K; i++) {
    const uint32_t x = lt ? e[i] : d[i];
    s.n[i] = sw ? s.a[i] : s.n[i];
    s.a[i] = odd ? x : s.a[i];
  }
}
// one phase: steps on K limbs until every lane is finished (returns true) or may drop limb K - 1 (false)
template <int K> AVRF_DI bool jac_phase(jac_state &s1, jac_state &s2, int &budget) {
#pragma unroll 1
  for (;;) {
    uint32_t z1 = 0, z2 = 0;
#pragma unroll
    for (int i = 0; i < K; i++) { z1 |= s1.a[i]; z2 |= s2.a[i]; }
    if (__all((z1 | z2) == 0) || budget <= 0) return true;
    if constexpr (K > 1)
      if (__all((z1 == 0 || (s1.a[K - 1] | s1.n[K - 1]) == 0) && (z2 == 0 || (s2.a[K - 1] | s2.n[K - 1]) == 0))) return false;
#pragma unroll
    for (int r = 0; r < 4; r++) { jac_step<K>(s1); jac_step<K>(s2); }
    budget -= 4;
  }
}
AVRF_DI int jac_result(const jac_state &s) {
  uint32_t za = 0, zn = s.n[0] ^ 1u;                          // a = 0 and n = gcd = 1 (a finished lane's n is never touched again)
#pragma unroll
  for (int i = 0; i < 8; i++) za |= s.a[i];
#pragma unroll
  for (int i = 1; i < 8; i++) zn |= s.n[i];
  return (za | zn) ? 0 : ((s.t & 2u) ? -1 : 1);
}
template <class F> AVRF_DN void fp_jacobi2_nf(fp a, fp b, int *ja, int *jb) {
  jac_state s1, s2;
#pragma unroll
  for (int i = 0; i < 8; i++) { s1.a[i] = a.v[i]; s2.a[i] = b.v[i]; s1.n[i] = s2.n[i] = F::P[i]; }
  s1.t = s2.t = 0;
  int budget = 1024;                                           // (at most 2 x 256 shortening steps + the steps a zero low word takes)
  (void)(jac_phase<8>(s1, s2, budget) || jac_phase<7>(s1, s2, budget) || jac_phase<6>(s1, s2, budget) || jac_phase<5>(s1, s2, budget) ||
         jac_phase<4>(s1, s2, budget) || jac_phase<3>(s1, s2, budget) || jac_phase<2>(s1, s2, budget) || jac_phase<1>(s1, s2, budget));
  *ja = jac_result(s1); *jb = jac_result(s2);
}
// the fixed power a^(p-2), kept as the cross-check of the two Euclidean forms (tools/ubench.hip)
template <class F> AVRF_DN fp fp_inv_fermat_nf(fp a) {
  fp r = fp_one<F>();
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) r = fp_mul_nf<F>(r, r);
    if ((F::PM2[i >> 5] >> (i & 31)) & 1) { r = started ? fp_mul_nf<F>(r, a) : a; started = true; }
  }
  return r;
}
// a^-1 of a Montgomery-form a (0 -> 0, as a^(p-2) gives) by the binary extended Euclidean algorithm on the plain integers:
// ~1.4 * 255 halvings and ~0.7 * 255 subtractions of 8-limb values (~20 k carry instructions) instead of the 255 squarings and
// ~130 products of the fixed power (~110 k instructions, two thirds of them multiply-adds).  For the LATENCY kernels only
// (vrf_single.hip "few items"): there the lanes of an item hold the same value, so the data-dependent loops do not diverge (two
// items in a wave: at most both paths).  The lane-per-item kernels keep the fixed-length power -- 64 different trip counts per
// wave would serialise.  Invariants: x1 a = u, x2 a = v (mod p).
template <class F> AVRF_DN fp fp_inv_few(fp a) {
  if (fp_is_zero(a)) return a;
  const fp P = fp_const<F>(F::P);
  fp u = a, v = P, x1 = fp_zero(), x2 = fp_zero();
  x1.v[0] = 1;
  auto is_one = [](const fp &x) { uint32_t o = x.v[0] ^ 1u; for (int i = 1; i < 8; i++) o |= x.v[i]; return o == 0; };
  auto shr1 = [](fp &x, uint32_t top) { for (int i = 0; i < 7; i++) x.v[i] = (x.v[i] >> 1) | (x.v[i + 1] << 31); x.v[7] = (x.v[7] >> 1) | (top << 31); };
  auto halve_mod = [&](fp &x) { uint32_t c = 0; if (x.v[0] & 1u) c = add8(x, x, P); shr1(x, c); };
  // (u = 0 cannot happen for a canonical non-zero a -- gcd(a, p) = 1 -- but a loop on the device must end whatever it is fed)
#pragma unroll 1
  while (!is_one(u) && !is_one(v) && !fp_is_zero(u)) {
#pragma unroll 1
    while (!(u.v[0] & 1u)) { shr1(u, 0); halve_mod(x1); }
#pragma unroll 1
    while (!(v.v[0] & 1u)) { shr1(v, 0); halve_mod(x2); }
    fp t;
    if (sub8(t, u, v) == 0) { u = t; x1 = fp_sub<F>(x1, x2); }
    else { sub8(v, v, u); x2 = fp_sub<F>(x2, x1); }
  }
  const fp r = is_one(u) ? x1 : x2;                       // (a' R)^-1 for a = a' R; times R^3 / R gives a'^-1 R
  const fp r2 = fp_const<F>(F::R2);
  return fp_mul_nf<F>(r, fp_mul_nf<F>(r2, r2));
}
template <class F> AVRF_DN fp fp_pow_nf(fp a, int which) {   // which: 0 -> T_MINUS1_HALF
  fp r = fp_one<F>();
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) r = fp_mul_nf<F>(r, r);
    uint32_t bit = which == 0 ? (F::T_MINUS1_HALF[i >> 5] >> (i & 31)) & 1 : (F::PM2[i >> 5] >> (i & 31)) & 1;
    if (bit) { r = started ? fp_mul_nf<F>(r, a) : a; started = true; }
  }
  return r;
}
// sqrt(u / v) without an inversion and without a data-dependent loop (the lanes of a wave decompress 64 different points: the
// classic Tonelli-Shanks loop below runs every lane for the LONGEST of their discrete-log walks -- up to 32 x 32 squarings on a field
// with 2-adicity 32 -- and the quotient cost a ~45 k-instruction inversion first).  With p - 1 = 2^s t, a = u v, w = a^((t-1)/2):
//   a w^2 = a^t = g^e (g = ROOT, order 2^s),     1 / sqrt(a) = w g^(-e/2),     sqrt(u / v) = u / sqrt(u v) = u w g^(-e/2);
// e is read in 8-bit windows from the low end: the window's value j is looked up from d = c^(2^(s - 8i - w)), a 2^w-th root of unity
// (sqrt_window below: a perfect hash of the 256 roots on the low word of d), then c <- c g^(-j 256^i) (tables SQRT_HIDX / SQRT_G /
// SQRT_GH of consts_gen.h, tools/gen_consts.py).  48 squarings + 8 products for s = 32 after the one fixed exponentiation (4-bit
// windows with compare loops took 112 + 16); the same instruction stream in every lane.  Returns false when u / v is not a square
// (e odd), and checks x^2 v = u.  u = 0 gives x = 0; v = 0 is the caller's case.
template <class F> AVRF_DI uint32_t sqrt_window(uint32_t d_low, int wd) {        // d = h^(j 2^(hw - wd)) in canonical Montgomery form -> j
  return (uint32_t)F::SQRT_HIDX[(d_low * F::SQRT_HMUL) >> (32 - F::SQRT_HBITS)] >> (F::SQRT_HW - wd);
}
template <class F> AVRF_DN bool fp_sqrt_ratio_nf(fp u, fp v, fp *out) {
  constexpr int S = F::TWO_ADICITY, W = F::SQRT_W;
  const fp a = fp_mul_nf<F>(u, v);
  const fp w = fp_pow_nf<F>(a, 0);                                   // a^((t-1)/2)
  fp c = fp_mul_nf<F>(a, fp_mul_nf<F>(w, w));                        // a^t, in the group of 2^s-th roots of unity
  fp r = fp_mul_nf<F>(u, w);
  bool odd = false;
#pragma unroll 1
  for (int i = 0; i < F::SQRT_STEPS; i++) {
    const int wd = S - W * i < W ? S - W * i : W;                    // bits of this window
    fp d = c;
#pragma unroll 1
    for (int k = 0; k < S - W * i - wd; k++) d = fp_mul_nf<F>(d, d);
    const uint32_t j = sqrt_window<F>(d.v[0], wd);
    if (i == 0 && (j & 1u)) odd = true;
    fp gs, gh;
#pragma unroll
    for (int l = 0; l < 8; l++) { gs.v[l] = F::SQRT_G[i][j][l]; gh.v[l] = F::SQRT_GH[i][j][l]; }
    c = fp_mul_nf<F>(c, gs);
    r = fp_mul_nf<F>(r, gh);
  }
  *out = r;
  return !odd && fp_eq(fp_mul_nf<F>(fp_mul_nf<F>(r, r), v), u);
}
// Tonelli-Shanks, out-of-line (see fp_sqrt)
template <class F> AVRF_DN bool fp_sqrt_nf(fp a, fp *out) {
  if (fp_is_zero(a)) { *out = a; return true; }
  const fp one = fp_one<F>();
  fp z = fp_const<F>(F::ROOT);
  fp w = fp_pow_nf<F>(a, 0);
  fp x = fp_mul_nf<F>(w, a);
  fp b = fp_mul_nf<F>(x, w);
  int v = F::TWO_ADICITY;
  while (!fp_eq(b, one)) {
    int k = 0; fp b2k = b;
    while (!fp_eq(b2k, one)) { b2k = fp_mul_nf<F>(b2k, b2k); k++; if (k >= v) return false; }
    fp ww = z;
    for (int j = 0; j < v - k - 1; j++) ww = fp_mul_nf<F>(ww, ww);
    z = fp_mul_nf<F>(ww, ww);
    b = fp_mul_nf<F>(b, z);
    x = fp_mul_nf<F>(x, ww);
    v = k;
  }
  if (!fp_eq(fp_mul_nf<F>(x, x), a)) return false;
  *out = x; return true;
}

}  // namespace avrf
