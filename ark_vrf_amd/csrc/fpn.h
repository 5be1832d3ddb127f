// fpn.h -- N x u32-limb prime-field arithmetic for gfx950 VALU (Montgomery, R = 2^(32 N)); N = 12 for the
// BLS12-381 base field (381 bits), N = 8 for BN254's.  Device counterpart of arkworks
// `Fp<MontBackend<_, 6>>` / `<_, 4>` behind the KZG commit/open MSMs of the ring SNARK
// (w3f-ring-proof, reached from src/ring.rs:220,404,416,731).  Same carry-free CIOS as fp256.h
// (every modulus on the path leaves its top limb's high bit clear).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts_gen.h"
#include "mac96.h"

namespace avrf {

#ifndef AVRF_DI
#define AVRF_DI __device__ __forceinline__
#endif

template <int N> struct fpn { uint32_t v[N]; };
template <class F> using fe = fpn<F::N>;

template <int N> AVRF_DI fpn<N> fn_zero() { fpn<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = 0;
  return r; }
template <class F> AVRF_DI fe<F> fn_const(const uint32_t (&c)[F::N]) { fe<F> r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = c[i];
  return r; }
template <class F> AVRF_DI fe<F> fn_one() { return fn_const<F>(F::ONE); }
template <int N> AVRF_DI bool fn_is_zero(const fpn<N> &a) { uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a.v[i];
  return o == 0; }
template <int N> AVRF_DI bool fn_eq(const fpn<N> &a, const fpn<N> &b) { uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a.v[i] ^ b.v[i];
  return o == 0; }

template <class F> AVRF_DI uint32_t fn_sub_p(fe<F> &r, const fe<F> &a) {
  unsigned br = 0;                                       // (carry builtins: see fp256.h add8)
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = __builtin_subc(a.v[i], (unsigned)F::P[i], br, &br);
  return br;
}
template <class F> AVRF_DI bool fn_ge_p(const fe<F> &a) { fe<F> t; return fn_sub_p<F>(t, a) == 0; }
template <class F> AVRF_DI fe<F> fn_add(const fe<F> &a, const fe<F> &b) {
  fe<F> t, u; unsigned c = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) t.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  uint32_t br = fn_sub_p<F>(u, t);
#pragma unroll
  for (int i = 0; i < F::N; i++) t.v[i] = br ? t.v[i] : u.v[i];
  return t;
}
template <class F> AVRF_DI fe<F> fn_sub(const fe<F> &a, const fe<F> &b) {
  fe<F> t; unsigned br = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) t.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
  const uint32_t m = 0u - br; unsigned c = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) t.v[i] = __builtin_addc(t.v[i], (unsigned)(F::P[i] & m), c, &c);
  return t;
}
template <class F> AVRF_DI fe<F> fn_neg(const fe<F> &a) {
  fe<F> t; unsigned br = 0; const bool z = fn_is_zero(a);
#pragma unroll
  for (int i = 0; i < F::N; i++) { const unsigned d = __builtin_subc((unsigned)F::P[i], a.v[i], br, &br); t.v[i] = z ? 0u : d; }
  return t;
}
template <class F> AVRF_DI fe<F> fn_dbl(const fe<F> &a) { return fn_add<F>(a, a); }

// Montgomery product a*b/R mod p, product scanning (mac96.h)
template <class F> AVRF_DI fe<F> fn_mul(const fe<F> &a, const fe<F> &b) {
  constexpr int N = F::N;
  fe<F> r, u;
  mont_mul_ps<N, F>(r.v, a.v, b.v);
  uint32_t br = fn_sub_p<F>(u, r);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}
// the operand-scanning (carry-free CIOS) form, kept as the cross-check in tools/ubench.hip
template <class F> AVRF_DI fe<F> fn_mul_cios(const fe<F> &a, const fe<F> &b) {
  constexpr int N = F::N;
  uint32_t t[N];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
    uint32_t m = (uint32_t)A * F::NINV;
    uint64_t C = (uint64_t)m * F::P[0] + (uint32_t)A;
    A >>= 32; C >>= 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      A += (uint64_t)a.v[j] * b.v[i] + t[j];
      C += (uint64_t)m * F::P[j] + (uint32_t)A;
      t[j - 1] = (uint32_t)C;
      A >>= 32; C >>= 32;
    }
    t[N - 1] = (uint32_t)(A + C);
  }
  fe<F> r, u;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = t[i];
  uint32_t br = fn_sub_p<F>(u, r);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}
template <class F> AVRF_DI fe<F> fn_sqr(const fe<F> &a) {
  constexpr int N = F::N;
  fe<F> r, u;
  mont_sqr_ps<N, F>(r.v, a.v);
  uint32_t br = fn_sub_p<F>(u, r);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}
template <class F> AVRF_DI fe<F> fn_to_mont(const fe<F> &a) { return fn_mul<F>(a, fn_const<F>(F::R2)); }
template <class F> AVRF_DI fe<F> fn_from_mont(const fe<F> &a) { fe<F> one = fn_zero<F::N>(); one.v[0] = 1; return fn_mul<F>(a, one); }

// a^-1 (0 -> 0): branch-free binary GCD, one fused halving step per iteration -- the N-limb form of fp256.h's fp_inv_nf (see the
// comment there; invariants x1 a = u, x2 a = v mod p, v odd).  ~1.4 x BITS iterations of ~15 N carry / select instructions: for
// the 381-bit field ~95 k instructions against ~570 twelve-limb Montgomery products (~330 k instructions, most of them
// multiply-adds) for the fixed power below -- the inversion was 1.4 ms of a lone wave in k_g1_lincomb and in the pairing kernel.
template <class F> __device__ __noinline__ static fe<F> fn_inv(fe<F> a) {
  constexpr int N = F::N;
  fe<F> u = a, v = fn_const<F>(F::P), x1 = fn_zero<N>(), x2 = fn_zero<N>();
  x1.v[0] = 1;
#pragma unroll 1
  while (!fn_is_zero(u)) {
    const bool odd = (u.v[0] & 1u) != 0;
    fe<F> d1, d2; unsigned b1 = 0, b2 = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { d1.v[i] = __builtin_subc(u.v[i], v.v[i], b1, &b1); d2.v[i] = __builtin_subc(v.v[i], u.v[i], b2, &b2); }
    const bool lt = b1 != 0, sw = odd && lt;
    fe<F> xa, xb;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t un = odd ? (lt ? d2.v[i] : d1.v[i]) : u.v[i];
      v.v[i] = sw ? u.v[i] : v.v[i];
      u.v[i] = un;
      xa.v[i] = sw ? x2.v[i] : x1.v[i];
      xb.v[i] = odd ? (sw ? x1.v[i] : x2.v[i]) : 0u;
    }
#pragma unroll
    for (int i = 0; i < N - 1; i++) u.v[i] = (u.v[i] >> 1) | (u.v[i + 1] << 31);
    u.v[N - 1] >>= 1;
#pragma unroll
    for (int i = 0; i < N; i++) x2.v[i] = sw ? x1.v[i] : x2.v[i];
    fe<F> t = fn_sub<F>(xa, xb);
    const uint32_t m = 0u - (t.v[0] & 1u); unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t.v[i] = __builtin_addc(t.v[i], (unsigned)(F::P[i] & m), c, &c);
#pragma unroll
    for (int i = 0; i < N - 1; i++) x1.v[i] = (t.v[i] >> 1) | (t.v[i + 1] << 31);
    x1.v[N - 1] = (t.v[N - 1] >> 1) | ((uint32_t)c << 31);
  }
  const fe<F> r2 = fn_const<F>(F::R2);                      // x2 = (a' R)^-1 for a = a' R; times R^3 / R gives a'^-1 R
  return fn_mul<F>(x2, fn_mul<F>(r2, r2));
}
// a^(p-2), square-and-multiply over the constant exponent (the cross-check of fn_inv)
template <class F> AVRF_DI fe<F> fn_inv_fermat(const fe<F> &a) {
  fe<F> r = fn_one<F>();
  bool started = false;
  for (int i = 32 * F::N - 1; i >= 0; i--) {
    if (started) r = fn_sqr<F>(r);
    if ((F::PM2[i >> 5] >> (i & 31)) & 1) { r = started ? fn_mul<F>(r, a) : a; started = true; }
  }
  return r;
}

// 16-byte vectorised global loads / stores (N is a multiple of 4)
template <int N> AVRF_DI fpn<N> fn_load(const uint32_t *s) {
  fpn<N> r; const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
#pragma unroll
  for (int i = 0; i < N / 4; i++) { uint4 q = s4[i]; r.v[4 * i] = q.x; r.v[4 * i + 1] = q.y; r.v[4 * i + 2] = q.z; r.v[4 * i + 3] = q.w; }
  return r;
}
template <int N> AVRF_DI void fn_store(uint32_t *d, const fpn<N> &a) {
  uint4 *d4 = reinterpret_cast<uint4 *>(d);
#pragma unroll
  for (int i = 0; i < N / 4; i++) d4[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}
template <int N> AVRF_DI fpn<N> fn_shfl_down(const fpn<N> &a, int delta) {
  fpn<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __shfl_down(a.v[i], delta);
  return r;
}

}  // namespace avrf
