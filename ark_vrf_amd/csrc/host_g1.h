// host_g1.h -- host-side finishing arithmetic for G1 (short Weierstrass, a = 0): generic-width
// Montgomery field (64-bit limbs) and XYZZ point ops, used for the O(256)-step window Horner of a
// KZG MSM and for normalising its result.  Product code (not the oracle).
#pragma once
#include <stdint.h>
#include <string.h>
#include "consts_gen.h"

namespace avrf {

template <class F> struct HostFieldN {
  static constexpr int L = F::N / 2;                       // 64-bit limbs
  struct El { uint64_t l[L]; };
  static El from32(const uint32_t (&c)[F::N]) { El r; for (int i = 0; i < L; i++) r.l[i] = (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 32); return r; }
  static El P() { return from32(F::P); }
  static El one() { return from32(F::ONE); }
  static El zero() { El r; memset(&r, 0, sizeof r); return r; }
  static uint64_t ninv64() { uint64_t p0 = P().l[0], inv = 1; for (int i = 0; i < 7; i++) inv *= 2 - p0 * inv; return (uint64_t)0 - inv; }
  static bool is_zero(const El &a) { uint64_t o = 0; for (int i = 0; i < L; i++) o |= a.l[i]; return o == 0; }
  static bool eq(const El &a, const El &b) { uint64_t o = 0; for (int i = 0; i < L; i++) o |= a.l[i] ^ b.l[i]; return o == 0; }
  static uint64_t addc(El &o, const El &a, const El &b) {
    unsigned __int128 c = 0;
    for (int i = 0; i < L; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
  }
  static uint64_t subb(El &o, const El &a, const El &b) {
    uint64_t br = 0;
    for (int i = 0; i < L; i++) { unsigned __int128 t = (unsigned __int128)a.l[i] - b.l[i] - br; o.l[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
    return br;
  }
  static El add(const El &a, const El &b) { El t, u; uint64_t c = addc(t, a, b); uint64_t br = subb(u, t, P()); return (c || !br) ? u : t; }
  static El sub(const El &a, const El &b) { El t; if (subb(t, a, b)) addc(t, t, P()); return t; }
  static El neg(const El &a) { if (is_zero(a)) return a; El t; subb(t, P(), a); return t; }
  static El dbl(const El &a) { return add(a, a); }
  static constexpr uint64_t p_limb(int i) { return (uint64_t)F::P[2 * i] | ((uint64_t)F::P[2 * i + 1] << 32); }
  static constexpr uint64_t ninv_c() { uint64_t p0 = p_limb(0), inv = 1; for (int i = 0; i < 7; i++) inv *= 2 - p0 * inv; return (uint64_t)0 - inv; }
  // Montgomery product, operand scanning with the reduction interleaved (CIOS); constants folded at compile time
  static El mul(const El &a, const El &b) {
    constexpr uint64_t ninv = ninv_c();
    uint64_t t[L + 2];
#pragma GCC unroll 16
    for (int i = 0; i < L + 2; i++) t[i] = 0;
#pragma GCC unroll 16
    for (int i = 0; i < L; i++) {
      unsigned __int128 c = 0;
#pragma GCC unroll 16
      for (int j = 0; j < L; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[L]; t[L] = (uint64_t)c; t[L + 1] = (uint64_t)(c >> 64);
      const uint64_t q = t[0] * ninv;
      c = (unsigned __int128)q * p_limb(0) + t[0]; c >>= 64;
#pragma GCC unroll 16
      for (int j = 1; j < L; j++) { c += (unsigned __int128)q * p_limb(j) + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[L]; t[L - 1] = (uint64_t)c; t[L] = t[L + 1] + (uint64_t)(c >> 64);
    }
    El r, u; for (int i = 0; i < L; i++) r.l[i] = t[i];
    uint64_t br = subb(u, r, P());
    return (t[L] || !br) ? u : r;
  }
  static El sqr(const El &a) { return mul(a, a); }
  static El from_mont(const El &a) { El o = zero(); o.l[0] = 1; return mul(a, o); }
  static El to_mont(const El &a) { return mul(a, from32(F::R2)); }
  static El inv_fermat(const El &a) {                      // a^(p-2); kept as the cross-check of inv()
    El e = from32(F::PM2), r = one();
    for (int i = 64 * L - 1; i >= 0; i--) { r = sqr(r); if ((e.l[i / 64] >> (i % 64)) & 1) r = mul(r, a); }
    return r;
  }
  // halve modulo p (p odd): x/2 if even, (x + p)/2 otherwise
  static void half_mod(El &x, const El &p) {
    uint64_t carry = 0;
    if (x.l[0] & 1) carry = addc(x, x, p);
    for (int i = 0; i < L - 1; i++) x.l[i] = (x.l[i] >> 1) | (x.l[i + 1] << 63);
    x.l[L - 1] = (x.l[L - 1] >> 1) | (carry << 63);
  }
  static void shr1(El &x) { for (int i = 0; i < L - 1; i++) x.l[i] = (x.l[i] >> 1) | (x.l[i + 1] << 63); x.l[L - 1] >>= 1; }
  static bool geq(const El &a, const El &b) { for (int i = L - 1; i >= 0; i--) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; return true; }
  // Montgomery inverse by the binary extended Euclid (about 2 * bits shift/subtract steps; ~5x faster than a^(p-2)).
  // Not constant time: host-side, public data only (verifier, proof normalisation).
  static El inv(const El &a_mont) {
    const El p = P();
    El u = from_mont(a_mont), v = p, x1 = zero(), x2 = zero();
    if (is_zero(u)) return u;
    x1.l[0] = 1;
    El onep = zero(); onep.l[0] = 1;
    while (!eq(u, onep) && !eq(v, onep)) {
      while (!(u.l[0] & 1)) { shr1(u); half_mod(x1, p); }
      while (!(v.l[0] & 1)) { shr1(v); half_mod(x2, p); }
      if (geq(u, v)) { subb(u, u, v); if (subb(x1, x1, x2)) addc(x1, x1, p); }
      else { subb(v, v, u); if (subb(x2, x2, x1)) addc(x2, x2, p); }
    }
    const El r = eq(u, onep) ? x1 : x2;                       // plain inverse of the plain value
    return mul(r, from32(F::R2));                              // back to Montgomery form
  }
};

// XYZZ points on y^2 = x^3 + b; identity <=> zz = 0.  Same limb layout as the device accumulators.
template <class C> struct HostG1 {
  using Fq = HostFieldN<typename C::Fq>;
  using El = typename Fq::El;
  struct Pt { El x, y, zz, zzz; };
  static Pt identity() { Pt r; r.x = Fq::one(); r.y = Fq::one(); r.zz = Fq::zero(); r.zzz = Fq::zero(); return r; }
  static bool is_identity(const Pt &a) { return Fq::is_zero(a.zz); }
  static Pt from_raw32(const uint32_t *w) { Pt r; memcpy(&r, w, sizeof r); return r; }
  static Pt dbl(const Pt &a) {
    if (is_identity(a)) return a;
    El U = Fq::dbl(a.y), V = Fq::sqr(U), W = Fq::mul(U, V), S = Fq::mul(a.x, V);
    El X2 = Fq::sqr(a.x), M = Fq::add(Fq::dbl(X2), X2);
    Pt r;
    r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
    r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, a.y));
    r.zz = Fq::mul(V, a.zz); r.zzz = Fq::mul(W, a.zzz);
    return r;
  }
  static Pt add(const Pt &a, const Pt &b) {
    if (is_identity(a)) return b;
    if (is_identity(b)) return a;
    El U1 = Fq::mul(a.x, b.zz), U2 = Fq::mul(b.x, a.zz), S1 = Fq::mul(a.y, b.zzz), S2 = Fq::mul(b.y, a.zzz);
    El P = Fq::sub(U2, U1), R = Fq::sub(S2, S1);
    if (Fq::is_zero(P)) return Fq::is_zero(R) ? dbl(a) : identity();
    El PP = Fq::sqr(P), PPP = Fq::mul(P, PP), Q = Fq::mul(U1, PP);
    Pt r;
    r.x = Fq::sub(Fq::sub(Fq::sqr(R), PPP), Fq::dbl(Q));
    r.y = Fq::sub(Fq::mul(R, Fq::sub(Q, r.x)), Fq::mul(S1, PPP));
    r.zz = Fq::mul(Fq::mul(a.zz, b.zz), PP); r.zzz = Fq::mul(Fq::mul(a.zzz, b.zzz), PPP);
    return r;
  }
  // canonical affine x || y, little-endian, 8*L bytes each; all zero for the identity
  static void to_affine_bytes(const Pt &a, uint8_t *out) {
    constexpr int B = 8 * Fq::L;
    if (is_identity(a)) { memset(out, 0, 2 * B); return; }
    El zi = Fq::inv(a.zz), zzzi = Fq::inv(a.zzz);
    El x = Fq::from_mont(Fq::mul(a.x, zi)), y = Fq::from_mont(Fq::mul(a.y, zzzi));
    memcpy(out, x.l, B); memcpy(out + B, y.l, B);
  }
  // same for many points with one field inversion (Montgomery's trick over all zz and zzz)
  static void to_affine_bytes_batch(const Pt *pts, size_t n, uint8_t *out) {
    constexpr int B = 8 * Fq::L;
    if (n == 1) { to_affine_bytes(pts[0], out); return; }
    El *pre = new El[2 * n + 1];
    El run = Fq::one();
    for (size_t i = 0; i < n; i++) {
      const bool id = is_identity(pts[i]);
      pre[2 * i] = run; if (!id) run = Fq::mul(run, pts[i].zz);
      pre[2 * i + 1] = run; if (!id) run = Fq::mul(run, pts[i].zzz);
    }
    El inv = Fq::inv(run);
    for (size_t i = n; i-- > 0;) {
      uint8_t *o = out + 2 * B * i;
      if (is_identity(pts[i])) { memset(o, 0, 2 * B); continue; }
      El zzzi = Fq::mul(inv, pre[2 * i + 1]); inv = Fq::mul(inv, pts[i].zzz);
      El zzi = Fq::mul(inv, pre[2 * i]); inv = Fq::mul(inv, pts[i].zz);
      El x = Fq::from_mont(Fq::mul(pts[i].x, zzi)), y = Fq::from_mont(Fq::mul(pts[i].y, zzzi));
      memcpy(o, x.l, B); memcpy(o + B, y.l, B);
    }
    delete[] pre;
  }
};

}  // namespace avrf
