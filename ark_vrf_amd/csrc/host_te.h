// host_te.h -- host-side (CPU) finishing arithmetic of the product: 4x64-limb Montgomery fields
// and twisted-Edwards point ops, used for the O(256) tail of an MSM (window Horner), for
// combining per-GPU partial points, and for normalising results.  This is product code (it is
// NOT the oracle and shares no source with oracle/); the heavy lifting is in the HIP kernels.
#pragma once
#include <stdint.h>
#include <string.h>
#include "consts_gen.h"

namespace avrf {

struct H256 { uint64_t l[4]; };

template <class F> struct HostField {
  static H256 from32(const uint32_t (&c)[8]) {
    H256 r; for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 32); return r;
  }
  static H256 P() { return from32(F::P); }
  static H256 one() { return from32(F::ONE); }
  static H256 r2() { return from32(F::R2); }
  static uint64_t ninv64() {  // -p^-1 mod 2^64 from p
    uint64_t p0 = P().l[0], inv = 1;
    for (int i = 0; i < 7; i++) inv *= 2 - p0 * inv;
    return (uint64_t)0 - inv;
  }
  static bool is_zero(const H256 &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
  static bool eq(const H256 &a, const H256 &b) { return !((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])); }
  static uint64_t addc(H256 &o, const H256 &a, const H256 &b) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
  }
  static uint64_t subb(H256 &o, const H256 &a, const H256 &b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)a.l[i] - b.l[i] - br; o.l[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
    return br;
  }
  static bool geq_p(const H256 &a) { H256 t; return subb(t, a, P()) == 0; }
  static H256 add(const H256 &a, const H256 &b) {
    H256 t, u; uint64_t c = addc(t, a, b); uint64_t br = subb(u, t, P()); return (c || !br) ? u : t;
  }
  static H256 sub(const H256 &a, const H256 &b) { H256 t; if (subb(t, a, b)) addc(t, t, P()); return t; }
  static H256 neg(const H256 &a) { if (is_zero(a)) return a; H256 t; subb(t, P(), a); return t; }
  static H256 mul(const H256 &a, const H256 &b) {
    static const uint64_t ninv = ninv64();
    static const H256 p = P();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
      uint64_t q = t[0] * ninv;
      c = (unsigned __int128)q * p.l[0] + t[0]; c >>= 64;
      for (int j = 1; j < 4; j++) { c += (unsigned __int128)q * p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    H256 r = {{t[0], t[1], t[2], t[3]}}, u;
    uint64_t br = subb(u, r, p);
    return (t[4] || !br) ? u : r;
  }
  static H256 sqr(const H256 &a) { return mul(a, a); }
  static H256 to_mont(const H256 &a) { return mul(a, r2()); }
  static H256 from_mont(const H256 &a) { H256 o = {{1, 0, 0, 0}}; return mul(a, o); }
  static H256 inv(const H256 &a) {
    H256 e = from32(F::PM2), r = one();
    for (int i = 255; i >= 0; i--) { r = sqr(r); if ((e.l[i / 64] >> (i % 64)) & 1) r = mul(r, a); }
    return r;
  }
  static H256 load_le(const uint8_t *b) { H256 r; memcpy(r.l, b, 32); return r; }
  static void store_le(uint8_t *b, const H256 &a) { memcpy(b, a.l, 32); }
};

struct HostExt { H256 x, y, t, z; };

template <class S> struct HostTe {
  using Fq = HostField<typename S::Fq>;
  static H256 mul_a(const H256 &v) {
    if (S::A_KIND == 1) { H256 t = Fq::add(v, v); t = Fq::add(t, t); t = Fq::add(t, v); return Fq::neg(t); }
    if (S::A_KIND == 2) return Fq::neg(v);
    return v;
  }
  // S::SW_NATIVE (secp256r1): the same struct holds XYZZ coordinates (X, Y, ZZ = t, ZZZ = z), identity ZZ = 0 -- te.h
  static HostExt identity() {
    HostExt r; memset(&r, 0, sizeof r); r.y = Fq::one();
    if constexpr (S::SW_NATIVE) r.x = Fq::one(); else r.z = Fq::one();
    return r;
  }
  static bool is_identity(const HostExt &p) {
    if constexpr (S::SW_NATIVE) return Fq::is_zero(p.t);
    return Fq::is_zero(p.x) && Fq::eq(p.y, p.z);
  }
  static HostExt sw_dbl(const HostExt &a) {            // dbl-2008-s-1, a = -3
    H256 U = Fq::add(a.y, a.y), V = Fq::sqr(U), W = Fq::mul(U, V), Sx = Fq::mul(a.x, V);
    H256 M = Fq::mul(Fq::sub(a.x, a.t), Fq::add(a.x, a.t)); M = Fq::add(Fq::add(M, M), M);
    HostExt r;
    r.x = Fq::sub(Fq::sqr(M), Fq::add(Sx, Sx));
    r.y = Fq::sub(Fq::mul(M, Fq::sub(Sx, r.x)), Fq::mul(W, a.y));
    r.t = Fq::mul(V, a.t); r.z = Fq::mul(W, a.z);
    return r;
  }
  static HostExt sw_add(const HostExt &a, const HostExt &b) {   // add-2008-s with its exceptional cases
    if (Fq::is_zero(a.t)) return b;
    if (Fq::is_zero(b.t)) return a;
    H256 U1 = Fq::mul(a.x, b.t), P = Fq::sub(Fq::mul(b.x, a.t), U1);
    H256 S1 = Fq::mul(a.y, b.z), R = Fq::sub(Fq::mul(b.y, a.z), S1);
    if (Fq::is_zero(P)) return Fq::is_zero(R) ? sw_dbl(a) : identity();
    H256 PP = Fq::sqr(P), Q = Fq::mul(U1, PP), PPP = Fq::mul(P, PP), T = Fq::mul(S1, PPP);
    HostExt r;
    r.t = Fq::mul(Fq::mul(a.t, b.t), PP); r.z = Fq::mul(Fq::mul(a.z, b.z), PPP);
    r.x = Fq::sub(Fq::sub(Fq::sqr(R), PPP), Fq::add(Q, Q));
    r.y = Fq::sub(Fq::mul(R, Fq::sub(Q, r.x)), T);
    return r;
  }
  static HostExt add(const HostExt &p, const HostExt &q) {
    if constexpr (S::SW_NATIVE) return sw_add(p, q);
    static const H256 d = Fq::from32(S::D);
    H256 A = Fq::mul(p.x, q.x), B = Fq::mul(p.y, q.y), C = Fq::mul(Fq::mul(p.t, q.t), d), D = Fq::mul(p.z, q.z);
    H256 E = Fq::sub(Fq::sub(Fq::mul(Fq::add(p.x, p.y), Fq::add(q.x, q.y)), A), B);
    H256 F = Fq::sub(D, C), G = Fq::add(D, C), H = Fq::sub(B, mul_a(A));
    HostExt r; r.x = Fq::mul(E, F); r.y = Fq::mul(G, H); r.t = Fq::mul(E, H); r.z = Fq::mul(F, G); return r;
  }
  static HostExt dbl(const HostExt &p) {
    if constexpr (S::SW_NATIVE) return sw_dbl(p);
    H256 A = Fq::sqr(p.x), B = Fq::sqr(p.y), C = Fq::sqr(p.z); C = Fq::add(C, C);
    H256 D = mul_a(A), E = Fq::sub(Fq::sub(Fq::sqr(Fq::add(p.x, p.y)), A), B);
    H256 G = Fq::add(D, B), F = Fq::sub(G, C), H = Fq::sub(D, B);
    HostExt r; r.x = Fq::mul(E, F); r.y = Fq::mul(G, H); r.t = Fq::mul(E, H); r.z = Fq::mul(F, G); return r;
  }
  // canonical affine bytes x||y (LE32 each)
  static void to_affine_bytes(const HostExt &p, uint8_t out[64]) {
    if constexpr (S::SW_NATIVE) {                       // the identity: all-zero bytes
      H256 i = Fq::inv(Fq::mul(p.t, p.z));
      H256 x = Fq::from_mont(Fq::mul(p.x, Fq::mul(i, p.z))), y = Fq::from_mont(Fq::mul(p.y, Fq::mul(i, p.t)));
      Fq::store_le(out, x); Fq::store_le(out + 32, y); return;
    }
    H256 zi = Fq::inv(p.z);
    H256 x = Fq::from_mont(Fq::mul(p.x, zi)), y = Fq::from_mont(Fq::mul(p.y, zi));
    Fq::store_le(out, x); Fq::store_le(out + 32, y);
  }
  static bool from_affine_bytes(const uint8_t in[64], HostExt *o) {
    H256 x = Fq::load_le(in), y = Fq::load_le(in + 32);
    if (Fq::geq_p(x) || Fq::geq_p(y)) return false;
    if constexpr (S::SW_NATIVE) {
      if (Fq::is_zero(x) && Fq::is_zero(y)) { *o = identity(); return true; }
      o->x = Fq::to_mont(x); o->y = Fq::to_mont(y); o->t = Fq::one(); o->z = Fq::one(); return true;
    }
    o->x = Fq::to_mont(x); o->y = Fq::to_mont(y); o->t = Fq::mul(o->x, o->y); o->z = Fq::one();
    return true;
  }
  static HostExt from_raw32(const uint32_t *w) {  // device te_ext_raw -> host
    HostExt r; memcpy(&r, w, 128); return r;      // identical little-endian limb layout
  }
};

}  // namespace avrf
