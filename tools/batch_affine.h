// tools/batch_affine.h (probe only, not part of libavrf) -- affine + affine additions of short-Weierstrass G1 points with a shared field inversion (Montgomery's
// trick), the "batch-affine" bucket accumulation of fixed-base KZG MSMs (the commitments of the ring prover,
// w3f-ring-proof reached from src/ring.rs:404,416; arkworks computes them with `VariableBaseMSM`, mixed additions into
// projective buckets).
//
// One round turns 2 m points into m: output p = in[2p] + in[2p + 1].  A lane owns K pairs (p = wave base + j * 64 + lane);
// pass A walks them forwards collecting the prefix products of d_j = x_b - x_a (prefixes go to a global scratch), the lane
// inverts its own total once (binary extended Euclid: additions and shifts only, which gfx950 issues at 4.5x the rate of
// the 32x32 multiply-add), pass B walks backwards: 1/d_j = running * prefix_j, lambda = (y_b - y_a) / d_j,
// x3 = lambda^2 - x_a - x_b, y3 = lambda (x_a - x3) - y_a.  5M + 1S per addition plus 1/K of the inversion, against 8M + 2S
// for the XYZZ mixed addition.
// (0, 0) is the point at infinity; a pair with an infinite operand passes the other one through.  A pair with x_a = x_b
// (doubling or P + (-P): never for distinct table entries with honest scalars, but the kernel must not be wrong when it
// happens) sets *flag and the caller recomputes the launch with the projective path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fpn.h"
#include "curves.h"

namespace avrf {

// a^-1 for a in Montgomery form (result in Montgomery form); 0 -> 0.  Binary extended Euclid, 2 BITS + 2 fixed iterations,
// branch-free: x1 a = u R^-2 ..., kept as  x1 * abar = u * R^2,  x2 * abar = v * R^2  (mod p) with abar = a R, so that v = 1
// leaves x2 = R^2 / abar = a^-1 R.
template <class F> AVRF_DI fe<F> fn_inv_gcd(const fe<F> &a) {
  constexpr int N = F::N;
  uint32_t u[N], v[N], x1[N], x2[N];
#pragma unroll
  for (int i = 0; i < N; i++) { u[i] = a.v[i]; v[i] = F::P[i]; x1[i] = F::R2[i]; x2[i] = 0; }
#pragma unroll 1
  for (int it = 0; it < 2 * F::BITS + 2; it++) {
    const bool uo = u[0] & 1, vo = v[0] & 1;
    uint32_t d[N]; int64_t c = 0;                                           // d = u - v, borrow = u < v
#pragma unroll
    for (int i = 0; i < N; i++) { c += (int64_t)u[i] - (int64_t)v[i]; d[i] = (uint32_t)c; c >>= 32; }
    const bool lt = c != 0, both = uo && vo;
    const bool sw = (uo && !vo) || (both && lt);                             // the one to halve goes to u
    const uint32_t ms = sw ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint32_t t = (u[i] ^ v[i]) & ms; u[i] ^= t; v[i] ^= t;
      t = (x1[i] ^ x2[i]) & ms; x1[i] ^= t; x2[i] ^= t;
    }
    if (both) {                                                              // u -= v (u >= v now), x1 -= x2 mod p
      int64_t b = 0;
#pragma unroll
      for (int i = 0; i < N; i++) { b += (int64_t)u[i] - (int64_t)v[i]; u[i] = (uint32_t)b; b >>= 32; }
      int64_t e = 0;
#pragma unroll
      for (int i = 0; i < N; i++) { e += (int64_t)x1[i] - (int64_t)x2[i]; x1[i] = (uint32_t)e; e >>= 32; }
      const uint32_t mp = e ? 0xffffffffu : 0u;
      uint64_t g = 0;
#pragma unroll
      for (int i = 0; i < N; i++) { g += (uint64_t)x1[i] + (F::P[i] & mp); x1[i] = (uint32_t)g; g >>= 32; }
    }
    (void)d;
    {                                                                        // u >>= 1; x1 = x1 / 2 mod p
      const uint32_t mo = (x1[0] & 1) ? 0xffffffffu : 0u;
      uint64_t g = 0;
#pragma unroll
      for (int i = 0; i < N; i++) { g += (uint64_t)x1[i] + (F::P[i] & mo); x1[i] = (uint32_t)g; g >>= 32; }
      const uint32_t top = (uint32_t)g;
#pragma unroll
      for (int i = 0; i < N; i++) {
        u[i] = (u[i] >> 1) | (i + 1 < N ? u[i + 1] << 31 : 0u);
        x1[i] = (x1[i] >> 1) | ((i + 1 < N ? x1[i + 1] : top) << 31);
      }
    }
  }
  fe<F> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = x2[i];
  return r;
}

// entry e of the sorted array: table point (e & 0x7fffffff), negated when the top bit is set; 0xffffffff = padding (infinity).
// idx == nullptr callers pass the point's position instead.
template <class C> AVRF_DI typename G1Curve<C>::base_t ba_load_entry(const uint32_t *pts, uint32_t e, bool &inf) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  typename CV::base_t q;
  if (e == 0xffffffffu) { q.x = fn_zero<N>(); q.y = fn_zero<N>(); inf = true; return q; }
  q = CV::load_base(pts + (size_t)(e & 0x7fffffffu) * 2 * N);
  inf = fn_is_zero(q.x) && fn_is_zero(q.y);
  if ((e & 0x80000000u) && !inf) q.y = fn_neg<Fq>(q.y);
  return q;
}

template <class C, bool INDEXED>
__global__ void __launch_bounds__(256)
k_ba_round(const uint32_t *__restrict__ pts, const uint32_t *__restrict__ idx, uint32_t npairs, uint32_t K, uint32_t *__restrict__ dst,
           uint32_t *__restrict__ scratch, uint32_t *__restrict__ flag) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N; using el = fpn<N>;
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t base = (size_t)wave * 64 * K;
  if (base >= npairs) return;
  uint32_t *sc = scratch + (base + lane) * N;                                // prefix j of this lane: sc + j * 64 * N
  el acc = fn_one<Fq>();
  uint32_t live = 0;                                                         // pairs of this lane
#pragma unroll 1
  for (uint32_t j = 0; j < K; j++) {
    const size_t p = base + (size_t)j * 64 + lane;
    if (p >= npairs) break;
    live = j + 1;
    bool ia, ib;
    const typename CV::base_t A = ba_load_entry<C>(pts, INDEXED ? idx[2 * p] : (uint32_t)(2 * p), ia);
    const typename CV::base_t B = ba_load_entry<C>(pts, INDEXED ? idx[2 * p + 1] : (uint32_t)(2 * p + 1), ib);
    el d = fn_sub<Fq>(B.x, A.x);
    if (ia || ib) d = fn_one<Fq>();
    else if (fn_is_zero(d)) { atomicOr(flag, 1u); d = fn_one<Fq>(); }
    fn_store<N>(sc + (size_t)j * 64 * N, acc);
    acc = fn_mul<Fq>(acc, d);
  }
  el run = fn_inv_gcd<Fq>(acc);
#pragma unroll 1
  for (uint32_t j = live; j-- > 0;) {
    const size_t p = base + (size_t)j * 64 + lane;
    bool ia, ib;
    const typename CV::base_t A = ba_load_entry<C>(pts, INDEXED ? idx[2 * p] : (uint32_t)(2 * p), ia);
    const typename CV::base_t B = ba_load_entry<C>(pts, INDEXED ? idx[2 * p + 1] : (uint32_t)(2 * p + 1), ib);
    uint32_t *o = dst + p * 2 * N;
    if (ia || ib) { fn_store<N>(o, ia ? B.x : A.x); fn_store<N>(o + N, ia ? B.y : A.y); continue; }
    const el d = fn_sub<Fq>(B.x, A.x);
    if (fn_is_zero(d)) { fn_store<N>(o, A.x); fn_store<N>(o + N, A.y); continue; }          // flagged in pass A
    const el inv = fn_mul<Fq>(run, fn_load<N>(sc + (size_t)j * 64 * N));
    run = fn_mul<Fq>(run, d);
    const el lam = fn_mul<Fq>(fn_sub<Fq>(B.y, A.y), inv);
    const el x3 = fn_sub<Fq>(fn_sub<Fq>(fn_sqr<Fq>(lam), A.x), B.x);
    fn_store<N>(o, x3);
    fn_store<N>(o + N, fn_sub<Fq>(fn_mul<Fq>(lam, fn_sub<Fq>(A.x, x3)), A.y));
  }
}

// dst[p] = in[2p] + in[2p + 1], p < npairs; idx == nullptr: `pts` is the dense input itself.  scratch: npairs * N words
// (rounded up to whole waves of 64 K pairs).
template <class C>
static void ba_launch_round(const uint32_t *pts, const uint32_t *idx, uint32_t npairs, uint32_t K, uint32_t *dst, uint32_t *scratch, uint32_t *flag,
                            hipStream_t stream) {
  if (!npairs) return;
  const uint32_t waves = (uint32_t)(((size_t)npairs + (size_t)64 * K - 1) / ((size_t)64 * K));
  const dim3 grid((waves + 3) / 4), block(256);
  if (idx) hipLaunchKernelGGL((k_ba_round<C, true>), grid, block, 0, stream, pts, idx, npairs, K, dst, scratch, flag);
  else hipLaunchKernelGGL((k_ba_round<C, false>), grid, block, 0, stream, pts, idx, npairs, K, dst, scratch, flag);
}

}  // namespace avrf
