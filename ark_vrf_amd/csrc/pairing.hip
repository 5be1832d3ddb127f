// pairing.hip -- batched pairing-product checks on gfx950: Miller loop + final exponentiation of BLS12-381 and BN254 for
//   e(A_i, Q_0) * e(B_i, Q_1) * ... == 1,   i < n   (G2 arguments fixed per setup: the KZG verifier key g2, tau g2)
//
// Device counterpart of arkworks `Pairing::multi_pairing` + `final_exponentiation` as reached from
// `RingVerifier::verify` (src/ring.rs:242) when many ring proofs are verified INDEPENDENTLY (one 2-pairing check
// each; SURVEY.md 8 a11, config C4 "4 096 independent verifies => 8 192 pairings").  The batch verifier
// (src/ring.rs:731) needs two pairings per batch and keeps using the host (host_pairing.h).
//
// Layout: an Fp12 element lives in SIXTEEN LANES: lane k < 12 holds the coefficient of w^k in the direct representation
//   Fp12 = Fp[w] / (w^12 - 2 xi0 w^6 + (xi0^2 + 1)),   w^6 = xi = xi0 + u,  u^2 = -1
// (the same model as the oracle, oracle/pairing_py.py; lanes 12..15 of a group idle).  A wave carries four checks.
// One lane could not hold an Fp12 (144 limbs for BLS12-381) next to the temporaries of its multiplication; spread over
// lanes, a multiplication is 12 wide (2N-limb) products per lane accumulated lazily into two 2N-limb sums (raw coefficients
// k and k + 12), two Montgomery reductions, and one exchange with lane k +- 6 for the reduction by the modulus of w; a
// squaring needs 8 product slots (symmetry), a multiplication by a line 5 (its five non-zero coefficients).
// Operands move with ds_bpermute inside the 16-lane group; no LDS allocation, no tower bookkeeping.
//
// Miller loop: the G2 arguments are fixed, so the line coefficients (slope lam and c = lam x_T - y_T per doubling /
// addition step, in Fp2) are tabulated once per setup -- by k_g2_lines below, ON THE DEVICE (affine G2 arithmetic over
// Fp2) -- and lane k evaluates its coefficient of the line at P = (xP, yP) as T0[k] + TX[k] xP + TY[k] yP.
// Final exponentiation: easy part with ONE Fp inversion (norm Fp12 -> Fp6 -> Fp2 -> Fp by Frobenius maps), hard part by the
// x-chain (BLS12: (x-1)^2 (x+p) (x^2+p^2-1) + 3) or plain square-and-multiply (BN254).  Only "== 1" is returned.
#include "../../include/avrf.h"
#include "fpn.h"
#include "host_pairing.h"
#include "msm.h"
#include "pairing.h"
#include <vector>

namespace avrf {

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw HipFailure{e_, __FILE__, __LINE__}; } while (0)

// ---------------------------------------------------------------- wide arithmetic

// full 2N-limb product, product scanning (mac96.h)
template <int N> AVRF_DI void mul_wide(uint32_t (&t)[2 * N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  uint64_t lo = 0; uint32_t ex = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); i++) mac96(lo, ex, a[i], b[k - i]);
    t[k] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)ex << 32); ex = 0;
  }
  t[2 * N - 1] = (uint32_t)lo;
}
// acc += pred ? w : 0   (2N limbs; the caller keeps acc < 2 p R, so no carry leaves the top limb)
template <int N> AVRF_DI void wide_add_if(uint32_t (&acc)[2 * N], const uint32_t (&w)[2 * N], bool pred) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 2 * N; i++) { c += (uint64_t)acc[i] + (pred ? w[i] : 0u); acc[i] = (uint32_t)c; c >>= 32; }
}
// top half -= p when it is >= p: brings a sum < 2 p R back below p R
template <class F> AVRF_DI void wide_fold(uint32_t (&acc)[2 * F::N]) {
  constexpr int N = F::N;
  fe<F> h, u;
#pragma unroll
  for (int i = 0; i < N; i++) h.v[i] = acc[N + i];
  const uint32_t br = fn_sub_p<F>(u, h);
#pragma unroll
  for (int i = 0; i < N; i++) acc[N + i] = br ? h.v[i] : u.v[i];
}
// Montgomery reduction of t < p R: t / R mod p
template <class F> AVRF_DI fe<F> redc(const uint32_t (&t)[2 * F::N]) {
  constexpr int N = F::N;
  uint32_t m[N]; fe<F> r, u;
  uint64_t lo = 0; uint32_t ex = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    { const uint64_t s = lo + t[k]; ex += s < lo ? 1u : 0u; lo = s; }
#pragma unroll
    for (int i = 0; i < k; i++) mac96_k(lo, ex, m[i], F::P[k - i]);
    m[k] = (uint32_t)lo * F::NINV;
    mac96_k(lo, ex, m[k], F::P[0]);
    lo = (lo >> 32) | ((uint64_t)ex << 32); ex = 0;
  }
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    { const uint64_t s = lo + t[k]; ex += s < lo ? 1u : 0u; lo = s; }
#pragma unroll
    for (int i = k - N + 1; i < N; i++) mac96_k(lo, ex, m[i], F::P[k - i]);
    r.v[k - N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)ex << 32); ex = 0;
  }
  const uint32_t br = fn_sub_p<F>(u, r);
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = br ? r.v[i] : u.v[i];
  return r;
}

// ---------------------------------------------------------------- Fp12 over sixteen lanes

template <int N> AVRF_DI fpn<N> grp_shfl(const fpn<N> &a, int src_k) {   // value of lane src_k of this lane's 16-lane group
  fpn<N> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __shfl(a.v[i], src_k, 16);
  return r;
}

// per-curve constants in device memory (Montgomery form), built by the host: see PairingConsts in pairing.h
template <class C> struct Cst {
  using F = typename C::Fq; static constexpr int N = F::N;
  const uint32_t *p;
  AVRF_DI fe<F> at(int idx) const { return fn_load<N>(p + (size_t)idx * N); }
};

// the reduction of the two raw sums (coefficients k and k + 12) shared by the three multiplication forms
// w^12 = 2 xi0 w^6 - kappa:  c_k = r_k - kappa r_(k+12) - 2 xi0 kappa r_(k+18)            (k < 6)
//                            c_k = r_k + 2 xi0 r_(k+6) + (4 xi0^2 - kappa) r_(k+12)       (k >= 6)
template <class C> AVRF_DI fpn<C::Fq::N> f12_finish(uint32_t (&lo_acc)[2 * C::Fq::N], uint32_t (&hi_acc)[2 * C::Fq::N], int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  const fe<F> L = redc<F>(lo_acc), H = redc<F>(hi_acc);
  const bool lowk = k < 6;
  const fe<F> Hp = grp_shfl<N>(H, lowk ? k + 6 : k - 6);
  Cst<C> K{cst};
  const fe<F> c1 = K.at(lowk ? PC_KAPPA : PC_4XI2_MINUS_KAPPA), c2 = K.at(lowk ? PC_2XI_KAPPA : PC_2XI);
  const fe<F> t1 = fn_mul<F>(H, c1), t2 = fn_mul<F>(Hp, c2);
  return lowk ? fn_sub<F>(fn_sub<F>(L, t1), t2) : fn_add<F>(fn_add<F>(L, t1), t2);
}
// product of two Fp12 elements; k = this lane's coefficient index (lanes 12..15 produce garbage that nobody reads)
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_mul(fpn<C::Fq::N> a, fpn<C::Fq::N> b, int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  uint32_t lo_acc[2 * N], hi_acc[2 * N], w[2 * N];
#pragma unroll
  for (int i = 0; i < 2 * N; i++) { lo_acc[i] = 0; hi_acc[i] = 0; }
#pragma unroll 1
  for (int i = 0; i < 12; i++) {
    const fpn<N> ai = grp_shfl<N>(a, i);
    int j = k - i; if (j < 0) j += 12;
    const fpn<N> bj = grp_shfl<N>(b, j);
    mul_wide<N>(w, ai.v, bj.v);
    const bool low = i <= k;                                   // raw coefficient k (i + j = k) or k + 12
    wide_add_if<N>(lo_acc, w, low);
    wide_add_if<N>(hi_acc, w, !low);
    if ((i & 3) == 3) { wide_fold<F>(lo_acc); wide_fold<F>(hi_acc); }
  }
  return f12_finish<C>(lo_acc, hi_acc, k, cst);
}
// acc *= 2 (acc < p R on entry), folded back below p R
template <class F> AVRF_DI void wide_double(uint32_t (&acc)[2 * F::N]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 2 * F::N; i++) { const uint32_t v = acc[i]; acc[i] = (v << 1) | c; c = v >> 31; }
  wide_fold<F>(acc);
}

// a^2 with the symmetry of the schoolbook square: raw coefficient m = 2 * sum_{i<j, i+j=m} a_i a_j + [m even] a_(m/2)^2.
// Lane k owns m = k and m = k + 12: at most 6 cross products (5 on even lanes) and, on even lanes, the two squares
// a_(k/2)^2 and a_(k/2+6)^2 -- 8 product slots instead of the 12 of the general multiplication.
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_sqr(fpn<C::Fq::N> a, int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  uint32_t lo_acc[2 * N], hi_acc[2 * N], w[2 * N];
#pragma unroll
  for (int i = 0; i < 2 * N; i++) { lo_acc[i] = 0; hi_acc[i] = 0; }
  const int kk = k < 12 ? k : 0;
  const int nlow = (kk + 1) >> 1, nhigh = (11 - kk) >> 1;        // cross products of m = k (i < k - i) and of m = k + 12
#pragma unroll 1
  for (int t = 0; t < 6; t++) {
    const bool low = t < nlow;
    const int u = t - nlow;
    const bool live = low || u < nhigh;
    const int i = low ? t : kk + 1 + u, j = low ? kk - t : 11 - u;
    const fpn<N> ai = grp_shfl<N>(a, live ? i : 0), aj = grp_shfl<N>(a, live ? j : 0);
    mul_wide<N>(w, ai.v, aj.v);
    wide_add_if<N>(lo_acc, w, live && low);
    wide_add_if<N>(hi_acc, w, live && !low);
    if (t == 3) { wide_fold<F>(lo_acc); wide_fold<F>(hi_acc); }
  }
  wide_fold<F>(lo_acc); wide_fold<F>(hi_acc);
  wide_double<F>(lo_acc); wide_double<F>(hi_acc);
  const bool even = (kk & 1) == 0;
  {
    const fpn<N> s0 = grp_shfl<N>(a, kk >> 1);
    mul_wide<N>(w, s0.v, s0.v);
    wide_add_if<N>(lo_acc, w, even);
    const fpn<N> s1 = grp_shfl<N>(a, even ? (kk >> 1) + 6 : 0);       // k even <= 10: (k + 12) / 2 <= 11
    mul_wide<N>(w, s1.v, s1.v);
    wide_add_if<N>(hi_acc, w, even);
    wide_fold<F>(lo_acc); wide_fold<F>(hi_acc);
  }
  return f12_finish<C>(lo_acc, hi_acc, k, cst);
}

// f * l for a line l whose only non-zero coefficients sit at the five positions of the twist type (M: 0, 2, 3, 6, 8;
// D: 0, 1, 3, 7, 9): five product slots per lane instead of twelve
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_mul_line(fpn<C::Fq::N> f, fpn<C::Fq::N> l, int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  uint32_t lo_acc[2 * N], hi_acc[2 * N], w[2 * N];
#pragma unroll
  for (int i = 0; i < 2 * N; i++) { lo_acc[i] = 0; hi_acc[i] = 0; }
  const int kk = k < 12 ? k : 0;
#pragma unroll 1
  for (int t = 0; t < 5; t++) {
    const int j = C::MTWIST ? (t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 3 : t == 3 ? 6 : 8) : (t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 3 : t == 3 ? 7 : 9);
    const bool low = j <= kk;                                    // i + j = k, else i + j = k + 12
    const int i = low ? kk - j : kk + 12 - j;
    const fpn<N> fi = grp_shfl<N>(f, i), lj = grp_shfl<N>(l, j);
    mul_wide<N>(w, fi.v, lj.v);
    wide_add_if<N>(lo_acc, w, low);
    wide_add_if<N>(hi_acc, w, !low);
    if (t == 3) { wide_fold<F>(lo_acc); wide_fold<F>(hi_acc); }
  }
  wide_fold<F>(lo_acc); wide_fold<F>(hi_acc);
  return f12_finish<C>(lo_acc, hi_acc, k, cst);
}

template <class C> AVRF_DI fpn<C::Fq::N> f12_one(int k) { return k == 0 ? fn_one<typename C::Fq>() : fn_zero<C::Fq::N>(); }
template <class C> AVRF_DI fpn<C::Fq::N> f12_conj(const fpn<C::Fq::N> &a, int k) { return (k & 1) ? fn_neg<typename C::Fq>(a) : a; }   // a^(p^6)
// a^(p^2), a^(p^4): the Fp2 coefficient of w^(k mod 6) is scaled by gamma^(k mod 6), gamma = xi^((p^2-1)/6) in Fp
template <class C> AVRF_DI fpn<C::Fq::N> f12_frob2(const fpn<C::Fq::N> &a, int k, const uint32_t *cst) {
  Cst<C> K{cst}; return fn_mul<typename C::Fq>(a, K.at(PC_FROB2 + (k < 12 ? k % 6 : 0)));
}
// a^p: A_k -> conj(A_k) * gamma1_k with A_k = (d_k + xi0 d_(k+6)) + d_(k+6) u
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_frob1(fpn<C::Fq::N> a, int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  Cst<C> K{cst};
  const bool lowk = k < 6; const int kk = k < 12 ? k % 6 : 0;
  const fe<F> other = grp_shfl<N>(a, lowk ? k + 6 : (k < 12 ? k - 6 : 0));
  const fe<F> d_lo = lowk ? a : other, d_hi = lowk ? other : a;
  const fe<F> x0 = fn_add<F>(d_lo, fn_mul<F>(d_hi, K.at(PC_XI0))), x1 = fn_neg<F>(d_hi);   // conj(A) = x0 + x1 u
  const fe<F> g0 = K.at(PC_FROB1 + 2 * kk), g1 = K.at(PC_FROB1 + 2 * kk + 1);
  const fe<F> y1 = fn_add<F>(fn_mul<F>(x0, g1), fn_mul<F>(x1, g0));                        // imaginary part of the product
  if (!lowk) return y1;                                                                   // d'_(k+6) = y1
  const fe<F> y0 = fn_sub<F>(fn_mul<F>(x0, g0), fn_mul<F>(x1, g1));
  return fn_sub<F>(y0, fn_mul<F>(y1, K.at(PC_XI0)));                                       // d'_k = y0 - xi0 y1
}
// 1 / a: n = a conj(a) in Fp6, N = n n^(p^2) n^(p^4) in Fp2, one Fp inversion of its norm
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_inv(fpn<C::Fq::N> a, int k, const uint32_t *cst) {
  using F = typename C::Fq; constexpr int N = F::N;
  Cst<C> K{cst};
  const fe<F> ac = f12_conj<C>(a, k);
  const fe<F> n = f12_mul<C>(a, ac, k, cst);
  const fe<F> n2 = f12_frob2<C>(n, k, cst), n4 = f12_frob2<C>(n2, k, cst);
  const fe<F> t = f12_mul<C>(n2, n4, k, cst);
  const fe<F> Nn = f12_mul<C>(n, t, k, cst);                     // only coefficients 0 and 6 are non-zero
  const fe<F> d0 = grp_shfl<N>(Nn, 0), d6 = grp_shfl<N>(Nn, 6);
  const fe<F> xa = fn_add<F>(d0, fn_mul<F>(d6, K.at(PC_XI0)));   // N = xa + d6 u
  const fe<F> ni = fn_inv<F>(fn_add<F>(fn_sqr<F>(xa), fn_sqr<F>(d6)));
  const fe<F> ia = fn_mul<F>(xa, ni), ib = fn_neg<F>(fn_mul<F>(d6, ni));                    // 1/N = ia + ib u
  fe<F> Ninv = fn_zero<N>();
  if (k == 0) Ninv = fn_sub<F>(ia, fn_mul<F>(ib, K.at(PC_XI0)));
  if (k == 6) Ninv = ib;
  const fe<F> ninv = f12_mul<C>(t, Ninv, k, cst);
  return f12_mul<C>(ac, ninv, k, cst);
}
// a^|x| by square-and-multiply, conjugated when x < 0 (valid in the cyclotomic subgroup, where 1/a = conj(a))
template <class C> __device__ __noinline__ static fpn<C::Fq::N> f12_pow_x(fpn<C::Fq::N> a, int k, const uint32_t *cst) {
  fpn<C::Fq::N> r = a;
  int top = 63; while (!((C::X_ABS >> top) & 1)) top--;
#pragma unroll 1
  for (int bit = top - 1; bit >= 0; bit--) { r = f12_sqr<C>(r, k, cst); if ((C::X_ABS >> bit) & 1) r = f12_mul<C>(r, a, k, cst); }
  return C::X_NEG ? f12_conj<C>(r, k) : r;
}

// ---------------------------------------------------------------- the check

// pts: n x np affine G1 points (Montgomery x | y, N words each; (0, 0) = infinity: that pair contributes 1);
// tab: np line tables of `steps` entries {T0[12], TX[12], TY[12]} (Fp each); ok[i] = 1 iff the product is one.
template <class C>
__global__ void __launch_bounds__(64, 2)       // 256 VGPRs: two waves per SIMD hide each other's ds_bpermute / scratch latencies
k_pairing_check(const uint32_t *__restrict__ pts, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ cst, uint32_t n, uint32_t np,
                uint32_t steps, int32_t *__restrict__ ok) {
  using F = typename C::Fq; constexpr int N = F::N;
  const uint32_t lane = threadIdx.x & 63, k = lane & 15, grp = lane >> 4;
  uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = item < n;
  if (!live) item = n - 1;                                       // whole groups only: keeps the wave uniform
  const uint32_t kc = k < 12 ? k : 0;                            // table column of this lane
  fe<F> f = f12_one<C>(k);
  uint32_t s = 0;
#pragma unroll 1
  for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) {
    f = f12_sqr<C>(f, k, cst);
    const int nadd = ((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1) ? 2 : 1;
#pragma unroll 1
    for (int rep = 0; rep < nadd; rep++, s++) {
#pragma unroll 1
      for (uint32_t q = 0; q < np; q++) {
        const uint32_t *pp = pts + ((size_t)item * np + q) * 2 * N;
        const fe<F> xp = fn_load<N>(pp), yp = fn_load<N>(pp + N);
        if (fn_is_zero(xp) && fn_is_zero(yp)) continue;          // point at infinity (uniform inside the group)
        const uint32_t *tp = tab + (((size_t)q * steps + s) * 36 + kc) * N;
        fe<F> l = fn_add<F>(fn_load<N>(tp), fn_add<F>(fn_mul<F>(fn_load<N>(tp + 12 * N), xp), fn_mul<F>(fn_load<N>(tp + 24 * N), yp)));
        if (k >= 12) l = fn_zero<N>();
        f = f12_mul_line<C>(f, l, k, cst);
      }
    }
  }
  // final exponentiation
  fe<F> t = f12_mul<C>(f12_conj<C>(f, k), f12_inv<C>(f, k, cst), k, cst);          // f^(p^6 - 1)
  t = f12_mul<C>(f12_frob2<C>(t, k, cst), t, k, cst);                              // ^(p^2 + 1)
  fe<F> out;
  if (C::X_CHAIN) {
    const fe<F> a = f12_mul<C>(f12_pow_x<C>(t, k, cst), f12_conj<C>(t, k), k, cst);                 // t^(x-1)
    const fe<F> b = f12_mul<C>(f12_pow_x<C>(a, k, cst), f12_conj<C>(a, k), k, cst);                 // ^(x-1)
    const fe<F> c = f12_mul<C>(f12_pow_x<C>(b, k, cst), f12_frob1<C>(b, k, cst), k, cst);            // ^(x+p)
    fe<F> d = f12_pow_x<C>(f12_pow_x<C>(c, k, cst), k, cst);
    d = f12_mul<C>(f12_mul<C>(d, f12_frob2<C>(c, k, cst), k, cst), f12_conj<C>(c, k), k, cst);       // ^(x^2+p^2-1)
    out = f12_mul<C>(d, f12_mul<C>(t, f12_sqr<C>(t, k, cst), k, cst), k, cst);                    // * t^3
  } else {
    out = t;
#pragma unroll 1
    for (int bit = C::HARD_EXP_BITS - 2; bit >= 0; bit--) {
      out = f12_sqr<C>(out, k, cst);
      if ((C::HARD_EXP[bit >> 6] >> (bit & 63)) & 1) out = f12_mul<C>(out, t, k, cst);
    }
  }
  const bool mine = k >= 12 || (k == 0 ? fn_eq(out, fn_one<F>()) : fn_is_zero(out));
  const uint64_t m = __ballot(mine);
  if (live && k == 0) ok[item] = (((m >> (16 * grp)) & 0xffffu) == 0xffffu) ? 1 : 0;
}

// ---------------------------------------------------------------- G2 on the device: line tables of fixed arguments

// Fp2 = Fp[u] / (u^2 + 1) in one lane
template <class F> struct f2d { fe<F> a, b; };
template <class F> AVRF_DI f2d<F> f2_add(const f2d<F> &x, const f2d<F> &y) { return {fn_add<F>(x.a, y.a), fn_add<F>(x.b, y.b)}; }
template <class F> AVRF_DI f2d<F> f2_sub(const f2d<F> &x, const f2d<F> &y) { return {fn_sub<F>(x.a, y.a), fn_sub<F>(x.b, y.b)}; }
template <class F> __device__ __noinline__ static f2d<F> f2_mul(f2d<F> x, f2d<F> y) {
  const fe<F> t0 = fn_mul<F>(x.a, y.a), t1 = fn_mul<F>(x.b, y.b), t2 = fn_mul<F>(fn_add<F>(x.a, x.b), fn_add<F>(y.a, y.b));
  return {fn_sub<F>(t0, t1), fn_sub<F>(fn_sub<F>(t2, t0), t1)};
}
template <class F> __device__ __noinline__ static f2d<F> f2_inv(f2d<F> x) {
  const fe<F> n = fn_inv<F>(fn_add<F>(fn_sqr<F>(x.a), fn_sqr<F>(x.b)));
  return {fn_mul<F>(x.a, n), fn_neg<F>(fn_mul<F>(x.b, n))};
}
template <class F> AVRF_DI bool f2_is_zero(const f2d<F> &x) { return fn_is_zero(x.a) && fn_is_zero(x.b); }

// One lane per G2 point Q (affine on the sextic twist, Montgomery x.a | x.b | y.a | y.b): walks the ate loop with affine
// doubling / addition over Fp2 and writes, per step, the line's three coefficient vectors in the direct basis:
//   M-twist (BLS12-381), line * w^3:  c at w^0, -lam xP at w^2, yP at w^3
//   D-twist (BN254):                  yP at w^0, -lam xP at w^1, c at w^3            with c = lam x_T - y_T in Fp2,
// an Fp2 value (a + b u) at w^j being (a - xi0 b) w^j + b w^(j+6).  flag |= 1 if a denominator vanishes (Q not of order r).
template <class C>
__global__ void __launch_bounds__(64)
k_g2_lines(const uint32_t *__restrict__ q_in, uint32_t nq, uint32_t steps, const uint32_t *__restrict__ cst, uint32_t *__restrict__ tab,
           uint32_t *__restrict__ flag) {
  using F = typename C::Fq; constexpr int N = F::N;
  const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= nq) return;
  Cst<C> K{cst};
  const fe<F> xi0 = K.at(PC_XI0), zero = fn_zero<N>();
  const uint32_t *src = q_in + (size_t)qi * 4 * N;
  const f2d<F> qx = {fn_load<N>(src), fn_load<N>(src + N)}, qy = {fn_load<N>(src + 2 * N), fn_load<N>(src + 3 * N)};
  f2d<F> tx = qx, ty = qy;
  uint32_t s = 0;
  auto emit = [&](const f2d<F> &lam, const f2d<F> &c) {
    uint32_t *o = tab + ((size_t)qi * steps + s) * 36 * N;
    for (int i = 0; i < 36; i++) fn_store<N>(o + (size_t)i * N, zero);
    const fe<F> c_lo = fn_sub<F>(c.a, fn_mul<F>(c.b, xi0)), m_lo = fn_neg<F>(fn_sub<F>(lam.a, fn_mul<F>(lam.b, xi0))), m_hi = fn_neg<F>(lam.b);
    const fe<F> one = fn_one<F>();
    if (C::MTWIST) {
      fn_store<N>(o + 0 * N, c_lo); fn_store<N>(o + 6 * N, c.b);                          // T0: c at w^0
      fn_store<N>(o + (12 + 2) * N, m_lo); fn_store<N>(o + (12 + 8) * N, m_hi);           // TX: -lam at w^2
      fn_store<N>(o + (24 + 3) * N, one);                                                 // TY: 1 at w^3
    } else {
      fn_store<N>(o + (24 + 0) * N, one);                                                 // TY: 1 at w^0
      fn_store<N>(o + (12 + 1) * N, m_lo); fn_store<N>(o + (12 + 7) * N, m_hi);           // TX: -lam at w^1
      fn_store<N>(o + 3 * N, c_lo); fn_store<N>(o + 9 * N, c.b);                          // T0: c at w^3
    }
    s++;
  };
#pragma unroll 1
  for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) {
    {                                                            // doubling step: lam = 3 x^2 / (2 y)
      if (f2_is_zero<F>(ty)) atomicOr(flag, 1u);
      const f2d<F> x2 = f2_mul<F>(tx, tx);
      const f2d<F> num = f2_add<F>(f2_add<F>(x2, x2), x2);
      const f2d<F> lam = f2_mul<F>(num, f2_inv<F>(f2_add<F>(ty, ty)));
      emit(lam, f2_sub<F>(f2_mul<F>(lam, tx), ty));
      const f2d<F> nx = f2_sub<F>(f2_sub<F>(f2_mul<F>(lam, lam), tx), tx);
      ty = f2_sub<F>(f2_mul<F>(lam, f2_sub<F>(tx, nx)), ty); tx = nx;
    }
    if ((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1) {              // addition step: lam = (yQ - yT) / (xQ - xT)
      const f2d<F> dx = f2_sub<F>(qx, tx);
      if (f2_is_zero<F>(dx)) atomicOr(flag, 1u);
      const f2d<F> lam = f2_mul<F>(f2_sub<F>(qy, ty), f2_inv<F>(dx));
      emit(lam, f2_sub<F>(f2_mul<F>(lam, tx), ty));
      const f2d<F> nx = f2_sub<F>(f2_sub<F>(f2_mul<F>(lam, lam), tx), qx);
      ty = f2_sub<F>(f2_mul<F>(lam, f2_sub<F>(tx, nx)), ty); tx = nx;
    }
  }
}

// ---------------------------------------------------------------- host side

template <class C> static int ate_steps() {
  int s = 0;
  for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) s += ((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1) ? 2 : 1;
  return s;
}

template <class C> static void build_impl(PairingTables &pt, const uint8_t *g2_raw, size_t nq, hipStream_t stream) {
  using HP = HostPairing<C>; using Fp = typename HP::Fp; using El = typename HP::El; constexpr int N = C::Fq::N;
  // constants (Montgomery form)
  std::vector<El> cs(PC_COUNT, Fp::zero());
  const int xi0 = C::XI0, kappa = xi0 * xi0 + 1;
  cs[PC_XI0] = HP::small(xi0); cs[PC_KAPPA] = HP::small(kappa); cs[PC_2XI] = HP::small(2 * xi0);
  cs[PC_2XI_KAPPA] = HP::small(2 * xi0 * kappa); cs[PC_4XI2_MINUS_KAPPA] = HP::small(4 * xi0 * xi0 - kappa);
  { El g = Fp::from32(C::FROB2_GAMMA), p = Fp::one(); for (int k = 0; k < 6; k++) { cs[PC_FROB2 + k] = p; p = Fp::mul(p, g); } }
  cs[PC_FROB1] = Fp::one(); cs[PC_FROB1 + 1] = Fp::zero();
  for (int k = 1; k < 6; k++) { cs[PC_FROB1 + 2 * k] = Fp::from32(C::FROB1_GAMMA[k - 1][0]); cs[PC_FROB1 + 2 * k + 1] = Fp::from32(C::FROB1_GAMMA[k - 1][1]); }
  std::vector<uint32_t> csw((size_t)PC_COUNT * N);
  for (int i = 0; i < PC_COUNT; i++) memcpy(&csw[(size_t)i * N], cs[i].l, 4 * N);
  // G2 arguments (Montgomery x.a | x.b | y.a | y.b)
  const size_t g2len = C::Fq::N * 4 * 4;
  std::vector<uint32_t> qw(nq * 4 * N);
  for (size_t i = 0; i < nq; i++) {
    typename HP::G2 q; HP::g2_decode(g2_raw + i * g2len, &q);
    if (q.inf) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};
    memcpy(&qw[(i * 4 + 0) * N], q.x.a.l, 4 * N); memcpy(&qw[(i * 4 + 1) * N], q.x.b.l, 4 * N);
    memcpy(&qw[(i * 4 + 2) * N], q.y.a.l, 4 * N); memcpy(&qw[(i * 4 + 3) * N], q.y.b.l, 4 * N);
  }
  pt.steps = (uint32_t)ate_steps<C>(); pt.nq = (uint32_t)nq; pt.words = N;
  uint32_t *d_q = nullptr, *d_flag = nullptr;
  HIP_CHECK(hipMalloc(&pt.d_cst, csw.size() * 4)); HIP_CHECK(hipMalloc(&pt.d_tab, (size_t)nq * pt.steps * 36 * N * 4));
  HIP_CHECK(hipMalloc(&d_q, qw.size() * 4)); HIP_CHECK(hipMalloc(&d_flag, 4));
  HIP_CHECK(hipMemcpyAsync(pt.d_cst, csw.data(), csw.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipMemcpyAsync(d_q, qw.data(), qw.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, stream));
  hipLaunchKernelGGL(k_g2_lines<C>, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, stream, (const uint32_t *)d_q, (uint32_t)nq, pt.steps,
                     (const uint32_t *)pt.d_cst, pt.d_tab, d_flag);
  uint32_t flag = 0;
  HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream)); HIP_CHECK(hipGetLastError());
  (void)hipFree(d_q); (void)hipFree(d_flag);
  if (flag) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};
}

void PairingTables::build(int curve, const uint8_t *g2_raw, size_t nq, hipStream_t stream) {
  release();
  if (curve == 0) build_impl<G1Bls12381>(*this, g2_raw, nq, stream); else build_impl<G1Bn254>(*this, g2_raw, nq, stream);
  this->curve = curve;
}
void PairingTables::release() {
  if (d_cst) (void)hipFree(d_cst);
  if (d_tab) (void)hipFree(d_tab);
  d_cst = d_tab = nullptr; steps = nq = 0;
}

void launch_pairing_check(const PairingTables &pt, const uint32_t *d_pts, size_t n, int32_t *d_ok, hipStream_t stream) {
  if (!n) return;
  const dim3 grid((unsigned)((n * 16 + 63) / 64)), block(64);
  if (pt.curve == 0) hipLaunchKernelGGL(k_pairing_check<G1Bls12381>, grid, block, 0, stream, d_pts, (const uint32_t *)pt.d_tab, (const uint32_t *)pt.d_cst,
                                        (uint32_t)n, pt.nq, pt.steps, d_ok);
  else hipLaunchKernelGGL(k_pairing_check<G1Bn254>, grid, block, 0, stream, d_pts, (const uint32_t *)pt.d_tab, (const uint32_t *)pt.d_cst, (uint32_t)n,
                          pt.nq, pt.steps, d_ok);
}

}  // namespace avrf
