// te_quad.h -- twisted-Edwards addition spread over FOUR lanes (one coordinate per lane) for the latency-bound tails of
// the MSM (bucket reduction, src/thin.rs:319 / src/pedersen.rs:420 via msm.hip).
//
// One SIMD retires about one lane-per-point te_add every 9 us however few of its lanes carry live work, so a reduction tree
// is paced by the number of SEQUENTIAL point additions.  With lane j of a quad holding coordinate j of an extended point
// (0 X, 1 Y, 2 T, 3 Z) the ten field multiplications of add-2008-hwcd collapse into three rounds that all four lanes
// execute together:
//   round 1   lane j: c1_j * c2_j                      -> A = X1 X2, B = Y1 Y2, TT = T1 T2, D = Z1 Z2
//   round 2   lanes 0,1: (X1+Y1)(X2+Y2); lane 2: d TT   -> E', C
//   linear    lane 0: E = E' - (A+B); lane 1: H = B - aA; lane 2: G = D + C; lane 3: F = D - C
//   round 3   lane 0: E F; lane 1: G H; lane 2: E H; lane 3: F G      -> X3, Y3, T3, Z3
// Operands move between the lanes of a quad with DPP quad_perm moves (no LDS).  A wave then carries 16 points and a
// point addition costs ~1/3 of the instructions of the one-lane form.
#pragma once
#include "te.h"
#include "curves.h"

namespace avrf {

template <int P0, int P1, int P2, int P3> AVRF_DI fp qperm(const fp &a) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
  return r;
}
AVRF_DI fp fp_sel(bool c, const fp &a, const fp &b) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// coordinate j of the identity (0, 1, 0, 1)
template <class S> AVRF_DI fp q_identity(uint32_t j) { return (j & 1) ? fp_one<typename S::Fq>() : fp_zero(); }

// coordinate j of P1 + P2 given coordinate j of each (all four lanes of the quad must be active)
template <class S> __device__ __noinline__ static fp q_add(fp a, fp b, uint32_t j) {
  using Fq = typename S::Fq;
  const fp m1 = fp_mul<Fq>(a, b);
  const fp sa = fp_add<Fq>(a, qperm<1, 0, 3, 2>(a)), sb = fp_add<Fq>(b, qperm<1, 0, 3, 2>(b));
  const bool l2 = j == 2;
  const fp m2 = fp_mul<Fq>(fp_sel(l2, m1, sa), fp_sel(l2, fp_const<Fq>(S::D), sb));
  const fp A = qperm<0, 0, 0, 0>(m1), B = qperm<1, 1, 1, 1>(m1), D = qperm<3, 3, 3, 3>(m1);
  const fp C = qperm<2, 2, 2, 2>(m2), Ep = qperm<0, 0, 0, 0>(m2);
  const fp AB = fp_add<Fq>(A, B), aA = mul_a<S>(A), nC = fp_neg<Fq>(C);
  const fp X = fp_sel(j == 0, Ep, fp_sel(j == 1, B, D));
  const fp Y = fp_sel(j == 0, AB, fp_sel(j == 1, aA, fp_sel(l2, nC, C)));
  const fp U = fp_sub<Fq>(X, Y);                                   // lane 0 E, 1 H, 2 G, 3 F
  return fp_mul<Fq>(qperm<0, 1, 0, 3>(U), qperm<3, 2, 1, 2>(U));
}

// coordinate j of 2 P given coordinate j of P (dbl-2008-hwcd): two rounds instead of the addition's three --
//   round 1   lane 0: X^2, lane 1: Y^2, lane 2: (X + Y)^2, lane 3: Z^2        (T is not an input of a doubling)
//   linear    lane 0: E = (X+Y)^2 - A - B; lane 1: H = aA - B; lane 2: G = aA + B; lane 3: F = G - 2 Z^2
//   round 2   lane 0: E F; lane 1: G H; lane 2: E H; lane 3: F G
template <class S> __device__ __noinline__ static fp q_dbl(fp a, uint32_t j) {
  using Fq = typename S::Fq;
  const fp sa = fp_add<Fq>(a, qperm<1, 0, 3, 2>(a));
  const fp m1 = fp_sqr<Fq>(fp_sel(j == 2, qperm<0, 1, 0, 3>(sa), a));
  const fp A = qperm<0, 0, 0, 0>(m1), B = qperm<1, 1, 1, 1>(m1), Sq = qperm<2, 2, 2, 2>(m1), ZZ = qperm<3, 3, 3, 3>(m1);
  const fp aA = mul_a<S>(A);
  const fp X = fp_sel(j == 0, Sq, fp_sel(j == 1, aA, fp_add<Fq>(aA, B)));
  const fp Y = fp_sel(j == 0, fp_add<Fq>(A, B), fp_sel(j == 1, B, fp_sel(j == 2, fp_zero(), fp_add<Fq>(ZZ, ZZ))));
  const fp U = fp_sub<Fq>(X, Y);                                   // lane 0 E, 1 H, 2 G, 3 F
  return fp_mul<Fq>(qperm<0, 1, 0, 3>(U), qperm<3, 2, 1, 2>(U));
}

// coordinate j of P1 + Q for an AFFINE Q given as te_pre {x2, y2, k2 = d x2 y2} (every lane sees all three): Z2 = 1 leaves the
// fourth lane of the addition's first round free for (X1 + Y1)(x2 + y2), and k2 makes C = T1 k2 a product of that round too, so
// a mixed addition is TWO rounds:
//   round 1   lane 0: X1 x2 = A; lane 1: Y1 y2 = B; lane 2: T1 k2 = C; lane 3: (X1 + Y1)(x2 + y2) = E'
//   linear    lane 0: E = E' - A - B; lane 1: H = B - aA; lane 2: G = Z1 + C; lane 3: F = Z1 - C
//   round 2   lane 0: E F; lane 1: G H; lane 2: E H; lane 3: F G
template <class S> __device__ __noinline__ static fp q_madd(fp a, fp x2, fp y2, fp k2, uint32_t j) {
  using Fq = typename S::Fq;
  const fp sa = fp_add<Fq>(a, qperm<1, 0, 3, 2>(a));               // lanes 0, 1: X1 + Y1
  const fp opa = fp_sel(j == 3, qperm<0, 1, 2, 0>(sa), a);
  const fp opb = fp_sel(j == 0, x2, fp_sel(j == 1, y2, fp_sel(j == 2, k2, fp_add<Fq>(x2, y2))));
  const fp m1 = fp_mul<Fq>(opa, opb);
  const fp A = qperm<0, 0, 0, 0>(m1), B = qperm<1, 1, 1, 1>(m1), C = qperm<2, 2, 2, 2>(m1), Ep = qperm<3, 3, 3, 3>(m1), Z1 = qperm<3, 3, 3, 3>(a);
  const fp X = fp_sel(j == 0, Ep, fp_sel(j == 1, B, Z1));
  const fp Y = fp_sel(j == 0, fp_add<Fq>(A, B), fp_sel(j == 1, mul_a<S>(A), fp_sel(j == 2, fp_neg<Fq>(C), C)));
  const fp U = fp_sub<Fq>(X, Y);                                   // lane 0 E, 1 H, 2 G, 3 F
  return fp_mul<Fq>(qperm<0, 1, 0, 3>(U), qperm<3, 2, 1, 2>(U));
}

// coordinate j of k P from coordinate j of P: fixed 3-bit windows over a per-lane table {0, P, .., 7 P} (registers; the entry is
// picked with a select chain, so the quads of a wave may hold different scalars), NBITS <= 253 bits of the plain integer k.
// The window digits are read from the top of a left-aligned copy of k that moves up three bits per step.
template <class S, int NBITS> __device__ static fp q_smul(const fp &base, const fp &k, uint32_t j) {
  constexpr int NW = (NBITS + 2) / 3, W = (3 * NW + 31) / 32, SH = 32 * W - 3 * NW;     // 128 bits: 43 windows in 5 words; 253: 85 in 8
  static_assert(W <= 8 && SH < 32, "scalar window layout");
  fp tab[8];
  tab[0] = q_identity<S>(j); tab[1] = base; tab[2] = q_dbl<S>(base, j);
#pragma unroll 1
  for (int i = 3; i < 8; i++) tab[i] = q_add<S>(tab[i - 1], base, j);
  uint32_t kk[W];
#pragma unroll
  for (int i = W - 1; i >= 0; i--) {
    const uint32_t hi = i < 8 ? k.v[i] : 0u, lo = i >= 1 ? k.v[i - 1] : 0u;
    kk[i] = SH ? (hi << SH) | (lo >> (32 - SH)) : hi;
  }
  fp acc = q_identity<S>(j);
#pragma unroll 1
  for (int w = 0; w < NW; w++) {
    const uint32_t d = kk[W - 1] >> 29;
#pragma unroll
    for (int i = W - 1; i >= 1; i--) kk[i] = (kk[i] << 3) | (kk[i - 1] >> 29);
    kk[0] <<= 3;
    if (w) { acc = q_dbl<S>(acc, j); acc = q_dbl<S>(acc, j); acc = q_dbl<S>(acc, j); }
    fp e = tab[0];
#pragma unroll
    for (uint32_t i = 1; i < 8; i++) e = fp_sel(d == i, tab[i], e);
    acc = q_add<S>(acc, e, j);
  }
  return acc;
}

AVRF_DI fp fp_shfl_down(const fp &a, int delta) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], delta);
  return r;
}
AVRF_DI fp fp_shfl_xor(const fp &a, int mask) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_xor(a.v[i], mask);
  return r;
}

// ---- the same addition on the unsaturated limbs of fpu.h (the MSM's reduction tails: k_wsum_q1 / _q2, q_heavy_sum).  Lane j holds
// coordinate j as 9 signed limbs in the tables' Montgomery domain; the three rounds are the asm blocks of fpu_asm_gen.h
// (206 instructions against ~250 of the saturated multiplier) and the sums between them are limb-wise with three carry passes
// instead of six modular additions: ~860 instructions per lane against ~1 100.  A product of two running values comes out times
// 2^-SH, d is sliced with the shift, so every coordinate of a result carries the same factor: the same point (fpu_te.h).  Bounds:
// tools/fpu_model.py TEChain.add_gen (operands normalised to limbs <= 2^W + 4, closing operands carried, B + 5A unsigned).
template <int P0, int P1, int P2, int P3, int L> AVRF_DI fu<L> fu_qperm(const fu<L> &a) {
  fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
  return r;
}
template <int L> AVRF_DI fu<L> fu_sel(bool c, const fu<L> &a, const fu<L> &b) {
  fu<L> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
template <class S> __device__ __noinline__ static fuF<typename S::Fq> qu_add(fuF<typename S::Fq> a, fuF<typename S::Fq> b, uint32_t j) {
  using Fq = typename S::Fq;
  constexpr int L = UL<Fq>::L, SH = UL<Fq>::SH;
  const fu<L> m1 = fu_mul<Fq>(a, b);                                // X1 X2, Y1 Y2, T1 T2, Z1 Z2
  const fu<L> sa = fu_add<L>(a, fu_qperm<1, 0, 3, 2>(a)), sb = fu_carry<Fq>(fu_add<L>(b, fu_qperm<1, 0, 3, 2>(b)));
  const bool l2 = j == 2;
  const fu<L> m2 = fu_mul<Fq>(fu_sel<L>(l2, m1, sa), fu_sel<L>(l2, fu_slice<Fq, SH>(S::D), sb));   // lanes 0, 1: (X1 + Y1)(X2 + Y2); lane 2: d T1 T2
  const fu<L> A = fu_qperm<0, 0, 0, 0>(m1), B = fu_qperm<1, 1, 1, 1>(m1), D = fu_qperm<3, 3, 3, 3>(m1);
  const fu<L> C = fu_qperm<2, 2, 2, 2>(m2), Ep = fu_qperm<0, 0, 0, 0>(m2);
  // lane 0: E = E' - A - B; lane 1: H = B - a A; lane 2: G = D + C; lane 3: F = D - C -- carried to limbs in [0, 2^W + 4)
  fu<L> X, Y;
#pragma unroll
  for (int i = 0; i < L; i++) {
    X.v[i] = j == 0 ? Ep.v[i] : (j == 1 ? B.v[i] : D.v[i]);
    const int32_t y1 = S::A_KIND == 0 ? A.v[i] : -A.v[i];           // a = 1: B - A; a = -1 (and the first A of a = -5): B + A
    Y.v[i] = j == 0 ? A.v[i] + B.v[i] : (j == 1 ? y1 : (l2 ? -C.v[i] : C.v[i]));
  }
  fu<L> U = fu_carry<Fq>(fu_sub<L>(X, Y));
  if (S::A_KIND == 1) {                                              // a = -5: B + 5 A needs 32 unsigned bits (fpu_te.h teu_h)
    uint32_t h[L - 1];
#pragma unroll
    for (int i = 0; i < L - 1; i++) h[i] = (uint32_t)B.v[i] + 5u * (uint32_t)A.v[i];
    U = fu_sel<L>(j == 1, fu_carry_u<Fq>(h, B.v[L - 1] + 5 * A.v[L - 1]), U);
  }
  return fu_mul<Fq>(fu_qperm<0, 1, 0, 3>(U), fu_qperm<3, 2, 1, 2>(U));   // E F, H G, E H, F G
}

// what the reduction kernels below are written over: one coordinate of a point per lane, saturated (fp) or unsaturated (fu<9>)
template <class S> struct QuadSat {
  using el = fp;
  static AVRF_DI el identity(uint32_t j) { return q_identity<S>(j); }
  static AVRF_DI el add(const el &a, const el &b, uint32_t j) { return q_add<S>(a, b, j); }
  static AVRF_DI el sel(bool c, const el &a, const el &b) { return fp_sel(c, a, b); }
  static AVRF_DI el shfl_down(const el &a, int d) { return fp_shfl_down(a, d); }
  static AVRF_DI el shfl_xor(const el &a, int m) { return fp_shfl_xor(a, m); }
  static AVRF_DI el load(const uint32_t *p) { return load_fp(p); }                 // 8 canonical words
  static AVRF_DI void store(uint32_t *p, const el &v) { store_fp(p, v); }
  static AVRF_DI el load_part(const uint32_t *part, size_t k, int j) {             // coordinate j of partial sum k (the k_accumulate policy's format)
    if constexpr (TeCurve<S>::accum::PART_WORDS == 32) return load_fp(part + k * 32 + j * 8);
    else return teu_load_part_coord<S>(part + k * TEU_PART_WORDS, j);
  }
};
template <class S> struct QuadUns {
  using Fq = typename S::Fq;
  static constexpr int L = UL<Fq>::L;
  using el = fu<L>;
  static AVRF_DI el identity(uint32_t j) { return (j & 1) ? fu_const<Fq>(UL<Fq>::ONE) : fu_zero<L>(); }
  static AVRF_DI el add(const el &a, const el &b, uint32_t j) { return qu_add<S>(a, b, j); }
  static AVRF_DI el sel(bool c, const el &a, const el &b) { return fu_sel<L>(c, a, b); }
  static AVRF_DI el shfl_down(const el &a, int d) { el r;
#pragma unroll
    for (int i = 0; i < L; i++) r.v[i] = __shfl_down(a.v[i], d);
    return r; }
  static AVRF_DI el shfl_xor(const el &a, int m) { el r;
#pragma unroll
    for (int i = 0; i < L; i++) r.v[i] = __shfl_xor(a.v[i], m);
    return r; }
  static AVRF_DI el load(const uint32_t *p) { const fp v = load_fp(p); return fu_slice<Fq, 0>(v.v); }
  static AVRF_DI void store(uint32_t *p, const el &v) { fp r; fu_to_packed<Fq>(r.v, v); store_fp(p, r); }
  static AVRF_DI el load_part(const uint32_t *part, size_t k, int j) { return teu_load_part_coord_raw<S>(part + k * TEU_PART_WORDS, j); }
};
#if !defined(AVRF_NO_UNSAT) && !defined(AVRF_NO_UNSAT_QUAD)
template <class S> using Quad = std::conditional_t<FuAsm<typename S::Fq>::value && !S::SW_NATIVE, QuadUns<S>, QuadSat<S>>;
#else
template <class S> using Quad = QuadSat<S>;
#endif

// sum over the 16 quads of a wave, valid in quad 0
template <class S> AVRF_DI typename Quad<S>::el q_wave_sum(typename Quad<S>::el v, uint32_t q, uint32_t j) {
  using Q = Quad<S>;
#pragma unroll 1
  for (int off = 8; off >= 1; off >>= 1) {
    const typename Q::el o = Q::shfl_down(v, 4 * off);
    if ((int)q < off) v = Q::add(v, o, j);
  }
  return v;
}
// two sums at once: returns sum_q a_q in quad 0 and sum_q b_q in quad 8 (first step folds a into the low quads and b into
// the high quads, then both halves reduce together)
template <class S> AVRF_DI typename Quad<S>::el q_wave_sum2(const typename Quad<S>::el &a, const typename Quad<S>::el &b, uint32_t q, uint32_t j) {
  using Q = Quad<S>;
  const bool lowq = q < 8;
  const typename Q::el send = Q::sel(lowq, b, a);                     // what the partner quad (q ^ 8) accumulates
  const typename Q::el got = Q::shfl_xor(send, 32);
  typename Q::el v = Q::add(Q::sel(lowq, a, b), got, j);
#pragma unroll 1
  for (int off = 4; off >= 1; off >>= 1) {
    const typename Q::el o = Q::shfl_down(v, 4 * off);
    if ((int)(q & 7) < off) v = Q::add(v, o, j);
  }
  return v;
}
// suffix sums over the quads of a wave: result_q = sum_{q' >= q} v_q'
template <class S> AVRF_DI typename Quad<S>::el q_wave_suffix(typename Quad<S>::el v, uint32_t q, uint32_t j) {
  using Q = Quad<S>;
#pragma unroll 1
  for (int off = 1; off < 16; off <<= 1) {
    const typename Q::el o = Q::shfl_down(v, 4 * off);
    if ((int)q + off < 16) v = Q::add(v, o, j);
  }
  return v;
}

// Level 1 of sum_b b * B_b for one window of nb = wpw * 16 * m buckets (b = 1 .. nb, B[b-1] as te_ext in `buckets`):
// wave wi of the window, quad q owns buckets b0 = wi * 16 m + q + 16 k (k < m), weight b0 + 1 = wi * 16 m + (q + 1) + 16 k:
//   S_q = sum_k B, J_q = sum_k k B  (2m - 1 quad additions)
//   out[3 gw + 0] = V1 = sum_q (q + 1) S_q,  out[3 gw + 1] = V2 = sum_q J_q,  out[3 gw + 2] = Y = sum_q S_q
// so that the window's sum is  sum_wi V1 + 16 sum_wi V2 + 16 m sum_wi wi Y  -- no doublings on the device.
template <class S>
__global__ void __launch_bounds__(256)
k_wsum_q1(const uint32_t *__restrict__ buckets, uint32_t nb, uint32_t m, uint32_t wpw, uint32_t nwaves, uint32_t *__restrict__ out) {
  const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
  if (gw >= nwaves) return;
  const uint32_t v = gw / wpw, wi = gw - v * wpw;
  using Q = Quad<S>; using el = typename Q::el;
  const uint32_t *B = buckets + (size_t)v * nb * 32;
  el Sx = Q::identity(j), J = Q::identity(j);
#pragma unroll 1
  for (int k = (int)m - 1; k >= 0; k--) {
    const uint32_t b0 = wi * 16 * m + q + 16 * (uint32_t)k;
    if (b0 < nb) Sx = Q::add(Sx, Q::load(B + (size_t)b0 * 32 + j * 8), j);
    if (k > 0) J = Q::add(J, Sx, j);
  }
  const el A = q_wave_suffix<S>(Sx, q, j);                         // A_0 = Y; sum_q (q + 1) S_q = sum_q A_q
  const el V = q_wave_sum2<S>(A, J, q, j);                         // quad 0: V1, quad 8: V2
  uint32_t *o = out + (size_t)gw * 3 * 32 + j * 8;
  if (q == 0) { Q::store(o, V); Q::store(o + 64, A); }
  if (q == 8) Q::store(o + 32, V);
}

// Level 2: one wave per window over the wpw <= 16 triples of level 1:
//   out[3 v + 0] = P1 = sum V1, out[3 v + 1] = P2 = sum V2, out[3 v + 2] = P3 = sum_wi wi Y_wi
// and the host finishes  W_v = P1 + 2^4 P2 + 16 m P3  inside its window Horner.
template <class S>
__global__ void __launch_bounds__(64)
k_wsum_q2(const uint32_t *__restrict__ trip, uint32_t wpw, uint32_t *__restrict__ out) {
  const uint32_t v = blockIdx.x, lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
  using Q = Quad<S>; using el = typename Q::el;
  el v1 = Q::identity(j), v2 = v1, y = v1;
  if (q < wpw) {
    const uint32_t *t = trip + ((size_t)v * wpw + q) * 3 * 32 + j * 8;
    v1 = Q::load(t); v2 = Q::load(t + 32); y = Q::load(t + 64);
  }
  const el P12 = q_wave_sum2<S>(v1, v2, q, j);
  el A = q_wave_suffix<S>(y, q, j);                                // sum_q q Y_q = sum_{q >= 1} A_q
  if (q == 0) A = Q::identity(j);
  const el P3 = q_wave_sum<S>(A, q, j);
  uint32_t *o = out + (size_t)v * 3 * 32 + j * 8;
  if (q == 0) { Q::store(o, P12); Q::store(o + 64, P3); }
  if (q == 8) Q::store(o + 32, P12);
}

// Sum of a heavy bucket's np partials by one workgroup of 64 quads (see k_bucket_sum): quads stride over the partials,
// each wave folds its 16 quads, the four wave results meet in LDS.
template <class S> AVRF_DI void q_heavy_sum(const uint32_t *__restrict__ part, size_t p0, uint32_t np, uint32_t *__restrict__ dst, uint32_t *lds) {
  const uint32_t t = threadIdx.x, lane = t & 63, q = lane >> 2, j = lane & 3, gq = t >> 2, wv = t >> 6;
  using Q = Quad<S>; using el = typename Q::el;
  el a = Q::identity(j);
#pragma unroll 1
  for (uint32_t k = gq; k < np; k += 64) a = Q::add(a, Q::load_part(part, p0 + k, (int)j), j);
  a = q_wave_sum<S>(a, q, j);
  if (q == 0) Q::store(lds + wv * 32 + j * 8, a);
  __syncthreads();
  if (t < 4) {
#pragma unroll 1
    for (uint32_t w = 1; w < 4; w++) a = Q::add(a, Q::load(lds + w * 32 + j * 8), j);
    Q::store(dst + j * 8, a);
  }
  __syncthreads();
}

}  // namespace avrf
