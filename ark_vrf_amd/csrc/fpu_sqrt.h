// fpu_sqrt.h -- sqrt(u / v) as fp256.h's fp_sqrt_ratio_nf computes it (x = u w g^(-e/2), w = (u v)^((t-1)/2), the discrete log e of
// (u v)^t read in 8-bit windows), on the unsaturated limbs of fpu.h: the ~380 field operations of one point decompression are
// squarings and products by table entries with nothing between them, i.e. exactly the asm blocks of fpu_asm_gen.h (178 / 206
// vector instructions against ~250 of the saturated multiplier, which also squares by multiplying).  The fixed exponentiation
// reads its exponent in 4-bit windows (223 squarings + 56 products + 14 for the table, the binary form took 223 + ~111).
// Same results, bit for bit: the candidates' square roots are unique up to sign and the caller fixes the sign.
//
// Domains: everything here is x R' (R' = 2^(W L) = R 2^SH).  A saturated Montgomery value x R enters as its limbs times 2^SH
// (fu_slice<F, SH>); fu_mul(x R', y R') = x y R'; a value leaves through fu_mul(x R', R) = x R (lazily reduced, |value| < 1.1 p)
// and fu_to_packed (canonical).  Operand bounds of fu_mul / fu_sqr: every operand is a slice (limbs < 2^W) or a product's result
// (limbs 0 .. L-2 < 2^W, small signed top limb), values below 2 p: closed, tools/fpu_model.py check 1.
#pragma once
#include "fpu.h"

namespace avrf {

template <class F> AVRF_DI constexpr int fu_top_digit4(const uint32_t (&e)[8]) {
  for (int i = 63; i >= 0; i--) if ((e[i >> 3] >> (4 * (i & 7))) & 15u) return i;
  return 0;
}

template <class F> AVRF_DN bool fu_sqrt_ratio_nf(fp u, fp v, fp *out) {
  using U = UL<F>;
  constexpr int S = F::TWO_ADICITY, SH = U::SH, W = F::SQRT_W;
  const fuF<F> to_r = fu_const<F>(U::ONE);                            // R mod p, sliced plainly: fu_mul(x R', to_r) = x R
  const fuF<F> uu = fu_slice<F, SH>(u.v), vv = fu_slice<F, SH>(v.v);
  const fuF<F> a = fu_mul<F>(uu, vv);
  // w = a^((t-1)/2): 4-bit windows over the powers a^1 .. a^15 (private memory; the digits are the same in every lane)
  fuF<F> tab[16];
  tab[0] = to_r; tab[1] = a;
#pragma unroll 1
  for (int k = 2; k < 16; k++) tab[k] = fu_mul<F>(tab[k - 1], a);
  constexpr int TOP = fu_top_digit4<F>(F::T_MINUS1_HALF);
  fuF<F> w = tab[(F::T_MINUS1_HALF[TOP >> 3] >> (4 * (TOP & 7))) & 15u];
#pragma unroll 1
  for (int i = TOP - 1; i >= 0; i--) {
    w = fu_sqr<F>(fu_sqr<F>(fu_sqr<F>(fu_sqr<F>(w))));
    const uint32_t dgt = (F::T_MINUS1_HALF[i >> 3] >> (4 * (i & 7))) & 15u;
    if (dgt) w = fu_mul<F>(w, tab[dgt]);
  }
  fuF<F> c = fu_mul<F>(a, fu_sqr<F>(w));                              // a^t, in the group of 2^s-th roots of unity
  fuF<F> r = fu_mul<F>(uu, w);
  bool odd = false;
#pragma unroll 1
  for (int i = 0; i < F::SQRT_STEPS; i++) {
    const int wd = S - W * i < W ? S - W * i : W;                     // bits of this window
    fuF<F> d = c;
#pragma unroll 1
    for (int k = 0; k < S - W * i - wd; k++) d = fu_sqr<F>(d);
    fp dc; fu_to_packed<F>(dc.v, fu_mul<F>(d, to_r));                 // canonical, in the tables' domain
    const uint32_t j = sqrt_window<F>(dc.v[0], wd);                   // (fp256.h: perfect hash of the 2^w-th roots of unity)
    if (i == 0 && (j & 1u)) odd = true;
    fp gs, gh;
#pragma unroll
    for (int l = 0; l < 8; l++) { gs.v[l] = F::SQRT_G[i][j][l]; gh.v[l] = F::SQRT_GH[i][j][l]; }
    c = fu_mul<F>(c, fu_slice<F, SH>(gs.v));
    r = fu_mul<F>(r, fu_slice<F, SH>(gh.v));
  }
  fu_to_packed<F>(out->v, fu_mul<F>(r, to_r));
  fp chk; fu_to_packed<F>(chk.v, fu_mul<F>(fu_mul<F>(fu_sqr<F>(r), vv), to_r));    // x^2 v = u
  return !odd && fp_eq(chk, u);
}

}  // namespace avrf
