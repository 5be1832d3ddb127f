// sw_map.h -- the short-Weierstrass PRESENTATION of a twisted-Edwards suite (S::SW_CODEC; Bandersnatch-SW-SHA512-TAI,
// src/suites/bandersnatch_sw.rs:60-112).  The suite's Affine type is SWAffine, so every point that is serialised -- absorbed
// by a transcript, written into a proof, hashed -- takes ark-serialize's 33-byte compressed SW form: LE32(x) followed by a flag
// byte (bit 7: y is the larger root, bit 6: infinity; a 255-bit modulus leaves one spare bit and SWFlags needs two).  The group
// arithmetic of every kernel stays in the twisted-Edwards model; this header is the bridge, the maps of
// src/utils/te_sw_map.rs:32-68 (group isomorphisms that take the SW generator to the TE generator):
//   te_to_sw: (x, y) -> Montgomery (u, v) = ((1 + y) / (1 - y), (1 + y) / (x (1 - y))) -> ((u + A/3) / B, v / B)
//   sw_to_te: (x, y) -> (u, v) = (B x - A/3, B y) -> (u / v, (u - 1) / (u + 1))
#pragma once
#include "te.h"

namespace avrf {

struct sw_enc { fp x; uint8_t flag; };     // x: plain little-endian integer

// twisted-Edwards affine point (Montgomery-form coordinates) -> its 33-byte SW encoding; one inversion.
// The identity and the order-2 point have no SW image under the map (te_to_sw -> None): encoded as infinity.
template <class S> AVRF_DI sw_enc sw_encode_te(const fp &xm, const fp &ym) {
  using Fq = typename S::Fq;
  sw_enc r; r.x = fp_zero(); r.flag = 0x40;
  if constexpr (S::SW_NATIVE) {            // the point IS the short-Weierstrass point: LE32(x) || sign of y; (0, 0) = infinity
    if (fp_is_zero(xm) && fp_is_zero(ym)) return r;
    r.x = fp_from_mont<Fq>(xm); r.flag = fp_is_negative_mont<Fq>(ym) ? 0x80 : 0x00;
    return r;
  }
  const fp one = fp_one<Fq>();
  const fp vd = fp_sub<Fq>(one, ym), wd = fp_mul<Fq>(xm, vd);
  const fp den = fp_mul<Fq>(vd, wd);
  if (fp_is_zero(den)) return r;
  const fp i = fp_inv<Fq>(den);
  const fp num = fp_add<Fq>(one, ym);
  const fp v = fp_mul<Fq>(num, fp_mul<Fq>(i, wd)), w = fp_mul<Fq>(num, fp_mul<Fq>(i, vd));   // num / vd, num / wd
  const fp binv = fp_const<Fq>(S::MONT_BINV);
  const fp xs = fp_mul<Fq>(binv, fp_add<Fq>(v, fp_const<Fq>(S::MONT_A3))), ys = fp_mul<Fq>(binv, w);
  r.x = fp_from_mont<Fq>(xs);
  r.flag = fp_is_negative_mont<Fq>(ys) ? 0x80 : 0x00;
  return r;
}
// the same for K points (canonical plain coordinates in) with ONE inversion (Montgomery's trick): the batch verifiers' prepare
// kernels absorb 4 - 5 points per item, and a field inversion is ~400 multiplications
template <class S, int K> AVRF_DI void sw_encode_te_many(const fp (&xp)[K], const fp (&yp)[K], sw_enc (&out)[K]) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) {            // nothing to invert: the encoding reads the canonical coordinates
    for (int k = 0; k < K; k++) {
      out[k].x = xp[k];
      out[k].flag = (fp_is_zero(xp[k]) && fp_is_zero(yp[k])) ? 0x40 : fp_is_negative_plain<Fq>(yp[k]) ? 0x80 : 0x00;
    }
    return;
  }
  const fp one = fp_one<Fq>();
  fp vd[K], wd[K], den[K], pre[K], ym[K];
  bool inf[K];
  fp acc = one;
  for (int k = 0; k < K; k++) {
    ym[k] = fp_to_mont<Fq>(yp[k]);
    vd[k] = fp_sub<Fq>(one, ym[k]); wd[k] = fp_mul<Fq>(fp_to_mont<Fq>(xp[k]), vd[k]); den[k] = fp_mul<Fq>(vd[k], wd[k]);
    inf[k] = fp_is_zero(den[k]);
    if (inf[k]) den[k] = one;
    pre[k] = acc; acc = fp_mul<Fq>(acc, den[k]);
  }
  fp inv = fp_inv<Fq>(acc);
  const fp binv = fp_const<Fq>(S::MONT_BINV), a3 = fp_const<Fq>(S::MONT_A3);
  for (int k = K - 1; k >= 0; k--) {
    const fp i = fp_mul<Fq>(inv, pre[k]);                                   // 1 / den[k]
    inv = fp_mul<Fq>(inv, den[k]);
    if (inf[k]) { out[k].x = fp_zero(); out[k].flag = 0x40; continue; }
    const fp num = fp_add<Fq>(one, ym[k]);
    const fp v = fp_mul<Fq>(num, fp_mul<Fq>(i, wd[k])), w = fp_mul<Fq>(num, fp_mul<Fq>(i, vd[k]));
    out[k].x = fp_from_mont<Fq>(fp_mul<Fq>(binv, fp_add<Fq>(v, a3)));
    out[k].flag = fp_is_negative_mont<Fq>(fp_mul<Fq>(binv, w)) ? 0x80 : 0x00;
  }
}
// SWAffine::get_point_from_x_unchecked(x, greatest) followed by sw_to_te: x a plain integer < q.  false: no point with this x,
// or a point without a twisted-Edwards image (v = 0 or u = -1).
template <class S> AVRF_DI bool sw_decode_te(const fp &x_plain, bool greatest, fp &xm_out, fp &ym_out) {
  using Fq = typename S::Fq;
  const fp one = fp_one<Fq>();
  const fp x = fp_to_mont<Fq>(x_plain);
  const fp rhs = fp_add<Fq>(fp_mul<Fq>(fp_add<Fq>(fp_sqr<Fq>(x), fp_const<Fq>(S::SW_A)), x), fp_const<Fq>(S::SW_B));
  fp y;
  if (!fp_sqrt_nf<Fq>(rhs, &y)) return false;
  if (fp_is_negative_mont<Fq>(y) != greatest) y = fp_neg<Fq>(y);
  if constexpr (S::SW_NATIVE) { xm_out = x; ym_out = y; return true; }
  const fp b = fp_const<Fq>(S::MONT_B);
  const fp mx = fp_sub<Fq>(fp_mul<Fq>(b, x), fp_const<Fq>(S::MONT_A3)), my = fp_mul<Fq>(b, y);
  const fp up1 = fp_add<Fq>(mx, one), den = fp_mul<Fq>(my, up1);
  if (fp_is_zero(den)) return false;
  const fp i = fp_inv<Fq>(den);
  xm_out = fp_mul<Fq>(mx, fp_mul<Fq>(i, up1));                               // u / v
  ym_out = fp_mul<Fq>(fp_sub<Fq>(mx, one), fp_mul<Fq>(i, my));               // (u - 1) / (u + 1)
  return true;
}

}  // namespace avrf
