// vrf_single.hip -- batches of INDEPENDENT per-item proofs / verifications, one lane per item
// (SURVEY.md §7.1 K7): every lane runs its own Fiat-Shamir transcript and its own scalar
// multiplications; there is no cross-lane dependency, so the kernels are pure VALU streams.
//
//   k_thin_prove     thin::Prover::prove      src/thin.rs:111-129
//   k_thin_verify    thin::Verifier::verify   src/thin.rs:131-165
//   k_ped_prove      pedersen::Prover::prove  src/pedersen.rs:136-186
//   k_ped_verify     pedersen::Verifier::verify src/pedersen.rs:188-249
//   k_smul           Secret::from_scalar / Secret::output  src/lib.rs:331-334,391-393
// The merged I/O pair follows vrf_transcript_from_iter / merge_ios (src/utils/common.rs:181-202,
// 389-419); any summation order gives the same group element.
#include "te_quad.h"     // BEFORE proto_dev.h routes fp_mul to its out-of-line form: the quad operations of the few-items kernels are
                         // latency chains and inline their two or three multiplier blocks (and use the dedicated squaring)
#include "vrf_batch.h"
#include "proto_dev.h"
#include "glv.h"
#include "fpu_sqrt.h"
#include "suite_dispatch.h"

// Built once per suite (-DAVRF_TU_SUITE=<id>: the kernels of that suite and the explicit instantiation of SingleOps<S>) and
// once without it (the run-time dispatch below); csrc/Makefile.  One translation unit for all suites took eight minutes.
namespace avrf {
#ifdef AVRF_TU_SUITE
// waves per SIMD the register allocator must leave room for in the per-item protocol kernels (1: the whole register file)
#ifndef AVRF_ITEM_WAVES
#define AVRF_ITEM_WAVES 1
#endif

// (I_m, O_m) = sum_i z_i * (I_i, O_i) over `m` caller pairs; z stream from `dseed`.
// first_is_one: the first caller pair takes z = 1 (Pedersen); otherwise pair i takes chunk i (Thin,
// whose z_0 = 1 belongs to the Schnorr pair handled by the caller).
template <class S, class R>
AVRF_DI void merge_pairs(te_ext *ws, const uint8_t *ios_xy, uint32_t m, R &dseed, bool first_is_one,
                         te_ext &im, te_ext &om) {
  for (uint32_t i = 0; i < m; i++) {
    te_pre pi = pre_from_xy<S>(ios_xy + 128 * (size_t)i), po = pre_from_xy<S>(ios_xy + 128 * (size_t)i + 64);
    if (first_is_one && i == 0) { im = te_madd<S>(im, pi); om = te_madd<S>(om, po); continue; }
    fp z = xof128(dseed, first_is_one ? i - 1 : i);
    im = te_add<S>(im, te_smul_ws<S>(ws, pi, z, 128));
    om = te_add<S>(om, te_smul_ws<S>(ws, po, z, 128));
  }
}
template <class S> AVRF_DI te_pre g_pre() {
  using Fq = typename S::Fq; te_pre g; g.x = fp_const<Fq>(S::G_X); g.y = fp_const<Fq>(S::G_Y); g.k = fp_const<Fq>(S::G_K); return g;
}
template <class S> AVRF_DI te_pre b_pre() {
  using Fq = typename S::Fq; te_pre g; g.x = fp_const<Fq>(S::B_X); g.y = fp_const<Fq>(S::B_Y); g.k = fp_const<Fq>(S::B_K); return g;
}
template <class S> AVRF_DI te_pre pre_from_aff(const te_aff &a) { return te_make_pre<S>(a.x, a.y); }
// normalise two points with one inversion
template <class S> AVRF_DI void to_aff2(const te_ext &p, const te_ext &q, te_aff &pa, te_aff &qa) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) { pa = te_to_aff<S>(p); qa = te_to_aff<S>(q); return; }   // (either may be the point at infinity)
  fp zz = fp_mul<Fq>(p.z, q.z), inv = fp_inv<Fq>(zz);
  fp pi = fp_mul<Fq>(inv, q.z), qi = fp_mul<Fq>(inv, p.z);
  pa.x = fp_mul<Fq>(p.x, pi); pa.y = fp_mul<Fq>(p.y, pi); qa.x = fp_mul<Fq>(q.x, qi); qa.y = fp_mul<Fq>(q.y, qi);
}

// ---------------------------------------------------------------- scalar multiplication

// tab[(base * 32 + w) * 256 + d] = d * 2^(8w) * P for P = G (base 0) and BLINDING_BASE (base 1); one lane per entry
template <class S>
__global__ void __launch_bounds__(128)
k_fixed_table(te_pre *__restrict__ tab) {
  uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= FIXED_TABLE_POINTS) return;
  const uint32_t d = id & 255u, w = (id >> 8) & 31u, base = id >> 13;
  if (!d) return;
  fp k = fp_zero(); k.v[w >> 2] = d << (8 * (w & 3));
  te_aff r = te_to_aff<S>(te_smul<S>(base ? b_pre<S>() : g_pre<S>(), k, 8 * (int)w + 8));
  store_pre(tab + id, te_make_pre<S>(r.x, r.y));
}

// out = k * P (P = G when points_xy == nullptr: fixed-base table)
template <class S>
__global__ void __launch_bounds__(128)
k_smul(const uint8_t *__restrict__ scalars, const uint8_t *__restrict__ points_xy, uint32_t n, uint8_t *__restrict__ out_xy,
       uint32_t *__restrict__ flags, const te_pre *__restrict__ fixed) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fp k = fp_load_le(scalars + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(k) ? FLAG_SCALAR : 0;
  te_pre p;
  if (points_xy) {
    fp x = fp_load_le(points_xy + 64 * (size_t)j), y = fp_load_le(points_xy + 64 * (size_t)j + 32);
    f |= point_flags<S>(x, y) & FLAG_RANGE;
    p = pre_from_xy<S>(points_xy + 64 * (size_t)j);
  }
  // a variable base may be ANY curve point here (this is plain `P * k`, src/lib.rs:391-393): the literal 4-bit-window product, not
  // the endomorphism split, which equals k P only inside the prime-order subgroup (glv.h)
  te_aff r = te_to_aff<S>(points_xy ? te_smul<S>(p, k, Fr::BITS) : te_smul_fixed<S>(fixed, FIXED_G, k));
  store_xy<S>(out_xy + 64 * (size_t)j, r);
  if (f) atomicOr(flags, f);
}

// s * I_m - c * O_m for the Thin / Tiny verifiers, I_m = G + sum z_i I_i, O_m = pk + sum z_i O_i (thin.rs:158-161, tiny.rs:207).
// One caller pair (the common case): expanded to s G + (s z) I - c pk - (c z) O -- the generator term comes from the fixed-base
// table and no merged pair has to be normalised: 253 + 128 doublings instead of 2 x 128 + 253 and an inversion.  More pairs:
// merge first (one 128-bit multiplication per point), then the two-point form.  Any order gives the same group element.
template <class S>
AVRF_DI te_ext schnorr_lhs(const BatchDev &b, te_ext *ws, const uint8_t *ios, const uint8_t *pk_xy, uint32_t m, const suite_tr<S> &t, const fp &s, const fp &c) {
  using Fr = typename S::Fr;
  te_pre op = pre_from_xy<S>(pk_xy);
  if (m == 0) return te_add<S>(te_smul_fixed<S>(b.fixed, FIXED_G, s), te_smul_ws<S>(ws, te_pre_neg<S>(op), c, 128));
  auto dseed = delin_seed(t);
  if (m == 1) {
    const fp z = xof128(dseed, 0);
    const fp sz = fp_mul<Fr>(fp_to_mont<Fr>(s), z), cz = fp_mul<Fr>(fp_to_mont<Fr>(c), z);   // plain products mod r
    te_ext acc = te_smul_multi_glv<S, true>(ws, pre_from_xy<S>(ios), sz, te_pre_neg<S>(pre_from_xy<S>(ios + 64)), cz, te_pre_neg<S>(op), c);
    return te_add<S>(acc, te_smul_fixed<S>(b.fixed, FIXED_G, s));
  }
  te_ext im = te_from_pre<S>(g_pre<S>()), om = te_from_pre<S>(op);
  merge_pairs<S>(ws, ios, m, dseed, false, im, om);
  te_aff ia, oa; to_aff2<S>(im, om, ia, oa);
  return te_smul_multi_glv<S, false>(ws, pre_from_aff<S>(ia), s, pre_from_aff<S>(ia), fp_zero(), te_pre_neg<S>(pre_from_aff<S>(oa)), c);
}
// the item's window-table slots in the context's workspace (proto_dev.h te_smul_ws)
AVRF_DI te_ext *item_ws(const BatchDev &b, uint32_t j) { return b.tabs + (size_t)(j - b.first) * ITEM_TAB_SLOTS; }

// ---------------------------------------------------------------- Thin VRF

// TINY: tiny::Prover::prove (src/tiny.rs:163-176) -- the same steps under scheme tag 0x00; the proof keeps the challenge
// instead of the nonce commitment: LE16(c) || LE32(s), 48 bytes (src/tiny.rs:60-78)
template <class S, bool TINY>
__global__ void __launch_bounds__(128, AVRF_ITEM_WAVES)
k_thin_prove(BatchDev b, uint8_t *__restrict__ proofs_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = b.first + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0;
  // public key: cached by Secret in the reference (src/lib.rs:331-334)
  fp pkx, pky;                                                                  // canonical coordinates of the public key
  if (b.pks_xy) { pkx = fp_load_le(b.pks_xy + 64 * (size_t)j); pky = fp_load_le(b.pks_xy + 64 * (size_t)j + 32); }
  else { using Fq = typename S::Fq; te_aff pk = te_to_aff<S>(te_smul_fixed<S>(b.fixed, FIXED_G, sk)); pkx = fp_from_mont<Fq>(pk.x); pky = fp_from_mont<Fq>(pk.y); }
  suite_tr<S> t; uint32_t pf = 0;
  tr_base<S>(t, TINY ? DS_TINY : DS_THIN, true, pkx, pky, ios, m, b.ads + ad0, adl, &pf);   // thin.rs:112, tiny.rs:164
  f |= pf & FLAG_RANGE;
  // R = k I_m with the merged input I_m = G + sum z_i I_i (only the input is needed by the prover; thin.rs:115-119).
  // No pair: k G from the fixed-base table.  One pair: k G + (k z) I -- table + ONE 253-bit multiplication, nothing to merge or
  // normalise.  More: merge (a 128-bit multiplication per input), then k I_m.
  fp k = nonce<S>(sk, t);                                                       // thin.rs:115
  const fp k_plain = fp_from_mont<Fr>(k);
  te_ext rr;
  if (m <= 1) {
    rr = te_smul_fixed<S>(b.fixed, FIXED_G, k_plain);
    if (m) {
      auto dseed = delin_seed(t);
      const fp kz = fp_mul<Fr>(k, xof128(dseed, 0));                            // Montgomery k times plain z: plain k z mod r
      rr = te_add<S>(rr, te_smul_glv<S>(item_ws(b, j), pre_from_xy<S>(ios), kz));
    }
  } else {
    auto dseed = delin_seed(t);
    te_ext im = te_from_pre<S>(g_pre<S>());
    for (uint32_t i = 0; i < m; i++) {
      te_pre pi = pre_from_xy<S>(ios + 128 * (size_t)i);
      im = te_add<S>(im, te_smul_ws<S>(item_ws(b, j), pi, xof128(dseed, i), 128));
    }
    rr = te_smul_glv<S>(item_ws(b, j), pre_from_aff<S>(te_to_aff<S>(im)), k_plain);
  }
  te_aff r = te_to_aff<S>(rr);                                                  // thin.rs:119
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, r);    // thin.rs:122
  fp c = fp_to_mont<Fr>(challenge_finish(tc));
  fp s = fp_add<Fr>(k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));                      // thin.rs:125
  if (TINY) {
    const fp cp = fp_from_mont<Fr>(c);
    uint8_t *o = proofs_out + 48 * (size_t)j;
    for (int i = 0; i < 4; i++) for (int bb = 0; bb < 4; bb++) o[4 * i + bb] = (uint8_t)(cp.v[i] >> (8 * bb));
    fp_store_le(o + 16, fp_from_mont<Fr>(s));
  } else {
    store_xy<S>(proofs_out + 96 * (size_t)j, r);
    fp_store_le(proofs_out + 96 * (size_t)j + 64, fp_from_mont<Fr>(s));
  }
  if (f) atomicOr(flags, f);
}

// ONE proof (thin / tiny), split around the MSM engine: the prover's only group work is R = k I_m = k G + sum_i (k z_i) I_i
// (thin.rs:115-119), an MSM of 1 + m terms whose doubling chain the host folds ~10 x faster than a lone wave walks it (msm.hip
// k_msm_tiny_bits).  k_thin_prove_begin: transcript, nonce, the terms; the nonce and the transcript stay in device memory.
// k_thin_prove_end: challenge over the normalised R (it comes back as canonical x || y), response, proof bytes.  One lane each.
template <class S> struct ProveState { suite_tr<S> t; fp k; uint32_t f; };
template <class S, bool TINY>
__global__ void __launch_bounds__(64)
k_thin_prove_begin(BatchDev b, uint32_t *__restrict__ scalars, te_pre *__restrict__ pre, uint8_t *__restrict__ state) {
  using Fr = typename S::Fr;
  if (threadIdx.x) return;
  const uint32_t j = b.first;
  const uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0;
  fp pkx, pky;
  if (b.pks_xy) { pkx = fp_load_le(b.pks_xy + 64 * (size_t)j); pky = fp_load_le(b.pks_xy + 64 * (size_t)j + 32); }
  else { using Fq = typename S::Fq; te_aff pk = te_to_aff<S>(te_smul_fixed<S>(b.fixed, FIXED_G, sk)); pkx = fp_from_mont<Fq>(pk.x); pky = fp_from_mont<Fq>(pk.y); }
  ProveState<S> *st = reinterpret_cast<ProveState<S> *>(state);
  suite_tr<S> t; uint32_t pf = 0;
  tr_base<S>(t, TINY ? DS_TINY : DS_THIN, true, pkx, pky, ios, m, b.ads + ad0, adl, &pf);
  f |= pf & FLAG_RANGE;
  const fp k = nonce<S>(sk, t);                                                 // thin.rs:115 (Montgomery)
  store_fp(scalars, fp_from_mont<Fr>(k));
  store_pre(pre, g_pre<S>());
  if (m) {
    auto dseed = delin_seed(t);
    for (uint32_t i = 0; i < m; i++) {
      store_fp(scalars + 8 * (size_t)(1 + i), fp_mul<Fr>(k, xof128(dseed, i)));  // Montgomery k times plain z_i: plain k z_i mod r
      store_pre(pre + 1 + i, pre_from_xy<S>(ios + 128 * (size_t)i));
    }
  }
  st->t = t; st->k = k; st->f = f;
}
template <class S, bool TINY>
__global__ void __launch_bounds__(64)
k_thin_prove_end(BatchDev b, const uint8_t *__restrict__ state, const uint8_t *__restrict__ r_xy, uint8_t *__restrict__ proofs_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  if (threadIdx.x) return;
  const uint32_t j = b.first;
  const ProveState<S> *st = reinterpret_cast<const ProveState<S> *>(state);
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j), k = st->k;
  te_aff r; r.x = fp_to_mont<Fq>(fp_load_le(r_xy)); r.y = fp_to_mont<Fq>(fp_load_le(r_xy + 32));   // thin.rs:119
  suite_tr<S> tc = st->t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, r);                 // thin.rs:122
  const fp c = fp_to_mont<Fr>(challenge_finish(tc));
  const fp s = fp_add<Fr>(k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));                                  // thin.rs:125
  if (TINY) {
    const fp cp = fp_from_mont<Fr>(c);
    uint8_t *o = proofs_out + 48 * (size_t)j;
    for (int i = 0; i < 4; i++) for (int bb = 0; bb < 4; bb++) o[4 * i + bb] = (uint8_t)(cp.v[i] >> (8 * bb));
    fp_store_le(o + 16, fp_from_mont<Fr>(s));
  } else {
    store_xy<S>(proofs_out + 96 * (size_t)j, r);
    fp_store_le(proofs_out + 96 * (size_t)j + 64, fp_from_mont<Fr>(s));
  }
  if (st->f) atomicOr(flags, st->f);
}

// tiny::Verifier::verify (src/tiny.rs:178-214): R = s I_m - c O_m, recompute the challenge, compare with c
template <class S>
__global__ void __launch_bounds__(128, AVRF_ITEM_WAVES)
k_tiny_verify(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr;
  uint32_t j = b.first + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pk_xy = b.pks_xy + 64 * (size_t)j, *pr = b.proofs + 48 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_TINY, true, pk_xy, ios, m, b.ads + ad0, adl, &f);
  fp c = fp_zero();
  for (int i = 0; i < 4; i++) c.v[i] = (uint32_t)pr[4 * i] | ((uint32_t)pr[4 * i + 1] << 8) | ((uint32_t)pr[4 * i + 2] << 16) | ((uint32_t)pr[4 * i + 3] << 24);
  fp s = fp_load_le(pr + 16);
  if (ge_p<Fr>(s)) f |= FLAG_SCALAR;
  if (f) { status[j] = 2; return; }                                             // InvalidData, tiny.rs:186-198
  te_aff r = te_to_aff<S>(schnorr_lhs<S>(b, item_ws(b, j), ios, pk_xy, m, t, s, c));            // tiny.rs:207
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, r);
  const fp c_exp = challenge_finish(tc);                                        // plain, 128 bits
  status[j] = fp_eq(c_exp, c) ? 0 : 1;
}

template <class S>
__global__ void __launch_bounds__(128, AVRF_ITEM_WAVES)
k_thin_verify(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  uint32_t j = b.first + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pk_xy = b.pks_xy + 64 * (size_t)j, *pr = b.proofs + 96 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_THIN, true, pk_xy, ios, m, b.ads + ad0, adl, &f);
  fp rx = fp_load_le(pr), ry = fp_load_le(pr + 32), s = fp_load_le(pr + 64);
  if (point_flags<S>(rx, ry) & FLAG_RANGE) f |= FLAG_RANGE;
  if (ge_p<Fr>(s)) f |= FLAG_SCALAR;
  if (f) { status[j] = 2; return; }                                             // InvalidData, thin.rs:140-149
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_xy<S>(tc, rx, ry);
  fp c = challenge_finish(tc);                                                  // plain, 128 bits
  // s*I_m - c*O_m == R   (thin.rs:158-161)
  te_ext lhs = schnorr_lhs<S>(b, item_ws(b, j), ios, pk_xy, m, t, s, c);
  te_pre rp = te_make_pre<S>(fp_to_mont<Fq>(rx), fp_to_mont<Fq>(ry));
  status[j] = ext_eq_aff<S>(lhs, rp) ? 0 : 1;
}

// ---------------------------------------------------------------- Thin VRF, few items: one item spread over half a wave
//
// The lane-per-item kernels above walk ~3 000 DEPENDENT field multiplications per item (128 doublings + up to 192 additions on
// one chain): 2.1-2.4 ms however few items there are, eleven times the reference's 188 us for one verification
// (benches/SUMMARY.md:53-54).  For calls of up to AVRF_WAVE_ITEMS_MAX items an item takes 32 lanes instead: eight quads, each quad one
// scalar multiplication with the four coordinates of a point on its four lanes (te_quad.h: a doubling is two rounds of
// multiplications, an addition three).  The equation's terms -- s G, (s z) I, -c pk, -(c z) O for the verifier (thin.rs:158-161,
// one pair), k G, (k z) I for the prover (thin.rs:115-119) -- are split with the endomorphism where the curve has one
// (k P = k1 P + k2 psi(P), 127-bit halves, glv.h), one half per quad, so every quad runs the SAME 43 windows of three doublings
// and one addition; suites without the endomorphism run 85 windows on four quads.  The eight partial results are summed with three
// butterfly steps.  Transcript, challenge and scalar preparation are computed redundantly by the 32 lanes of the item (same
// instruction stream, no divergence).  Results are the same group elements, hence the same verdicts and proof bytes.
// One pair per item and a twisted-Edwards suite only; everything else stays on the lane-per-item kernels.
template <class S> struct WaveTerm { fp coord; fp k; bool ok; };
// coordinate jc of this quad's base point and the quad's scalar: term (p affine Montgomery, scalar k plain, negated?) split over the
// quads 2 t (k1, P) and 2 t + 1 (k2, psi(P)) when the suite has the endomorphism; else quad 2 t takes (k, P), quad 2 t + 1 idles
template <class S> AVRF_DI WaveTerm<S> wave_term(const fp &px, const fp &py, const fp &k, bool neg, uint32_t half, uint32_t jc) {
  using Fq = typename S::Fq;
  WaveTerm<S> r; r.ok = true;
  te_ext e; e.x = px; e.y = py; e.t = fp_mul<Fq>(px, py); e.z = fp_one<Fq>();
  fp ks = k; bool ng = neg;
  if constexpr (S::HAS_GLV) {
    te_pre p; p.x = px; p.y = py; p.k = fp_zero();
    te_ext q;
    r.ok = te_endo<S>(p, q);                                        // (x y = 0 or y^2 = b: the caller sends the item to the lane-per-item kernel)
    const glv_scalars g = glv_decompose<S>(k);
    ks = half ? g.k2 : g.k1; ng = neg != (half ? g.n2 : g.n1);
    if (half) e = q;
  } else if (half) { e = te_identity<S>(); ks = fp_zero(); }
  fp c = jc == 0 ? e.x : jc == 1 ? e.y : jc == 2 ? e.t : e.z;
  if (ng && (jc == 0 || jc == 2)) c = fp_neg<Fq>(c);               // -(X, Y, T, Z) = (-X, Y, -T, Z)
  r.coord = c; r.k = ks;
  return r;
}
// sum of the eight quads' points of a 32-lane item group, in every quad
template <class S> AVRF_DI fp wave_group_sum(fp v, uint32_t jc) {
#pragma unroll 1
  for (int off = 16; off >= 4; off >>= 1) v = q_add<S>(v, fp_shfl_xor(v, off), jc);
  return v;
}

template <class S>
__global__ void __launch_bounds__(64)
k_thin_verify_wave(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  constexpr int NBITS = S::HAS_GLV ? 128 : Fr::BITS;
  const uint32_t lane = threadIdx.x & 63, gl = lane & 31, q = gl >> 2, jc = gl & 3, tq = q >> 1, half = q & 1;
  uint32_t j = b.first + 2 * blockIdx.x + (lane >> 5);
  const bool live = j < b.n;
  if (!live) j = b.n - 1;                                                       // (all lanes stay in step: the cross-lane moves need them)
  const uint32_t io0 = b.io_off[j], ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pk_xy = b.pks_xy + 64 * (size_t)j, *pr = b.proofs + 96 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_THIN, true, pk_xy, ios, 1, b.ads + ad0, adl, &f);
  const fp rx = fp_load_le(pr), ry = fp_load_le(pr + 32), s = fp_load_le(pr + 64);
  if (point_flags<S>(rx, ry) & FLAG_RANGE) f |= FLAG_RANGE;
  if (ge_p<Fr>(s)) f |= FLAG_SCALAR;                                            // InvalidData, thin.rs:140-149 (reported below)
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_xy<S>(tc, rx, ry);
  const fp c = challenge_finish(tc);                                            // plain, 128 bits
  auto dseed = delin_seed(t);
  const fp z = xof128(dseed, 0);
  // this quad's term of  s G + (s z) I - c pk - (c z) O  (thin.rs:158-161 expanded for one pair, as schnorr_lhs does)
  const uint8_t *src = tq == 1 ? ios : tq == 2 ? pk_xy : ios + 64;
  const te_pre g = g_pre<S>();
  const fp px = tq == 0 ? g.x : fp_to_mont<Fq>(fp_load_le(src)), py = tq == 0 ? g.y : fp_to_mont<Fq>(fp_load_le(src + 32));
  const fp sm = fp_to_mont<Fr>(tq < 2 ? s : c);
  const fp k = (tq & 1) ? fp_mul<Fr>(sm, z) : (tq < 2 ? s : c);                 // s, s z, c, c z as plain integers mod r
  const WaveTerm<S> w = wave_term<S>(px, py, k, tq >= 2, half, jc);
  fp v = q_smul<S, NBITS>(w.coord, w.k, jc);
  v = wave_group_sum<S>(v, jc);
  const fp X = qperm<0, 0, 0, 0>(v), Y = qperm<1, 1, 1, 1>(v), Z = qperm<3, 3, 3, 3>(v);
  const bool eq = fp_eq(X, fp_mul<Fq>(fp_to_mont<Fq>(rx), Z)) && fp_eq(Y, fp_mul<Fq>(fp_to_mont<Fq>(ry), Z));   // == R  (ext_eq_aff)
  // a degenerate base point in ANY quad of the item: no verdict from here
  uint32_t bad = w.ok ? 0u : 1u;
  for (int off = 16; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
  if (live && gl == 0) status[j] = f ? 2 : bad ? (int32_t)AVRF_WAVE_FALLBACK : eq ? 0 : 1;
}

// thin::Prover::prove (thin.rs:111-135) for one pair, the same layout: R = k G + (k z) I on four quads (the other four idle)
template <class S, bool TINY>
__global__ void __launch_bounds__(64)
k_thin_prove_wave(BatchDev b, uint8_t *__restrict__ proofs_out, uint32_t *__restrict__ flags, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  constexpr int NBITS = S::HAS_GLV ? 128 : Fr::BITS;
  const uint32_t lane = threadIdx.x & 63, gl = lane & 31, q = gl >> 2, jc = gl & 3, tq = q >> 1, half = q & 1;
  uint32_t j = b.first + 2 * blockIdx.x + (lane >> 5);
  const bool live = j < b.n;
  if (!live) j = b.n - 1;
  const uint32_t io0 = b.io_off[j], ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0;
  const fp pkx = fp_load_le(b.pks_xy + 64 * (size_t)j), pky = fp_load_le(b.pks_xy + 64 * (size_t)j + 32);
  suite_tr<S> t; uint32_t pf = 0;
  tr_base<S>(t, TINY ? DS_TINY : DS_THIN, true, pkx, pky, ios, 1, b.ads + ad0, adl, &pf);   // thin.rs:112, tiny.rs:164
  f |= pf & FLAG_RANGE;
  const fp k = nonce<S>(sk, t);                                                  // thin.rs:115 (Montgomery)
  auto dseed = delin_seed(t);
  const fp kz = fp_mul<Fr>(k, xof128(dseed, 0));                                 // Montgomery k times plain z: plain k z mod r
  const te_pre g = g_pre<S>();
  const fp px = tq == 0 ? g.x : fp_to_mont<Fq>(fp_load_le(ios)), py = tq == 0 ? g.y : fp_to_mont<Fq>(fp_load_le(ios + 32));
  const fp ks = tq == 0 ? fp_from_mont<Fr>(k) : tq == 1 ? kz : fp_zero();
  const WaveTerm<S> w = wave_term<S>(px, py, ks, false, half, jc);
  fp v = q_smul<S, NBITS>(w.coord, w.k, jc);
  v = wave_group_sum<S>(v, jc);
  const fp zi = fp_inv_few<Fq>(qperm<3, 3, 3, 3>(v));
  te_aff r; r.x = fp_mul<Fq>(qperm<0, 0, 0, 0>(v), zi); r.y = fp_mul<Fq>(qperm<1, 1, 1, 1>(v), zi);      // thin.rs:119
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, r);    // thin.rs:122
  const fp c = fp_to_mont<Fr>(challenge_finish(tc));
  const fp s = fp_add<Fr>(k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));                 // thin.rs:125
  uint32_t bad = (w.ok || tq >= 2) ? 0u : 1u;
  for (int off = 16; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
  if (live && gl == 0) {
    status[j] = bad ? (int32_t)AVRF_WAVE_FALLBACK : 0;
    if (TINY) {                                                                   // LE16(c) || LE32(s), tiny.rs:60-78
      const fp cp = fp_from_mont<Fr>(c);
      uint8_t *o = proofs_out + 48 * (size_t)j;
      for (int i = 0; i < 4; i++) for (int bb = 0; bb < 4; bb++) o[4 * i + bb] = (uint8_t)(cp.v[i] >> (8 * bb));
      fp_store_le(o + 16, fp_from_mont<Fr>(s));
    } else {
      store_xy<S>(proofs_out + 96 * (size_t)j, r);
      fp_store_le(proofs_out + 96 * (size_t)j + 64, fp_from_mont<Fr>(s));
    }
    if (f) atomicOr(flags, f);
  }
}

// tiny::Verifier::verify (tiny.rs:178-214), one pair: R = s G + (s z) I - c pk - (c z) O on the eight quads, normalised (one
// inversion), the challenge recomputed over it and compared with the proof's
template <class S>
__global__ void __launch_bounds__(64)
k_tiny_verify_wave(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  constexpr int NBITS = S::HAS_GLV ? 128 : Fr::BITS;
  const uint32_t lane = threadIdx.x & 63, gl = lane & 31, q = gl >> 2, jc = gl & 3, tq = q >> 1, half = q & 1;
  uint32_t j = b.first + 2 * blockIdx.x + (lane >> 5);
  const bool live = j < b.n;
  if (!live) j = b.n - 1;
  const uint32_t io0 = b.io_off[j], ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pk_xy = b.pks_xy + 64 * (size_t)j, *pr = b.proofs + 48 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_TINY, true, pk_xy, ios, 1, b.ads + ad0, adl, &f);
  fp c = fp_zero();
  for (int i = 0; i < 4; i++) c.v[i] = (uint32_t)pr[4 * i] | ((uint32_t)pr[4 * i + 1] << 8) | ((uint32_t)pr[4 * i + 2] << 16) | ((uint32_t)pr[4 * i + 3] << 24);
  const fp s = fp_load_le(pr + 16);
  if (ge_p<Fr>(s)) f |= FLAG_SCALAR;                                            // InvalidData, tiny.rs:186-198 (reported below)
  auto dseed = delin_seed(t);
  const fp z = xof128(dseed, 0);
  const uint8_t *src = tq == 1 ? ios : tq == 2 ? pk_xy : ios + 64;
  const te_pre g = g_pre<S>();
  const fp px = tq == 0 ? g.x : fp_to_mont<Fq>(fp_load_le(src)), py = tq == 0 ? g.y : fp_to_mont<Fq>(fp_load_le(src + 32));
  const fp sm = fp_to_mont<Fr>(tq < 2 ? s : c);
  const fp k = (tq & 1) ? fp_mul<Fr>(sm, z) : (tq < 2 ? s : c);
  const WaveTerm<S> w = wave_term<S>(px, py, k, tq >= 2, half, jc);
  fp v = q_smul<S, NBITS>(w.coord, w.k, jc);
  v = wave_group_sum<S>(v, jc);
  const fp zi = fp_inv_few<Fq>(qperm<3, 3, 3, 3>(v));
  te_aff r; r.x = fp_mul<Fq>(qperm<0, 0, 0, 0>(v), zi); r.y = fp_mul<Fq>(qperm<1, 1, 1, 1>(v), zi);      // tiny.rs:207
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, r);
  const fp c_exp = challenge_finish(tc);                                        // plain, 128 bits
  uint32_t bad = w.ok ? 0u : 1u;
  for (int off = 16; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
  if (live && gl == 0) status[j] = f ? 2 : bad ? (int32_t)AVRF_WAVE_FALLBACK : fp_eq(c_exp, c) ? 0 : 1;
}

// ---------------------------------------------------------------- Pedersen VRF

template <class S>
__global__ void __launch_bounds__(128, AVRF_ITEM_WAVES)
k_ped_prove(BatchDev b, uint8_t *__restrict__ proofs_out, uint8_t *__restrict__ blindings_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = b.first + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0, pf = 0;
  suite_tr<S> t;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, m, b.ads + ad0, adl, &pf);   // pedersen.rs:142
  f |= pf & FLAG_RANGE;
  // merged input (common.rs:186-199): identity for m = 0, the pair itself for m = 1
  te_pre ip; bool have_input = m > 0;
  if (m == 1) ip = pre_from_xy<S>(ios);
  else if (m > 1) {
    auto dseed = delin_seed(t);
    te_ext im = te_identity<S>(), om = te_identity<S>();
    for (uint32_t i = 0; i < m; i++) {
      te_pre pi = pre_from_xy<S>(ios + 128 * (size_t)i);
      if (i == 0) im = te_madd<S>(im, pi); else im = te_add<S>(im, te_smul_ws<S>(item_ws(b, j), pi, xof128(dseed, i - 1), 128));
    }
    (void)om;
    ip = pre_from_aff<S>(te_to_aff<S>(im));
  }
  suite_tr<S> tb = t; tr_byte(tb, DS_PEDERSEN_BLINDING);
  fp bl = nonce<S>(sk, tb);                                                     // pedersen.rs:51-54,145
  fp bl_plain = fp_from_mont<Fr>(bl);
  te_pre pkp;
  if (b.pks_xy) pkp = pre_from_xy<S>(b.pks_xy + 64 * (size_t)j);
  else pkp = pre_from_aff<S>(te_to_aff<S>(te_smul_fixed<S>(b.fixed, FIXED_G, sk)));
  te_aff yb = te_to_aff<S>(te_madd<S>(te_smul_fixed<S>(b.fixed, FIXED_B, bl_plain), pkp));   // :148-149
  absorb_point_mont<S>(t, yb);                                                  // :152
  fp k = nonce<S>(sk, t), kb = nonce<S>(bl_plain, t);                           // :155-156
  fp k_plain = fp_from_mont<Fr>(k);
  te_ext R = te_add<S>(te_smul_fixed<S>(b.fixed, FIXED_G, k_plain), te_smul_fixed<S>(b.fixed, FIXED_B, fp_from_mont<Fr>(kb)));   // :159-161
  te_ext OK = have_input ? te_smul_glv<S>(item_ws(b, j), ip, k_plain) : te_identity<S>();     // :164
  te_aff ra, oka; to_aff2<S>(R, OK, ra, oka);                                   // :166-167
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, ra); absorb_point_mont<S>(tc, oka);
  fp c = fp_to_mont<Fr>(challenge_finish(tc));                                  // :170
  fp s = fp_add<Fr>(k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));                      // :173
  fp sb = fp_add<Fr>(kb, fp_mul<Fr>(c, bl));                                    // :175
  uint8_t *o = proofs_out + 256 * (size_t)j;
  store_xy<S>(o, yb); store_xy<S>(o + 64, ra); store_xy<S>(o + 128, oka);
  fp_store_le(o + 192, fp_from_mont<Fr>(s)); fp_store_le(o + 224, fp_from_mont<Fr>(sb));
  if (blindings_out) fp_store_le(blindings_out + 32 * (size_t)j, bl_plain);
  if (f) atomicOr(flags, f);
}

// ONE Pedersen proof split around the MSM engine like k_thin_prove_begin / _end: its three commitments are MSMs whose doubling chains
// the host folds -- Yb = pk + bl B (pedersen.rs:148-149: two terms), then, once Yb is absorbed and the nonces exist, R = k G + kb B and
// Ok = k I_m (:159-164) as two scalar vectors over the bases {G, B, I_0 ..}.  Three one-lane kernels; blinding, nonces and transcript stay
// in device memory (PedState), the item's delinearisation weights (1, z_0, z_1, ..: common.rs:186-199) beside it.
template <class S> struct PedState { suite_tr<S> t; fp bl, k, kb; uint32_t f; };
template <class S>
__global__ void __launch_bounds__(64)
k_ped_prove_begin(BatchDev b, uint32_t *__restrict__ scalars, te_pre *__restrict__ pre, uint8_t *__restrict__ state, uint32_t *__restrict__ wts) {
  using Fr = typename S::Fr;
  if (threadIdx.x) return;
  const uint32_t j = b.first;
  const uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0, pf = 0;
  PedState<S> *st = reinterpret_cast<PedState<S> *>(state);
  suite_tr<S> t;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, m, b.ads + ad0, adl, &pf);   // pedersen.rs:142
  f |= pf & FLAG_RANGE;
  if (m > 1) {                                                                  // weights of the merged input: 1, z_0, z_1, ..
    auto dseed = delin_seed(t);
    for (uint32_t i = 1; i < m; i++) store_fp(wts + 8 * (size_t)i, xof128(dseed, i - 1));
  }
  suite_tr<S> tb = t; tr_byte(tb, DS_PEDERSEN_BLINDING);
  const fp bl = nonce<S>(sk, tb);                                               // pedersen.rs:51-54,145
  // Yb = pk + bl B (:148-149): (pk, 1), (B, bl) -- or (G, sk), (B, bl) when the caller gave no public key
  fp one = fp_zero(); one.v[0] = 1;
  if (b.pks_xy) { store_pre(pre, pre_from_xy<S>(b.pks_xy + 64 * (size_t)j)); store_fp(scalars, one); }
  else { store_pre(pre, g_pre<S>()); store_fp(scalars, sk); }
  store_pre(pre + 1, b_pre<S>()); store_fp(scalars + 8, fp_from_mont<Fr>(bl));
  st->t = t; st->bl = bl; st->f = f;
}
template <class S>
__global__ void __launch_bounds__(64)
k_ped_prove_mid(BatchDev b, uint32_t *__restrict__ scalars, te_pre *__restrict__ pre, uint8_t *__restrict__ state, const uint32_t *__restrict__ wts,
                const uint8_t *__restrict__ yb_xy) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  if (threadIdx.x) return;
  const uint32_t j = b.first;
  const uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, nt = 2 + m;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  PedState<S> *st = reinterpret_cast<PedState<S> *>(state);
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  suite_tr<S> t = st->t;
  te_aff yb; yb.x = fp_to_mont<Fq>(fp_load_le(yb_xy)); yb.y = fp_to_mont<Fq>(fp_load_le(yb_xy + 32));
  absorb_point_mont<S>(t, yb);                                                  // :152
  const fp bl_plain = fp_from_mont<Fr>(st->bl);
  const fp k = nonce<S>(sk, t), kb = nonce<S>(bl_plain, t);                     // :155-156
  // vector 0: R = k G + kb B; vector 1: Ok = k I_m = sum_i (k w_i) I_i
  const fp zero = fp_zero();
  store_pre(pre, g_pre<S>()); store_pre(pre + 1, b_pre<S>());
  store_fp(scalars, fp_from_mont<Fr>(k)); store_fp(scalars + 8, fp_from_mont<Fr>(kb));
  store_fp(scalars + 8 * (size_t)nt, zero); store_fp(scalars + 8 * (size_t)(nt + 1), zero);
  for (uint32_t i = 0; i < m; i++) {
    store_pre(pre + 2 + i, pre_from_xy<S>(ios + 128 * (size_t)i));
    store_fp(scalars + 8 * (size_t)(2 + i), zero);
    store_fp(scalars + 8 * (size_t)(nt + 2 + i), i == 0 ? fp_from_mont<Fr>(k) : fp_mul<Fr>(k, load_fp(wts + 8 * (size_t)i)));   // Montgomery k times plain w_i: plain
  }
  st->t = t; st->k = k; st->kb = kb;
}
template <class S>
__global__ void __launch_bounds__(64)
k_ped_prove_end(BatchDev b, const uint8_t *__restrict__ state, const uint8_t *__restrict__ pts_xy, uint8_t *__restrict__ proofs_out,
                uint8_t *__restrict__ blindings_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  if (threadIdx.x) return;
  const uint32_t j = b.first;
  const PedState<S> *st = reinterpret_cast<const PedState<S> *>(state);
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  te_aff yb, ra, oka;                                                           // pts_xy: Yb | R | Ok, canonical x || y
  yb.x = fp_to_mont<Fq>(fp_load_le(pts_xy)); yb.y = fp_to_mont<Fq>(fp_load_le(pts_xy + 32));
  ra.x = fp_to_mont<Fq>(fp_load_le(pts_xy + 64)); ra.y = fp_to_mont<Fq>(fp_load_le(pts_xy + 96));
  oka.x = fp_to_mont<Fq>(fp_load_le(pts_xy + 128)); oka.y = fp_to_mont<Fq>(fp_load_le(pts_xy + 160));
  suite_tr<S> tc = st->t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, ra); absorb_point_mont<S>(tc, oka);
  const fp c = fp_to_mont<Fr>(challenge_finish(tc));                            // :170
  const fp s = fp_add<Fr>(st->k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));            // :173
  const fp sb = fp_add<Fr>(st->kb, fp_mul<Fr>(c, st->bl));                      // :175
  uint8_t *o = proofs_out + 256 * (size_t)j;
  store_xy<S>(o, yb); store_xy<S>(o + 64, ra); store_xy<S>(o + 128, oka);
  fp_store_le(o + 192, fp_from_mont<Fr>(s)); fp_store_le(o + 224, fp_from_mont<Fr>(sb));
  if (blindings_out) fp_store_le(blindings_out + 32 * (size_t)j, fp_from_mont<Fr>(st->bl));
  if (st->f) atomicOr(flags, st->f);
}

template <class S>
__global__ void __launch_bounds__(128, AVRF_ITEM_WAVES)
k_ped_verify(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  uint32_t j = b.first + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pr = b.proofs + 256 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, m, b.ads + ad0, adl, &f);
  fp ybx = fp_load_le(pr), yby = fp_load_le(pr + 32), rx = fp_load_le(pr + 64), ry = fp_load_le(pr + 96);
  fp okx = fp_load_le(pr + 128), oky = fp_load_le(pr + 160), s = fp_load_le(pr + 192), sb = fp_load_le(pr + 224);
  f |= point_flags<S>(ybx, yby);                                               // Yb == 0 rejected, pedersen.rs:204-206
  f |= (point_flags<S>(rx, ry) | point_flags<S>(okx, oky)) & FLAG_RANGE;
  if (ge_p<Fr>(s) || ge_p<Fr>(sb)) f |= FLAG_SCALAR;
  if (f) { status[j] = 2; return; }
  te_pre ip, op; bool have_io = m > 0;
  if (m == 1) { ip = pre_from_xy<S>(ios); op = pre_from_xy<S>(ios + 64); }
  else if (m > 1) {
    auto dseed = delin_seed(t);
    te_ext im = te_identity<S>(), om = te_identity<S>();
    merge_pairs<S>(item_ws(b, j), ios, m, dseed, true, im, om);
    te_aff ia, oa; to_aff2<S>(im, om, ia, oa);
    ip = pre_from_aff<S>(ia); op = pre_from_aff<S>(oa);
  }
  if constexpr (S::SW_CODEC) {                                                 // three 33-byte SW forms, one inversion
    const fp xs[3] = {ybx, rx, okx}, ys[3] = {yby, ry, oky};
    sw_enc enc[3]; sw_encode_te_many<S, 3>(xs, ys, enc);
    absorb_sw_enc(t, enc[0]); tr_byte(t, DS_CHALLENGE); absorb_sw_enc(t, enc[1]); absorb_sw_enc(t, enc[2]);
  } else {
    absorb_point_xy<S>(t, ybx, yby);                                           // :219
    tr_byte(t, DS_CHALLENGE); absorb_point_xy<S>(t, rx, ry); absorb_point_xy<S>(t, okx, oky);
  }
  fp c = challenge_finish(t);                                                  // :222
  // Eq1: s*I - c*O == Ok   (:229-232)
  te_ext lhs1 = have_io ? te_smul_multi_glv<S, false>(item_ws(b, j), ip, s, ip, fp_zero(), te_pre_neg<S>(op), c) : te_identity<S>();
  te_pre okp = te_make_pre<S>(fp_to_mont<Fq>(okx), fp_to_mont<Fq>(oky));
  if (!ext_eq_aff<S>(lhs1, okp)) { status[j] = 1; return; }
  // Eq2: s*G + sb*B - c*Yb == R   (:238-245)
  te_pre ybp = te_make_pre<S>(fp_to_mont<Fq>(ybx), fp_to_mont<Fq>(yby));
  // G and BLINDING_BASE are fixed: 2 x 32 table additions instead of a 253-bit joint double-and-add
  te_ext lhs2 = te_add<S>(te_smul_fixed<S>(b.fixed, FIXED_G, s), te_smul_fixed<S>(b.fixed, FIXED_B, sb));
  lhs2 = te_add<S>(lhs2, te_smul_ws<S>(item_ws(b, j), te_pre_neg<S>(ybp), c, 128));
  te_pre rp = te_make_pre<S>(fp_to_mont<Fq>(rx), fp_to_mont<Fq>(ry));
  status[j] = ext_eq_aff<S>(lhs2, rp) ? 0 : 1;
}


// ---------------------------------------------------------------- Pedersen VRF, few items (see "Thin VRF, few items" above)
//
// pedersen::Verifier::verify (pedersen.rs:188-249), one pair: ONE item per wave.  Lanes 0-31 evaluate s I - c O (== Ok, :229-232),
// lanes 32-63 s G + sb B - c Yb (== R, :238-245): five terms, each split into its two endomorphism halves, one half per quad.
template <class S>
__global__ void __launch_bounds__(64)
k_ped_verify_wave(BatchDev b, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  constexpr int NBITS = S::HAS_GLV ? 128 : Fr::BITS;
  const uint32_t lane = threadIdx.x & 63, h = lane >> 5, gl = lane & 31, q = gl >> 2, jc = gl & 3, tq = q >> 1, half = q & 1;
  const uint32_t j = b.first + blockIdx.x;
  const uint32_t io0 = b.io_off[j], ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pr = b.proofs + 256 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, 1, b.ads + ad0, adl, &f);
  const fp ybx = fp_load_le(pr), yby = fp_load_le(pr + 32), rx = fp_load_le(pr + 64), ry = fp_load_le(pr + 96);
  const fp okx = fp_load_le(pr + 128), oky = fp_load_le(pr + 160), s = fp_load_le(pr + 192), sb = fp_load_le(pr + 224);
  f |= point_flags<S>(ybx, yby);                                               // Yb == 0 rejected, pedersen.rs:204-206
  f |= (point_flags<S>(rx, ry) | point_flags<S>(okx, oky)) & FLAG_RANGE;
  if (ge_p<Fr>(s) || ge_p<Fr>(sb)) f |= FLAG_SCALAR;                           // (reported below: all lanes stay in step)
  if constexpr (S::SW_CODEC) {                                                 // three 33-byte SW forms, one inversion
    const fp xs[3] = {ybx, rx, okx}, ys[3] = {yby, ry, oky};
    sw_enc enc[3]; sw_encode_te_many<S, 3>(xs, ys, enc);
    absorb_sw_enc(t, enc[0]); tr_byte(t, DS_CHALLENGE); absorb_sw_enc(t, enc[1]); absorb_sw_enc(t, enc[2]);
  } else {
    absorb_point_xy<S>(t, ybx, yby);                                           // :219
    tr_byte(t, DS_CHALLENGE); absorb_point_xy<S>(t, rx, ry); absorb_point_xy<S>(t, okx, oky);
  }
  const fp c = challenge_finish(t);                                            // :222
  // this quad's term: lower half  (I, s), (-O, c);  upper half  (G, s), (B, sb), (-Yb, c);  the other quads carry a zero scalar
  const bool from_mem = h == 0 ? tq < 2 : tq == 2;
  const uint8_t *src = h == 0 ? (tq == 1 ? ios + 64 : ios) : pr;
  const te_pre cst = tq == 1 ? b_pre<S>() : g_pre<S>();
  const fp px = from_mem ? fp_to_mont<Fq>(fp_load_le(src)) : cst.x, py = from_mem ? fp_to_mont<Fq>(fp_load_le(src + 32)) : cst.y;
  const bool used = h == 0 ? tq < 2 : tq < 3, neg = h == 0 ? tq == 1 : tq == 2;
  const fp k = !used ? fp_zero() : (neg ? c : (h == 1 && tq == 1) ? sb : s);
  const WaveTerm<S> w = wave_term<S>(px, py, k, neg, half, jc);
  fp v = q_smul<S, NBITS>(w.coord, w.k, jc);
  v = wave_group_sum<S>(v, jc);
  const fp X = qperm<0, 0, 0, 0>(v), Y = qperm<1, 1, 1, 1>(v), Z = qperm<3, 3, 3, 3>(v);
  const fp tx = h == 0 ? okx : rx, ty = h == 0 ? oky : ry;
  uint32_t eq = (fp_eq(X, fp_mul<Fq>(fp_to_mont<Fq>(tx), Z)) && fp_eq(Y, fp_mul<Fq>(fp_to_mont<Fq>(ty), Z))) ? 1u : 0u;
  eq &= __shfl_xor(eq, 32);                                                    // both equations
  uint32_t bad = (w.ok || !used) ? 0u : 1u;
  for (int off = 32; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
  if (lane == 0) status[j] = f ? 2 : bad ? (int32_t)AVRF_WAVE_FALLBACK : eq ? 0 : 1;
}

// k P from the context's fixed-base table (proto_dev.h te_smul_fixed) in the quad form: 32 mixed additions of two rounds each
template <class S> AVRF_DI fp q_smul_fixed(const te_pre *tab, int base, const fp &k, uint32_t jc) {
  using Fq = typename S::Fq;
  const te_pre *t = tab + (size_t)base * 32 * 256;
  fp acc = q_identity<S>(jc);
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    const uint32_t d = (k.v[w >> 2] >> (8 * (w & 3))) & 255u;
    te_pre e; e.x = fp_zero(); e.y = fp_one<Fq>(); e.k = fp_zero();           // digit 0: the identity (an addition of it is harmless)
    if (d) e = load_pre(t + w * 256 + d);
    acc = q_madd<S>(acc, e.x, e.y, e.k, jc);
  }
  return acc;
}

// pedersen::Prover::prove (pedersen.rs:136-186), one pair: an item on 32 lanes.  Yb = pk + bl B, which the nonces wait for, is
// computed by EVERY quad of the item alike from the context's fixed-base table (32 two-round mixed additions, no divergence);
// R = k G + kb B and Ok = k I then share one doubling chain, a term's endomorphism half per quad.  The two normalisations use the
// binary-Euclid inversion (fp256.h fp_inv_few).
template <class S>
__global__ void __launch_bounds__(64)
k_ped_prove_wave(BatchDev b, uint8_t *__restrict__ proofs_out, uint8_t *__restrict__ blindings_out, uint32_t *__restrict__ flags, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  constexpr int NBITS = S::HAS_GLV ? 128 : Fr::BITS;
  const uint32_t lane = threadIdx.x & 63, gl = lane & 31, q = gl >> 2, jc = gl & 3, tq = q >> 1, half = q & 1;
  uint32_t j = b.first + 2 * blockIdx.x + (lane >> 5);
  const bool live = j < b.n;
  if (!live) j = b.n - 1;
  const uint32_t io0 = b.io_off[j], ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0;
  const fp sk = fp_load_le(b.sks + 32 * (size_t)j);
  uint32_t f = ge_p<Fr>(sk) ? FLAG_SCALAR : 0, pf = 0;
  suite_tr<S> t;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, 1, b.ads + ad0, adl, &pf);   // pedersen.rs:142
  f |= pf & FLAG_RANGE;
  suite_tr<S> tb = t; tr_byte(tb, DS_PEDERSEN_BLINDING);
  const fp bl = nonce<S>(sk, tb);                                               // pedersen.rs:51-54,145
  const fp bl_plain = fp_from_mont<Fr>(bl);
  const te_pre pkp = pre_from_xy<S>(b.pks_xy + 64 * (size_t)j);
  fp yq = q_smul_fixed<S>(b.fixed, FIXED_B, bl_plain, jc);                      // :148-149
  yq = q_madd<S>(yq, pkp.x, pkp.y, pkp.k, jc);
  te_aff yb;
  { const fp zi = fp_inv_few<Fq>(qperm<3, 3, 3, 3>(yq)); yb.x = fp_mul<Fq>(qperm<0, 0, 0, 0>(yq), zi); yb.y = fp_mul<Fq>(qperm<1, 1, 1, 1>(yq), zi); }
  absorb_point_mont<S>(t, yb);                                                  // :152
  const fp k = nonce<S>(sk, t), kb = nonce<S>(bl_plain, t);                     // :155-156
  const fp k_plain = fp_from_mont<Fr>(k);
  // R = k G + kb B (:159-161) and Ok = k I (:164) on ONE doubling chain: term pair 0 = (I, k), 1 = (G, k), 2 = (B, kb), each as its
  // two endomorphism halves; pair 3 carries a zero scalar.  Two butterfly steps then leave Ok in quads 0-1 and R in quads 2-5
  // (quads 0-1 and 6-7 only meet pair 3's identity in the second step).
  const te_pre cst = tq == 2 ? b_pre<S>() : g_pre<S>();
  const fp px = tq == 0 ? fp_to_mont<Fq>(fp_load_le(ios)) : cst.x, py = tq == 0 ? fp_to_mont<Fq>(fp_load_le(ios + 32)) : cst.y;
  const fp kt = tq == 2 ? fp_from_mont<Fr>(kb) : tq == 3 ? fp_zero() : k_plain;
  const WaveTerm<S> w = wave_term<S>(px, py, kt, false, half, jc);
  fp sq = q_smul<S, NBITS>(w.coord, w.k, jc);
  sq = q_add<S>(sq, fp_shfl_xor(sq, 4), jc);
  sq = q_add<S>(sq, fp_shfl_xor(sq, 24), jc);
  fp oq, rq;                                                                    // every lane of the item: its coordinate of Ok and of R
  {
    const int base_lane = (int)(lane & 32u) + (int)jc;
#pragma unroll
    for (int i = 0; i < 8; i++) { oq.v[i] = __shfl(sq.v[i], base_lane); rq.v[i] = __shfl(sq.v[i], base_lane + 8); }
  }
  te_aff ra, oka;                                                               // :166-167, one inversion for both
  {
    const fp rz = qperm<3, 3, 3, 3>(rq), oz = qperm<3, 3, 3, 3>(oq);
    const fp inv = fp_inv_few<Fq>(fp_mul<Fq>(rz, oz)), ri = fp_mul<Fq>(inv, oz), oi = fp_mul<Fq>(inv, rz);
    ra.x = fp_mul<Fq>(qperm<0, 0, 0, 0>(rq), ri); ra.y = fp_mul<Fq>(qperm<1, 1, 1, 1>(rq), ri);
    oka.x = fp_mul<Fq>(qperm<0, 0, 0, 0>(oq), oi); oka.y = fp_mul<Fq>(qperm<1, 1, 1, 1>(oq), oi);
  }
  suite_tr<S> tc = t; tr_byte(tc, DS_CHALLENGE); absorb_point_mont<S>(tc, ra); absorb_point_mont<S>(tc, oka);
  const fp c = fp_to_mont<Fr>(challenge_finish(tc));                            // :170
  const fp s = fp_add<Fr>(k, fp_mul<Fr>(c, fp_to_mont<Fr>(sk)));                // :173
  const fp sbv = fp_add<Fr>(kb, fp_mul<Fr>(c, bl));                             // :175
  uint32_t bad = (w.ok || tq != 0) ? 0u : 1u;
  for (int off = 16; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
  if (live && gl == 0) {
    status[j] = bad ? (int32_t)AVRF_WAVE_FALLBACK : 0;
    uint8_t *o = proofs_out + 256 * (size_t)j;
    store_xy<S>(o, yb); store_xy<S>(o + 64, ra); store_xy<S>(o + 128, oka);
    fp_store_le(o + 192, fp_from_mont<Fr>(s)); fp_store_le(o + 224, fp_from_mont<Fr>(sbv));
    if (blindings_out) fp_store_le(blindings_out + 32 * (size_t)j, bl_plain);
    if (f) atomicOr(flags, f);
  }
}

// ---------------------------------------------------------------- point codecs

// CanonicalDeserialize for TE affine points, compressed form: y = LE32 with the
// top bit cleared, x recovered from x^2 = (1 - y^2) / (a - d y^2), sign chosen by the flag.
// validate: additionally require the prime-order subgroup and non-identity (src/lib.rs:410-433).
// ---------------------------------------------------------------- hash to curve (Input::new, src/lib.rs -> Suite::data_to_point)

// absorb a digest (eight big-endian words) byte for byte
AVRF_DI void sha512_digest(Sha512 &s, const uint64_t (&d)[8]) {
  for (int i = 0; i < 8; i++) for (int k = 7; k >= 0; k--) sha512_byte(s, (uint8_t)(d[i] >> (8 * k)));
}
// big-endian integer of six consecutive digest words -> (low 256 bits, high 128 bits) as plain little-endian limbs
AVRF_DI void be384_split(const uint64_t (&w)[6], fp &lo, fp &hi) {
  hi = fp_zero();
  for (int i = 0; i < 4; i++) { lo.v[2 * i] = (uint32_t)w[5 - i]; lo.v[2 * i + 1] = (uint32_t)(w[5 - i] >> 32); }
  hi.v[0] = (uint32_t)w[1]; hi.v[1] = (uint32_t)(w[1] >> 32); hi.v[2] = (uint32_t)w[0]; hi.v[3] = (uint32_t)(w[0] >> 32);
}
// ark-ec Elligator2Map::map_to_curve for the Montgomery model (J, K) of the curve, Z = 5; TE point out
// (src/utils/hash_to_curve.rs:66-100)
template <class S> AVRF_DN te_aff ell2_map(fp u) {
  using Fq = typename S::Fq;
  const fp one = fp_one<Fq>(), jk = fp_const<Fq>(S::ELL2_JK), k2i = fp_const<Fq>(S::ELL2_KINV2), K = fp_const<Fq>(S::ELL2_K);
  fp u2 = fp_sqr<Fq>(u), t0 = fp_add<Fq>(fp_dbl<Fq>(fp_dbl<Fq>(u2)), u2);   // Z u^2, Z = 5
  fp den = fp_add<Fq>(one, t0);
  if (fp_is_zero(den)) den = one;
  fp x1 = fp_neg<Fq>(fp_mul<Fq>(jk, fp_inv<Fq>(den)));
  auto g = [&](const fp &x) { fp x2 = fp_sqr<Fq>(x); return fp_add<Fq>(fp_add<Fq>(fp_mul<Fq>(x2, x), fp_mul<Fq>(x2, jk)), fp_mul<Fq>(x, k2i)); };
  fp xs = x1, ys; bool want_odd = true;
  if (!fp_sqrt_nf<Fq>(g(x1), &ys)) {
    xs = fp_sub<Fq>(fp_neg<Fq>(x1), jk); want_odd = false;
    (void)fp_sqrt_nf<Fq>(g(xs), &ys);
  }
  if (((fp_from_mont<Fq>(ys).v[0] & 1u) != 0) != want_odd) ys = fp_neg<Fq>(ys);
  fp sx = fp_mul<Fq>(xs, K), ty = fp_mul<Fq>(ys, K), sp1 = fp_add<Fq>(sx, one);
  fp d2 = fp_mul<Fq>(ty, sp1);
  te_aff o;
  if (fp_is_zero(d2)) { o.x = fp_zero(); o.y = one; return o; }
  fp di = fp_inv<Fq>(d2);                                                   // 1 / (t (s + 1))
  o.x = fp_mul<Fq>(sx, fp_mul<Fq>(di, sp1));                                // s / t
  o.y = fp_mul<Fq>(fp_sub<Fq>(sx, one), fp_mul<Fq>(di, ty));                // (s - 1) / (s + 1)
  return o;
}

// one lane per message: out_xy[j] = hash_to_curve(data[off[j] .. off[j+1])), status 2 if try-and-increment finds nothing
template <class S>
__global__ void __launch_bounds__(128)
k_hash_to_curve(const uint8_t *__restrict__ data, const uint32_t *__restrict__ off, uint32_t n, uint8_t *__restrict__ out_xy, int32_t *__restrict__ status) {
  using Fq = typename S::Fq;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint8_t *msg = data + off[j]; const uint32_t len = off[j + 1] - off[j];
  constexpr uint8_t DS_H2C = 0x60;
  te_ext acc; bool ok = false;
  if constexpr (S::H2C_ELL2 && S::XOF_SHAKE) {
    // XofFieldHasher (src/utils/hash_to_curve.rs:103-150), RFC 9380 expand_message_xof with SHAKE128:
    // uniform = SHAKE128(msg || I2OSP(96, 2) || DST || I2OSP(len(DST), 1))[0..96], DST = suite id || 0x60;
    // two field elements of 48 big-endian bytes each
    Shake128 h; tr_init(h);
    tr_bytes(h, msg, len);
    tr_byte(h, 0); tr_byte(h, 96);
    for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(h, S::SUITE_ID[i]);
    tr_byte(h, DS_H2C); tr_byte(h, (uint8_t)(S::SUITE_ID_LEN + 1));
    ShakeReader rd = tr_reader(h);
    uint64_t w[12];
    for (int i = 0; i < 12; i++) w[i] = __builtin_bswap64(rd_word(rd, (uint32_t)i));   // stream bytes as big-endian words
    uint64_t w0[6] = {w[0], w[1], w[2], w[3], w[4], w[5]}, w1[6] = {w[6], w[7], w[8], w[9], w[10], w[11]};
    fp lo, hi;
    be384_split(w0, lo, hi); fp u0 = fp_from_wide_mont<Fq>(lo, hi);
    be384_split(w1, lo, hi); fp u1 = fp_from_wide_mont<Fq>(lo, hi);
    te_aff q0 = ell2_map<S>(u0), q1 = ell2_map<S>(u1);
    acc = te_madd<S>(te_from_pre<S>(te_make_pre<S>(q0.x, q0.y)), te_make_pre<S>(q1.x, q1.y));
    for (int c = S::COFACTOR; c > 1; c >>= 1) acc = te_dbl<S>(acc);
    ok = true;
  } else if (S::H2C_ELL2) {
    // RFC 9380 expand_message_xmd(SHA-512) as instantiated by ark-ff's DefaultFieldHasher: Z_pad = 48 zero bytes,
    // DST = suite id || 0x60, 96 uniform bytes -> two field elements of 48 big-endian bytes each
    Sha512 h; uint64_t b0[8], b1[8], b2[8];
    auto dst = [&](Sha512 &s) { for (int i = 0; i < S::SUITE_ID_LEN; i++) sha512_byte(s, S::SUITE_ID[i]); sha512_byte(s, DS_H2C); sha512_byte(s, (uint8_t)(S::SUITE_ID_LEN + 1)); };
    sha512_init(h);
    for (int i = 0; i < 48; i++) sha512_byte(h, 0);
    sha512_bytes(h, msg, len);
    sha512_byte(h, 0); sha512_byte(h, 96); sha512_byte(h, 0); dst(h);
    sha512_final(h, b0);
    sha512_init(h); sha512_digest(h, b0); sha512_byte(h, 1); dst(h); sha512_final(h, b1);
    uint64_t x[8]; for (int i = 0; i < 8; i++) x[i] = b0[i] ^ b1[i];
    sha512_init(h); sha512_digest(h, x); sha512_byte(h, 2); dst(h); sha512_final(h, b2);
    uint64_t w0[6] = {b1[0], b1[1], b1[2], b1[3], b1[4], b1[5]}, w1[6] = {b1[6], b1[7], b2[0], b2[1], b2[2], b2[3]};
    fp lo, hi;
    be384_split(w0, lo, hi); fp u0 = fp_from_wide_mont<Fq>(lo, hi);
    be384_split(w1, lo, hi); fp u1 = fp_from_wide_mont<Fq>(lo, hi);
    te_aff q0 = ell2_map<S>(u0), q1 = ell2_map<S>(u1);
    acc = te_madd<S>(te_from_pre<S>(te_make_pre<S>(q0.x, q0.y)), te_make_pre<S>(q1.x, q1.y));
    for (int c = S::COFACTOR; c > 1; c >>= 1) acc = te_dbl<S>(acc);
    ok = true;
  } else {
    // try-and-increment (src/utils/hash_to_curve.rs:34-57): transcript(suite id, 0x60, LE64(len), data, ctr) -> 32 bytes
    // -> Affine::from_random_bytes (top bit = sign of x, bits above the modulus size cleared)
    suite_tr<S> pre; tr_init(pre);
    for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(pre, S::SUITE_ID[i]);
    tr_byte(pre, DS_H2C); tr_u64le(pre, (uint64_t)len); tr_bytes(pre, msg, len);
    for (int ctr = 0; ctr <= 255 && !ok; ctr++) {
      suite_tr<S> t = pre; tr_byte(t, (uint8_t)ctr);
      auto rd = tr_reader(t);
      fp y; uint32_t w4[4];
      rd_chunk16(rd, 0, w4); for (int i = 0; i < 4; i++) y.v[i] = w4[i];
      rd_chunk16(rd, 1, w4); for (int i = 0; i < 4; i++) y.v[4 + i] = w4[i];
      const bool neg = (y.v[7] >> 31) != 0;
      y.v[7] &= 0xffffffffu >> (256 - Fq::BITS);
      if (ge_p<Fq>(y)) continue;
      if constexpr (S::SW_CODEC) {
        // SWAffine::from_random_bytes on the 32 squeezed bytes: they are the x-coordinate; the flag byte of the 33-byte buffer
        // is absent, i.e. zero, and the root taken is the LARGER one (pinned by the alpha -> h entries of the reference's
        // bandersnatch_sw vectors); then the map to the twisted-Edwards model, cofactor clearing there
        fp xm, ym;
        if (!sw_decode_te<S>(y, true, xm, ym)) continue;
        acc = te_from_pre<S>(te_make_pre<S>(xm, ym));
        for (int c = S::COFACTOR; c > 1; c >>= 1) acc = te_dbl<S>(acc);
        if (te_is_identity<S>(acc)) continue;
        ok = true;
        continue;
      }
      fp ym = fp_to_mont<Fq>(y), y2 = fp_sqr<Fq>(ym), one = fp_one<Fq>();
      fp a_const = mul_a<S>(one);   // the curve coefficient a (1, -5 or -1)
      fp den = fp_sub<Fq>(a_const, fp_mul<Fq>(fp_const<Fq>(S::D), y2)), xm;
      if (fp_is_zero(den) || !fp_sqrt_nf<Fq>(fp_mul<Fq>(fp_sub<Fq>(one, y2), fp_inv<Fq>(den)), &xm)) continue;
      if (fp_is_negative_mont<Fq>(xm) != neg) xm = fp_neg<Fq>(xm);
      if (fp_is_zero(xm) && neg) continue;
      acc = te_from_pre<S>(te_make_pre<S>(xm, ym));
      for (int c = S::COFACTOR; c > 1; c >>= 1) acc = te_dbl<S>(acc);
      if (te_is_identity<S>(acc)) continue;
      ok = true;
    }
  }
  if (ok) { store_xy<S>(out_xy + 64 * (size_t)j, te_to_aff<S>(acc)); status[j] = 0; }
  else { for (int i = 0; i < 64; i++) out_xy[64 * (size_t)j + i] = 0; status[j] = 2; }
}

// decode point j of `in`: Montgomery (xm, ym), canonical (x, y) for the output, status 0 / 2; *check = the subgroup test is still due
template <class S> AVRF_DI int32_t decode_point_at(const uint8_t *__restrict__ src, int validate, fp &xm, fp &ym, fp &x, fp &y, bool &check);
template <class S> AVRF_DI int32_t decode_point(const uint8_t *__restrict__ in, uint32_t j, int validate, fp &xm, fp &ym, fp &x, fp &y, bool &check) {
  return decode_point_at<S>(in + (size_t)S::POINT_LEN * j, validate, xm, ym, x, y, check);
}
template <class S> AVRF_DI int32_t decode_point_at(const uint8_t *__restrict__ src, int validate, fp &xm, fp &ym, fp &x, fp &y, bool &check) {
  using Fq = typename S::Fq;
  check = false;
  if constexpr (S::SW_CODEC) {
    // SWAffine::deserialize_compressed (33 bytes; unused flag bits and the infinity flag are rejected: the point at infinity has
    // no twisted-Edwards image, sw_to_te -> None, and the reference's verifiers refuse the identity anyway)
    fp xs;
    for (int i = 0; i < 8; i++) xs.v[i] = (uint32_t)src[4 * i] | ((uint32_t)src[4 * i + 1] << 8) | ((uint32_t)src[4 * i + 2] << 16) | ((uint32_t)src[4 * i + 3] << 24);
    const uint8_t flag = src[32];
    const fp idy = S::SW_NATIVE ? fp_zero() : fp_one<Fq>();      // y of the identity in the xy flavour
    int32_t st = 0; xm = fp_zero(); ym = idy;
    if ((flag & 0x7f) || ge_p<Fq>(xs) || !sw_decode_te<S>(xs, (flag & 0x80) != 0, xm, ym)) { st = 2; xm = fp_zero(); ym = idy; }
    else check = validate != 0;
    x = fp_from_mont<Fq>(xm); y = fp_from_mont<Fq>(ym);
    return st;
  }
  y = fp_load_le(src);
  bool neg = (y.v[7] >> 31) != 0; y.v[7] &= 0x7fffffffu;
  int32_t st = 0;
  x = fp_zero(); xm = fp_zero(); ym = fp_zero();
  if (ge_p<Fq>(y)) st = 2;
  else {
    ym = fp_to_mont<Fq>(y);
    fp y2 = fp_sqr<Fq>(ym), one = fp_one<Fq>();
    fp num = fp_sub<Fq>(one, y2);
    fp a_const = mul_a<S>(one);   // the curve coefficient a (1, -5 or -1)
    fp den = fp_sub<Fq>(a_const, fp_mul<Fq>(fp_const<Fq>(S::D), y2));
    bool sq = false;                                                       // x^2 = (1 - y^2) / (a - d y^2): no inversion, no data-dependent loop
    if (!fp_is_zero(den)) {
#ifndef AVRF_NO_FPU_SQRT
      if constexpr (FuAsm<Fq>::value) sq = fu_sqrt_ratio_nf<Fq>(num, den, &xm);   // the same on unsaturated limbs (fpu_sqrt.h)
      else
#endif
        sq = fp_sqrt_ratio_nf<Fq>(num, den, &xm);
    }
    if (!sq) st = 2;
    else {
      if (fp_is_negative_mont<Fq>(xm) != neg) xm = fp_neg<Fq>(xm);
      if (fp_is_zero(xm) && neg) st = 2;
      x = fp_from_mont<Fq>(xm);
      if (!st && validate) {
        fp onep = fp_zero(); onep.v[0] = 1;
        if (fp_is_zero(x) && fp_eq(y, onep)) st = 2;                       // identity
        else check = true;
      }
    }
  }
  return st;
}
template <class S>
__global__ void __launch_bounds__(128)
k_decompress(const uint8_t *__restrict__ in, uint32_t n, uint8_t *__restrict__ out_xy, int validate, int32_t *__restrict__ status) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fp xm, ym, x, y; bool check;
  int32_t st = decode_point<S>(in, j, validate, xm, ym, x, y, check);
  if (check) {
    if constexpr (S::HAS_2DESCENT) { if (!te_in_subgroup_2descent<S>(ym)) st = 2; }      // two Jacobi symbols (glv.h)
    else {
      te_ext rp = te_smul<S>(te_make_pre<S>(xm, ym), fp_const<Fr>(Fr::P), Fr::BITS);   // r * P == 0
      if (!te_is_identity<S>(rp)) st = 2;
    }
  }
  fp_store_le(out_xy + 64 * (size_t)j, x); fp_store_le(out_xy + 64 * (size_t)j + 32, y);
  status[j] = st;
}
// The same decoding for points that sit INSIDE wire records (a proof's R at the head of `R || s`) and go INTO staged records (the
// x || y proof layout): point j is read at in + j * in_stride and written at out + j * out_stride; a point that does not decode --
// or, with validate, is the identity or outside the prime-order subgroup -- raises FLAG_CURVE in *flags (InvalidData for the batch).
// This is what lets the wire flavour of the batch verifiers stage straight into a context's device buffers (capi.hip ctx_stage_wire).
template <class S>
__global__ void __launch_bounds__(128)
k_decompress_strided(const uint8_t *__restrict__ in, uint32_t in_stride, uint32_t n, uint8_t *__restrict__ out_xy, uint32_t out_stride, int validate,
                     uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fp xm, ym, x, y; bool check;
  int32_t st = decode_point_at<S>(in + (size_t)in_stride * j, validate, xm, ym, x, y, check);
  if (check) {
    if constexpr (S::HAS_2DESCENT) { if (!te_in_subgroup_2descent<S>(ym)) st = 2; }
    else {
      te_ext rp = te_smul<S>(te_make_pre<S>(xm, ym), fp_const<Fr>(Fr::P), Fr::BITS);
      if (!te_is_identity<S>(rp)) st = 2;
    }
  }
  fp_store_le(out_xy + (size_t)out_stride * j, x); fp_store_le(out_xy + (size_t)out_stride * j + 32, y);
  if (st) atomicOr(flags, (uint32_t)FLAG_CURVE);
}
// few points with the subgroup test: FOUR lanes per point -- the decoding is computed alike by the four lanes, r P runs as one
// quad scalar multiplication (te_quad.h: 85 windows of three doublings and an addition, 0.6 ms instead of 1.7 ms on one lane)
template <class S>
__global__ void __launch_bounds__(64)
k_decompress_wave(const uint8_t *__restrict__ in, uint32_t n, uint8_t *__restrict__ out_xy, int32_t *__restrict__ status) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  const uint32_t jc = threadIdx.x & 3;
  if constexpr (S::HAS_GLV) {
    // With the endomorphism the test is  a2 P + b2 psi(P) == 0  for the lattice vector (a2, b2), a2 + b2 lambda = 0 mod r, both
    // below 2^127: two quads per point run 43 windows instead of one quad running 85.  WHICH basis vector matters: on a point
    // S + T (S of order r, T in the 2-torsion: (0, -1) and the two points at infinity of this model) the sum is a2 T + b2 psi(T),
    // and (a2, b2) = (3, 0) mod 4 leaves T itself, while (a1, b1) = (0, 1) mod 4 would ACCEPT the coset of (0, -1)
    // (checked over all cosets by tests/test_gpu_wire.py::test_torsion_points; a point at infinity has Z = 0 != Y).
    const uint32_t h = (threadIdx.x >> 2) & 1u;
    uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool live = j < n;
    if (!live) j = n - 1;
    fp xm, ym, x, y; bool check;
    int32_t st = decode_point<S>(in, j, 1, xm, ym, x, y, check);
    if (!check) { xm = fp_zero(); ym = fp_one<Fq>(); }
    te_pre p; p.x = xm; p.y = ym; p.k = fp_zero();
    te_ext e; e.x = xm; e.y = ym; e.t = fp_mul<Fq>(xm, ym); e.z = fp_one<Fq>();
    te_ext q;
    const bool endo_ok = te_endo<S>(p, q);                                      // false: x y = 0 (order <= 2) or y^2 = b
    if (__any(check && !endo_ok)) {                                             // (rare, uniform: the whole wave takes the plain r P test)
      const fp coord = jc == 0 ? e.x : jc == 1 ? e.y : jc == 2 ? e.t : e.z;
      const fp v = q_smul<S, Fr::BITS>(coord, fp_const<Fr>(Fr::P), jc);
      const fp X = qperm<0, 0, 0, 0>(v), Y = qperm<1, 1, 1, 1>(v), Z = qperm<3, 3, 3, 3>(v);
      if (check && !(fp_is_zero(X) && fp_eq(Y, Z))) st = 2;
    } else {
      if (h) e = q;
      fp k = fp_zero();
#pragma unroll
      for (int i = 0; i < 4; i++) k.v[i] = h ? S::GLV_B2[i] : S::GLV_A2[i];
      const fp coord = jc == 0 ? e.x : jc == 1 ? e.y : jc == 2 ? e.t : e.z;
      fp v = q_smul<S, 128>(coord, k, jc);
      v = q_add<S>(v, fp_shfl_xor(v, 4), jc);
      const fp X = qperm<0, 0, 0, 0>(v), Y = qperm<1, 1, 1, 1>(v), Z = qperm<3, 3, 3, 3>(v);
      if (check && !(fp_is_zero(X) && fp_eq(Y, Z))) st = 2;
    }
    if (live && jc == 0 && h == 0) { fp_store_le(out_xy + 64 * (size_t)j, x); fp_store_le(out_xy + 64 * (size_t)j + 32, y); status[j] = st; }
  } else {
  uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const bool live = j < n;
  if (!live) j = n - 1;
  fp xm, ym, x, y; bool check;
  int32_t st = decode_point<S>(in, j, 1, xm, ym, x, y, check);
  if (!check) { xm = fp_zero(); ym = fp_one<Fq>(); }                           // (all quads run the multiplication; an undecodable point runs it on the identity)
  const fp coord = jc == 0 ? xm : jc == 1 ? ym : jc == 2 ? fp_mul<Fq>(xm, ym) : fp_one<Fq>();
  const fp v = q_smul<S, Fr::BITS>(coord, fp_const<Fr>(Fr::P), jc);            // r * P == 0  <=>  X = 0 and Y = Z
  const fp X = qperm<0, 0, 0, 0>(v), Y = qperm<1, 1, 1, 1>(v), Z = qperm<3, 3, 3, 3>(v);
  if (check && !(fp_is_zero(X) && fp_eq(Y, Z))) st = 2;
  if (live && jc == 0) { fp_store_le(out_xy + 64 * (size_t)j, x); fp_store_le(out_xy + 64 * (size_t)j + 32, y); status[j] = st; }
  }
}
// Validate::Yes for points that cross the ABI as canonical x || y (the reference's typed points have passed
// CanonicalDeserialize / the checked constructors, src/lib.rs:410-433,471-494): point p of record j sits at
// base + j * stride + 64 p.  level 1: coordinates < q and on the curve; level 2: also in the prime-order subgroup (r P = 0).
// A failing point sets FLAG_CURVE in *flags and, when rec_status is given, rec_status[j] = 2 (InvalidData).
template <class S>
__global__ void __launch_bounds__(128)
k_validate_xy(const uint8_t *__restrict__ base, uint32_t stride, uint32_t ppr, uint32_t nrec, int level, uint32_t *__restrict__ flags,
              int32_t *__restrict__ rec_status, const uint32_t *__restrict__ item_off, uint32_t n_items) {
  using Fq = typename S::Fq; using Fr = typename S::Fr;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrec * ppr) return;
  uint32_t j = t / ppr; const uint32_t p = t - j * ppr;
  const uint8_t *src = base + (size_t)j * stride + 64 * (size_t)p;
  if (item_off) {                                   // records are I/O pairs of items with different pair counts: item of pair j
    uint32_t lo = 0, hi = n_items;                  // the last i with item_off[i] <= j (item_off: n_items + 1 exclusive prefix sums)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (item_off[mid] <= j) lo = mid; else hi = mid; }
    j = lo;
  }
  fp x = fp_load_le(src), y = fp_load_le(src + 32);
  bool bad = ge_p<Fq>(x) || ge_p<Fq>(y);
  if (!bad) {
    fp xm = fp_to_mont<Fq>(x), ym = fp_to_mont<Fq>(y);
    bad = !te_on_curve<S>(xm, ym);
    if (!bad && level >= 2) {
      if constexpr (S::HAS_2DESCENT) bad = !te_in_subgroup_2descent<S>(ym);
      else {
        te_ext rp = te_smul<S>(te_make_pre<S>(xm, ym), fp_const<Fr>(Fr::P), Fr::BITS);
        bad = !te_is_identity<S>(rp);
      }
    }
  }
  if (bad) { atomicOr(flags, (uint32_t)FLAG_CURVE); if (rec_status) rec_status[j] = 2; }
}

// Output::hash::<N> = Suite::point_to_hash (src/lib.rs:605-609, src/utils/common.rs:290-305, mul_by_cofactor = false):
// Transcript::new(SUITE_ID) || 0x20 || serialize_compressed(point), first `len` squeezed bytes (len <= 64).
template <class S>
__global__ void __launch_bounds__(128)
k_output_hash(const uint8_t *__restrict__ in_xy, uint32_t n, uint32_t len, uint8_t *__restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const fp x = fp_load_le(in_xy + 64 * (size_t)j), y = fp_load_le(in_xy + 64 * (size_t)j + 32);
  suite_tr<S> h; tr_init(h);
  for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(h, S::SUITE_ID[i]);
  tr_byte(h, DS_POINT_TO_HASH);
  absorb_point_xy<S>(h, x, y);
  fp lo, hi; squeeze64(h, lo, hi);
  uint8_t *o = out + (size_t)len * j;
  for (uint32_t i = 0; i < len; i++) { const fp &w = i < 32 ? lo : hi; o[i] = (uint8_t)(w.v[(i & 31) >> 2] >> (8 * (i & 3))); }
}
// Secret::from_seed (src/lib.rs:346-369): sk0 = seed mod r; scalar = nonce(sk0, Transcript(SUITE_ID || seed [|| cnt])) for the first
// cnt in 0..255 that gives a non-zero scalar (cnt is absorbed only when > 0).  Writes the canonical scalar; the public key is the
// caller's avrf_scalar_mul_base (Secret::from_scalar).
template <class S>
__global__ void __launch_bounds__(128)
k_secret_from_seed(const uint8_t *__restrict__ seeds, uint32_t n, uint8_t *__restrict__ sk_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint8_t *sd = seeds + 32 * (size_t)j;
  const fp raw = fp_load_le(sd);
  const fp sk0 = fp_from_mont<Fr>(fp_from_wide_mont<Fr>(raw, fp_zero()));      // from_le_bytes_mod_order(seed), canonical
  fp k = fp_zero(); bool ok = false;
  for (int cnt = 0; cnt < 256 && !ok; cnt++) {
    suite_tr<S> t; tr_init(t);
    for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(t, S::SUITE_ID[i]);
    absorb_fp_le(t, raw);
    if (cnt > 0) tr_byte(t, (uint8_t)cnt);
    k = nonce<S>(sk0, t);
    ok = !fp_is_zero(k);
  }
  if (!ok) atomicOr(flags, 1u);                                                 // (unreachable under standard assumptions, lib.rs:361-366)
  fp_store_le(sk_out + 32 * (size_t)j, fp_from_mont<Fr>(k));
}

template <class S>
__global__ void __launch_bounds__(256)
k_compress(const uint8_t *__restrict__ in_xy, uint32_t n, uint8_t *__restrict__ out) {
  using Fq = typename S::Fq;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fp x = fp_load_le(in_xy + 64 * (size_t)j), y = fp_load_le(in_xy + 64 * (size_t)j + 32);
  if constexpr (S::SW_CODEC) {                                        // 33-byte SW form of the suite's Affine (sw_map.h)
    const sw_enc e = sw_encode_te<S>(fp_to_mont<Fq>(x), fp_to_mont<Fq>(y));
    uint8_t *o = out + 33 * (size_t)j;
    for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) o[4 * i + b] = (uint8_t)(e.x.v[i] >> (8 * b));
    o[32] = e.flag;
    return;
  }
  if (fp_is_negative_plain<Fq>(x)) y.v[7] |= 0x80000000u;
  fp_store_le(out + 32 * (size_t)j, y);
}

// ---------------------------------------------------------------- launchers (per-suite unit)

template <class S> void SingleOps<S>::fixed_table(struct te_pre_raw *d_tab, hipStream_t st) {
  hipLaunchKernelGGL(k_fixed_table<S>, dim3((FIXED_TABLE_POINTS + 127) / 128), dim3(128), 0, st, (te_pre *)d_tab);
}
template <class S> void SingleOps<S>::smul(const uint8_t *d_scalars, const uint8_t *d_points_xy, uint32_t n, uint8_t *d_out, uint32_t *d_flags,
                                           const struct te_pre_raw *d_fixed, hipStream_t st) {
  hipLaunchKernelGGL(k_smul<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_scalars, d_points_xy, n, d_out, d_flags, (const te_pre *)d_fixed);
}
template <class S> void SingleOps<S>::thin_prove(const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny) {
  const dim3 g((b.n - b.first + 127) / 128), bl(128);
  if (tiny) hipLaunchKernelGGL((k_thin_prove<S, true>), g, bl, 0, st, b, d_proofs_out, d_flags);
  else hipLaunchKernelGGL((k_thin_prove<S, false>), g, bl, 0, st, b, d_proofs_out, d_flags);
}
template <class S> size_t SingleOps<S>::prove_state_bytes() { return (sizeof(ProveState<S>) + 63) / 64 * 64; }
template <class S> void SingleOps<S>::thin_prove_begin(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, hipStream_t st, bool tiny) {
  if (tiny) hipLaunchKernelGGL((k_thin_prove_begin<S, true>), dim3(1), dim3(64), 0, st, b, d_scalars, (te_pre *)d_pre, d_state);
  else hipLaunchKernelGGL((k_thin_prove_begin<S, false>), dim3(1), dim3(64), 0, st, b, d_scalars, (te_pre *)d_pre, d_state);
}
template <class S> void SingleOps<S>::thin_prove_end(const BatchDev &b, const uint8_t *d_state, const uint8_t *d_rxy, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny) {
  if (tiny) hipLaunchKernelGGL((k_thin_prove_end<S, true>), dim3(1), dim3(64), 0, st, b, d_state, d_rxy, d_proofs_out, d_flags);
  else hipLaunchKernelGGL((k_thin_prove_end<S, false>), dim3(1), dim3(64), 0, st, b, d_state, d_rxy, d_proofs_out, d_flags);
}
template <class S> size_t SingleOps<S>::ped_state_bytes() { return (sizeof(PedState<S>) + 63) / 64 * 64; }
template <class S> void SingleOps<S>::ped_prove_begin(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, uint32_t *d_wts, hipStream_t st) {
  hipLaunchKernelGGL(k_ped_prove_begin<S>, dim3(1), dim3(64), 0, st, b, d_scalars, (te_pre *)d_pre, d_state, d_wts);
}
template <class S> void SingleOps<S>::ped_prove_mid(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, const uint32_t *d_wts, const uint8_t *d_yb, hipStream_t st) {
  hipLaunchKernelGGL(k_ped_prove_mid<S>, dim3(1), dim3(64), 0, st, b, d_scalars, (te_pre *)d_pre, d_state, d_wts, d_yb);
}
template <class S> void SingleOps<S>::ped_prove_end(const BatchDev &b, const uint8_t *d_state, const uint8_t *d_pts, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st) {
  hipLaunchKernelGGL(k_ped_prove_end<S>, dim3(1), dim3(64), 0, st, b, d_state, d_pts, d_proofs_out, d_blind, d_flags);
}
template <class S> void SingleOps<S>::tiny_verify(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  hipLaunchKernelGGL(k_tiny_verify<S>, dim3((b.n - b.first + 127) / 128), dim3(128), 0, st, b, d_status);
}
template <class S> void SingleOps<S>::thin_verify(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  hipLaunchKernelGGL(k_thin_verify<S>, dim3((b.n - b.first + 127) / 128), dim3(128), 0, st, b, d_status);
}
template <class S> bool SingleOps<S>::thin_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if constexpr (S::SW_NATIVE) return false;
  else { hipLaunchKernelGGL(k_thin_verify_wave<S>, dim3((b.n - b.first + 1) / 2), dim3(64), 0, st, b, d_status); return true; }
}
template <class S> bool SingleOps<S>::thin_prove_wave(const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, int32_t *d_status, hipStream_t st, bool tiny) {
  if constexpr (S::SW_NATIVE) return false;
  else {
    if (tiny) hipLaunchKernelGGL((k_thin_prove_wave<S, true>), dim3((b.n - b.first + 1) / 2), dim3(64), 0, st, b, d_proofs_out, d_flags, d_status);
    else hipLaunchKernelGGL((k_thin_prove_wave<S, false>), dim3((b.n - b.first + 1) / 2), dim3(64), 0, st, b, d_proofs_out, d_flags, d_status);
    return true;
  }
}
template <class S> bool SingleOps<S>::tiny_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if constexpr (S::SW_NATIVE) return false;
  else { hipLaunchKernelGGL(k_tiny_verify_wave<S>, dim3((b.n - b.first + 1) / 2), dim3(64), 0, st, b, d_status); return true; }
}
template <class S> bool SingleOps<S>::ped_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if constexpr (S::SW_NATIVE) return false;
  else { hipLaunchKernelGGL(k_ped_verify_wave<S>, dim3(b.n - b.first), dim3(64), 0, st, b, d_status); return true; }
}
template <class S> bool SingleOps<S>::ped_prove_wave(const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, int32_t *d_status, hipStream_t st) {
  if constexpr (S::SW_NATIVE) return false;
  else { hipLaunchKernelGGL(k_ped_prove_wave<S>, dim3((b.n - b.first + 1) / 2), dim3(64), 0, st, b, d_proofs_out, d_blind, d_flags, d_status); return true; }
}
template <class S> void SingleOps<S>::ped_prove(const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st) {
  hipLaunchKernelGGL(k_ped_prove<S>, dim3((b.n - b.first + 127) / 128), dim3(128), 0, st, b, d_proofs_out, d_blind, d_flags);
}
template <class S> void SingleOps<S>::ped_verify(const BatchDev &b, int32_t *d_status, hipStream_t st) {
  hipLaunchKernelGGL(k_ped_verify<S>, dim3((b.n - b.first + 127) / 128), dim3(128), 0, st, b, d_status);
}
template <class S> void SingleOps<S>::hash_to_curve(const uint8_t *d_data, const uint32_t *d_off, uint32_t n, uint8_t *d_out, int32_t *d_status, hipStream_t st) {
  hipLaunchKernelGGL(k_hash_to_curve<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_data, d_off, n, d_out, d_status);
}
template <class S> void SingleOps<S>::decompress(const uint8_t *d_in, uint32_t n, uint8_t *d_out, int validate, int32_t *d_status, hipStream_t st) {
  // few points: four lanes per point for the subgroup test r P (where the test is two Jacobi symbols the lane-per-point kernel is the fast one at every n)
  if constexpr (!S::SW_NATIVE && !S::HAS_2DESCENT) if (validate && n <= 4096) {
    hipLaunchKernelGGL(k_decompress_wave<S>, dim3(((S::HAS_GLV ? 8 : 4) * n + 63) / 64), dim3(64), 0, st, d_in, n, d_out, d_status);   // (two quads per point with the endomorphism)
    return;
  }
  hipLaunchKernelGGL(k_decompress<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_in, n, d_out, validate, d_status);
}
template <class S> void SingleOps<S>::validate_xy(const uint8_t *d_base, uint32_t stride, uint32_t ppr, uint32_t nrec, int level, uint32_t *d_flags,
                                                  int32_t *d_rec_status, hipStream_t st, const uint32_t *d_item_off, uint32_t n_items) {
  const uint32_t tot = nrec * ppr;
  hipLaunchKernelGGL(k_validate_xy<S>, dim3((tot + 127) / 128), dim3(128), 0, st, d_base, stride, ppr, nrec, level, d_flags, d_rec_status, d_item_off, n_items);
}
template <class S> void SingleOps<S>::decompress_strided(const uint8_t *d_in, uint32_t in_stride, uint32_t n, uint8_t *d_out, uint32_t out_stride, int validate, uint32_t *d_flags, hipStream_t st) {
  hipLaunchKernelGGL(k_decompress_strided<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_in, in_stride, n, d_out, out_stride, validate, d_flags);
}
template <class S> void SingleOps<S>::compress(const uint8_t *d_in, uint32_t n, uint8_t *d_out, hipStream_t st) {
  hipLaunchKernelGGL(k_compress<S>, dim3((n + 255) / 256), dim3(256), 0, st, d_in, n, d_out);
}
template <class S> void SingleOps<S>::output_hash(const uint8_t *d_in, uint32_t n, uint32_t len, uint8_t *d_out, hipStream_t st) {
  hipLaunchKernelGGL(k_output_hash<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_in, n, len, d_out);
}
template <class S> void SingleOps<S>::secret_from_seed(const uint8_t *d_seeds, uint32_t n, uint8_t *d_sk, uint32_t *d_flags, hipStream_t st) {
  hipLaunchKernelGGL(k_secret_from_seed<S>, dim3((n + 127) / 128), dim3(128), 0, st, d_seeds, n, d_sk, d_flags);
}
template struct SingleOps<suite_by_id<AVRF_TU_SUITE>::type>;

}  // namespace avrf

#else   // ---------------------------------------------------------------- run-time dispatch unit

#define AVRF_SINGLE(suite, CALL) with_suite((suite), [&](auto tag_) { using S_ = typename decltype(tag_)::type; SingleOps<S_>::CALL; })

void launch_fixed_table(int suite, struct te_pre_raw *d_tab, hipStream_t st) { AVRF_SINGLE(suite, fixed_table(d_tab, st)); }
void launch_smul(int suite, const uint8_t *d_scalars, const uint8_t *d_points_xy, uint32_t n, uint8_t *d_out, uint32_t *d_flags,
                 const struct te_pre_raw *d_fixed, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, smul(d_scalars, d_points_xy, n, d_out, d_flags, d_fixed, st));
}
size_t thin_prove_state_bytes(int suite) {
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::prove_state_bytes(); });
}
void launch_thin_prove_begin(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, hipStream_t st, bool tiny) {
  AVRF_SINGLE(suite, thin_prove_begin(b, d_scalars, d_pre, d_state, st, tiny));
}
void launch_thin_prove_end(int suite, const BatchDev &b, const uint8_t *d_state, const uint8_t *d_rxy, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny) {
  AVRF_SINGLE(suite, thin_prove_end(b, d_state, d_rxy, d_proofs_out, d_flags, st, tiny));
}
size_t ped_prove_state_bytes(int suite) {
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::ped_state_bytes(); });
}
void launch_ped_prove_begin(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, uint32_t *d_wts, hipStream_t st) {
  AVRF_SINGLE(suite, ped_prove_begin(b, d_scalars, d_pre, d_state, d_wts, st));
}
void launch_ped_prove_mid(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, const uint32_t *d_wts, const uint8_t *d_yb, hipStream_t st) {
  AVRF_SINGLE(suite, ped_prove_mid(b, d_scalars, d_pre, d_state, d_wts, d_yb, st));
}
void launch_ped_prove_end(int suite, const BatchDev &b, const uint8_t *d_state, const uint8_t *d_pts, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st) {
  AVRF_SINGLE(suite, ped_prove_end(b, d_state, d_pts, d_proofs_out, d_blind, d_flags, st));
}
void launch_thin_prove(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny) {
  if (!b.n) return;
  AVRF_SINGLE(suite, thin_prove(b, d_proofs_out, d_flags, st, tiny));
}
void launch_tiny_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return;
  AVRF_SINGLE(suite, tiny_verify(b, d_status, st));
}
void launch_thin_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return;
  AVRF_SINGLE(suite, thin_verify(b, d_status, st));
}
bool launch_thin_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return false;
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::thin_verify_wave(b, d_status, st); });
}
bool launch_thin_prove_wave(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, int32_t *d_status, hipStream_t st, bool tiny) {
  if (!b.n) return false;
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::thin_prove_wave(b, d_proofs_out, d_flags, d_status, st, tiny); });
}
bool launch_tiny_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return false;
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::tiny_verify_wave(b, d_status, st); });
}
bool launch_ped_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return false;
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::ped_verify_wave(b, d_status, st); });
}
bool launch_ped_prove_wave(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, int32_t *d_status, hipStream_t st) {
  if (!b.n) return false;
  return with_suite(suite, [&](auto tag_) { using S_ = typename decltype(tag_)::type; return SingleOps<S_>::ped_prove_wave(b, d_proofs_out, d_blind, d_flags, d_status, st); });
}
void launch_ped_prove(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st) {
  if (!b.n) return;
  AVRF_SINGLE(suite, ped_prove(b, d_proofs_out, d_blind, d_flags, st));
}
void launch_ped_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st) {
  if (!b.n) return;
  AVRF_SINGLE(suite, ped_verify(b, d_status, st));
}
void launch_hash_to_curve(int suite, const uint8_t *d_data, const uint32_t *d_off, uint32_t n, uint8_t *d_out, int32_t *d_status, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, hash_to_curve(d_data, d_off, n, d_out, d_status, st));
}
void launch_decompress(int suite, const uint8_t *d_in, uint32_t n, uint8_t *d_out, int validate, int32_t *d_status, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, decompress(d_in, n, d_out, validate, d_status, st));
}
void launch_validate_xy(int suite, const uint8_t *d_base, uint32_t stride, uint32_t ppr, uint32_t nrec, int level, uint32_t *d_flags,
                        int32_t *d_rec_status, hipStream_t st, const uint32_t *d_item_off, uint32_t n_items) {
  if (!nrec || !ppr || level <= 0) return;
  AVRF_SINGLE(suite, validate_xy(d_base, stride, ppr, nrec, level, d_flags, d_rec_status, st, d_item_off, n_items));
}
void launch_decompress_strided(int suite, const uint8_t *d_in, uint32_t in_stride, uint32_t n, uint8_t *d_out, uint32_t out_stride, int validate, uint32_t *d_flags, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, decompress_strided(d_in, in_stride, n, d_out, out_stride, validate, d_flags, st));
}
void launch_compress(int suite, const uint8_t *d_in, uint32_t n, uint8_t *d_out, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, compress(d_in, n, d_out, st));
}
void launch_output_hash(int suite, const uint8_t *d_in, uint32_t n, uint32_t len, uint8_t *d_out, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, output_hash(d_in, n, len, d_out, st));
}
void launch_secret_from_seed(int suite, const uint8_t *d_seeds, uint32_t n, uint8_t *d_sk, uint32_t *d_flags, hipStream_t st) {
  if (!n) return;
  AVRF_SINGLE(suite, secret_from_seed(d_seeds, n, d_sk, d_flags, st));
}

}  // namespace avrf
#endif
