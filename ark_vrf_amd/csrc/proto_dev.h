// proto_dev.h -- device-side restatement of the reference's Fiat-Shamir protocol glue
// (src/utils/common.rs) on top of sha512_dev.h / fp256.h / te.h.  One lane = one VRF item.
//
//   vrf_transcript_base   common.rs:159-173   -> tr_base()
//   chain_ios             common.rs:231-240   -> schnorr argument of tr_base()
//   DelinearizeScalars    common.rs:335-369   -> delin_seed() + xof128()
//   challenge             common.rs:270-280   -> challenge_begin()/challenge_finish()
//   nonce                 common.rs:313-328   -> nonce()
//   challenge_scalar      common.rs:72-76, nonce_scalar :57-70
#pragma once
#include "sha512_dev.h"
#include "shake_dev.h"
#include "sha256_dev.h"
#include "te.h"
#include <type_traits>

// Everything that includes this header is a per-item protocol kernel: route the field / curve
// calls to their out-of-line forms (fp256.h, te.h) so that each kernel holds ONE copy of the
// Montgomery multiplier and of each point operation instead of hundreds of inlined ones.
// (The MSM hot loops in msm.hip do not include this header and stay fully inlined.)
#define fp_mul fp_mul_nf
#define fp_sqr fp_sqr_nf
#define fp_inv fp_inv_nf
#define fp_to_mont fp_to_mont_nf
#define fp_from_mont fp_from_mont_nf
#define fp_is_negative_mont fp_is_negative_mont_nf
#define fp_from_wide_mont fp_from_wide_mont_nf
#define te_to_aff te_to_aff_nf
#define te_make_pre te_make_pre_nf
#include "sw_map.h"      // (after the routing above: its maps use the out-of-line field operations too)
#include "fpu_te.h"      // unsaturated-limb running point of the scalar-multiplication chains

namespace avrf {

enum : uint8_t {
  DS_TINY = 0x00, DS_THIN = 0x01, DS_PEDERSEN = 0x02, DS_NONCE_EXPAND = 0x10, DS_NONCE = 0x11, DS_PEDERSEN_BLINDING = 0x12,
  DS_POINT_TO_HASH = 0x20, DS_DELINEARIZE = 0x30, DS_CHALLENGE = 0x40, DS_BATCH_VERIFY = 0x50
};
enum { FLAG_RANGE = 1, FLAG_IDENTITY = 2, FLAG_SCALAR = 4, FLAG_CURVE = 8 };

// A staged batch in HBM.  Offsets are exclusive prefix sums (n + 1 entries).
struct BatchDev {
  const uint8_t *pks_xy;    // n x 64 (thin only)
  const uint8_t *ios_xy;    // tot_io x 128 (input_xy || output_xy)
  const uint32_t *io_off;   // n + 1
  const uint8_t *ads;       // concatenated additional data
  const uint32_t *ad_off;   // n + 1
  const uint8_t *proofs;    // thin: n x 96 (R_xy || s); pedersen: n x 256
  const uint8_t *sks;       // n x 32 (provers only)
  uint32_t n;
  const te_pre *fixed;      // fixed-base tables of the suite's G and BLINDING_BASE: [2][32][256] (provers only)
  uint8_t *records;         // batch verifiers, counter-mode transcripts: the prepare kernel also writes item j's record of the weight
                            // transcript, c(16) || 0(16) || s(32) [|| sb(32)], to records + (64 | 96) j -- the host then hashes
                            // one contiguous buffer that came back in a single copy; nullptr: not wanted
  const uint8_t *weights;   // batch verifiers, sponge transcripts only: the squeezed weight stream of THIS batch (16 / 32 bytes per
                            // item), produced by the host -- a sponge's output is sequential; nullptr: counter-mode stream from the seed
  te_ext *tabs;             // per-item kernels: ITEM_TAB_SLOTS window-table entries per item of THIS LAUNCH, item j's at
                            // tabs + (j - first) * ITEM_TAB_SLOTS (te_smul_ws below)
  uint32_t first;           // per-item kernels: a launch covers items first .. n - 1 (the host walks a large call in chunks so
                            // that the table workspace stays bounded, capi.hip per_item_chunks)
};
struct Seed64 { uint64_t w[8]; };  // a SHA-512 digest as big-endian words
// Suite::Transcript (src/lib.rs:177-250): HashTranscript<Sha512>, or the SHAKE128 sponge for the suites that say so
template <class S> using suite_tr = std::conditional_t<S::XOF_SHAKE, Shake128, std::conditional_t<S::TR_SHA256, Sha256, Sha512>>;

// absorb the ark-serialize compressed encoding of an affine point given as canonical x||y
// (LE32 each): LE32(y) with bit 255 set iff x > (q-1)/2   
template <class S, class T> AVRF_DI void absorb_point_xy(T &h, const fp &x, const fp &y) {
  using Fq = typename S::Fq;
  if constexpr (S::SW_CODEC) {           // the suite's Affine is SWAffine: 33-byte form, LE32(x_sw) || flags (sw_map.h)
    absorb_sw_enc(h, sw_encode_te<S>(fp_to_mont<Fq>(x), fp_to_mont<Fq>(y)));
    return;
  }
  uint32_t sign = fp_is_negative_plain<Fq>(x) ? 0x80000000u : 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) tr_u32le(h, y.v[i] | (i == 7 ? sign : 0u));
}
template <class T> AVRF_DI void absorb_sw_enc(T &h, const sw_enc &e) {
#pragma unroll
  for (int i = 0; i < 8; i++) tr_u32le(h, e.x.v[i]);
  tr_byte(h, e.flag);
}
// the serialised generator (chain_ios, src/utils/common.rs:231-240), a per-suite constant
template <class S, class T> AVRF_DI void absorb_generator(T &h) {
  for (int i = 0; i < S::POINT_LEN; i++) tr_byte(h, S::G_ENC[i]);
}
template <class T> AVRF_DI void absorb_fp_le(T &h, const fp &a) {
#pragma unroll
  for (int i = 0; i < 8; i++) tr_u32le(h, a.v[i]);
}
template <class S> AVRF_DI uint32_t point_flags(const fp &x, const fp &y) {
  using Fq = typename S::Fq;
  uint32_t f = 0;
  if (ge_p<Fq>(x) || ge_p<Fq>(y)) f |= FLAG_RANGE;
  fp one = fp_zero(); one.v[0] = 1;
  if constexpr (S::SW_NATIVE) { if (fp_is_zero(x) && fp_is_zero(y)) f |= FLAG_IDENTITY; }   // (0, 0): the point at infinity
  else if (fp_is_zero(x) && fp_eq(y, one)) f |= FLAG_IDENTITY;
  return f;
}

// Transcript after vrf_transcript_base: SUITE_ID, scheme tag, io count + pairs (optionally the
// Schnorr pair (G, pk) first), ad length + ad.  Accumulates point flags of pk / ios into *flags.
// (the public key comes as two field elements in registers: a prover that derives it never spills it to a private byte array)
template <class S, class T> AVRF_DI void tr_base(T &h, uint8_t scheme, bool schnorr, const fp &pkx, const fp &pky,
                                        const uint8_t *ios_xy, uint32_t m, const uint8_t *ad, uint32_t adl, uint32_t *flags) {
  tr_init(h);
  for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(h, S::SUITE_ID[i]);
  tr_byte(h, scheme);
  tr_u64le(h, (uint64_t)m + (schnorr ? 1 : 0));
  uint32_t f = 0;
  if constexpr (S::SW_CODEC) if (m == 1) {
    // short-Weierstrass presentation, one pair: the 33-byte forms of (pk,) I, O with ONE inversion (sw_map.h)
    fp xs[3], ys[3]; sw_enc enc[3];
    xs[0] = fp_load_le(ios_xy); ys[0] = fp_load_le(ios_xy + 32); xs[1] = fp_load_le(ios_xy + 64); ys[1] = fp_load_le(ios_xy + 96);
    if (schnorr) { xs[2] = pkx; ys[2] = pky; } else { xs[2] = xs[0]; ys[2] = ys[0]; }
    sw_encode_te_many<S, 3>(xs, ys, enc);
    if (schnorr) { absorb_generator<S>(h); f |= point_flags<S>(xs[2], ys[2]); absorb_sw_enc(h, enc[2]); }
    f |= point_flags<S>(xs[0], ys[0]) | point_flags<S>(xs[1], ys[1]);
    absorb_sw_enc(h, enc[0]); absorb_sw_enc(h, enc[1]);
    tr_u64le(h, (uint64_t)adl);
    tr_bytes(h, ad, adl);
    *flags |= f;
    return;
  }
  if (schnorr) {
    absorb_generator<S>(h);
    f |= point_flags<S>(pkx, pky);
    absorb_point_xy<S>(h, pkx, pky);
  }
  for (uint32_t i = 0; i < m; i++) {
    const uint8_t *p = ios_xy + 128 * (size_t)i;
    fp x = fp_load_le(p), y = fp_load_le(p + 32);
    f |= point_flags<S>(x, y); absorb_point_xy<S>(h, x, y);
    x = fp_load_le(p + 64); y = fp_load_le(p + 96);
    f |= point_flags<S>(x, y); absorb_point_xy<S>(h, x, y);
  }
  tr_u64le(h, (uint64_t)adl);
  tr_bytes(h, ad, adl);
  *flags |= f;
}
template <class S, class T> AVRF_DI void tr_base(T &h, uint8_t scheme, bool schnorr, const uint8_t *pk_xy,
                                        const uint8_t *ios_xy, uint32_t m, const uint8_t *ad, uint32_t adl, uint32_t *flags) {
  fp x = fp_zero(), y = fp_zero();
  if (schnorr) { x = fp_load_le(pk_xy); y = fp_load_le(pk_xy + 32); }        // pk_xy: global memory (staged batch)
  tr_base<S>(h, scheme, schnorr, x, y, ios_xy, m, ad, adl, flags);
}
// reader of the delinearisation stream: fork + [0x30] + finalize (DelinearizeScalars, common.rs:335-369)
template <class T> AVRF_DI auto delin_seed(const T &h) {
  T hd = h; tr_byte(hd, DS_DELINEARIZE); return tr_reader(hd);
}
// i-th 16-byte chunk of a squeeze stream as a 128-bit plain integer (readers only move forward)
template <class R> AVRF_DI fp xof128(R &rd, uint32_t i) {
  uint32_t w[4]; rd_chunk16(rd, i, w);
  fp a = fp_zero(); a.v[0] = w[0]; a.v[1] = w[1]; a.v[2] = w[2]; a.v[3] = w[3]; return a;
}
// challenge_scalar of a finished transcript: first 16 squeezed bytes (plain 128-bit integer < r)
template <class T> AVRF_DI fp challenge_finish(const T &h) {
  auto rd = tr_reader(h); return xof128(rd, 0);
}
// the first 64 squeezed bytes as little-endian integers lo (bytes 0..31), hi (32..63)
template <class T> AVRF_DI void squeeze64(const T &h, fp &lo, fp &hi) {
  auto rd = tr_reader(h);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t w[4]; rd_chunk16(rd, (uint32_t)k, w);
    fp &dst = k < 2 ? lo : hi;
#pragma unroll
    for (int i = 0; i < 4; i++) dst.v[4 * (k & 1) + i] = w[i];
  }
}
// nonce(sk, transcript): returns the nonce in Montgomery form over Fr.  `sk_plain` canonical.
template <class S, class T> AVRF_DN fp nonce(fp sk_plain, T t) {
  using Fr = typename S::Fr;
  T te = t; tr_byte(te, DS_NONCE_EXPAND); absorb_fp_le(te, sk_plain);
  fp lo, hi;
  squeeze64(te, lo, hi);                            // sk_hash = 64 squeezed bytes
  T tn = t; tr_byte(tn, DS_NONCE);
  absorb_fp_le(tn, lo); absorb_fp_le(tn, hi);       // absorb the 64 bytes of sk_hash in order
  squeeze64(tn, lo, hi);
  // nonce_scalar: ceil((bits+128)/8) = 48 bytes for both suites (253 / 251 bits)
  static_assert((Fr::BITS + 128 + 7) / 8 == 48, "nonce length");
#pragma unroll
  for (int i = 4; i < 8; i++) hi.v[i] = 0;           // keep bytes 32..47 only
  return fp_from_wide_mont<Fr>(lo, hi);
}

// ---- group helpers for the per-item kernels

template <class S> AVRF_DI te_pre pre_from_xy(const uint8_t *xy) {
  using Fq = typename S::Fq;
  return te_make_pre<S>(fp_to_mont<Fq>(fp_load_le(xy)), fp_to_mont<Fq>(fp_load_le(xy + 32)));
}
// The running point of a scalar multiplication: on the twisted-Edwards suites whose field has the asm multiplier it lives in
// unsaturated limbs (fpu_te.h teu4: doubling 4M + 4S and the additions as asm blocks with limb-wise sums between them; the
// window-table entries stay saturated in memory and are sliced as they arrive), elsewhere (secp256r1: XYZZ coordinates behind the
// same types, a field that fills its top bit) in the saturated form.  The doublings of a window run as a loop of ONE inlined
// doubling: a chain function holds one doubling and one addition, ~30 KB of code.
template <class S> struct Chain {
#ifndef AVRF_NO_UNSAT_CHAINS
  static constexpr bool U = !S::SW_NATIVE && FuAsm<typename S::Fq>::value;
#else
  static constexpr bool U = false;
#endif
  struct sat_t { te_ext p; };
  using acc_t = std::conditional_t<U, teu4<S>, te_ext>;
  static AVRF_DI acc_t identity() { if constexpr (U) return teu4_identity<S>(); else return te_identity<S>(); }
  static AVRF_DI acc_t dbl(const acc_t &a) { if constexpr (U) return teu4_dbl<S>(a); else return te_dbl<S>(a); }
  template <int N> static AVRF_DI acc_t dbln(acc_t a) {
#pragma unroll 1
    for (int t = 0; t < N; t++) a = dbl(a);
    return a;
  }
  static AVRF_DI acc_t add(const acc_t &a, const te_ext &e) { if constexpr (U) return teu4_add_sat<S>(a, e); else return te_add<S>(a, e); }
  static AVRF_DI acc_t madd(const acc_t &a, const te_pre &q) { if constexpr (U) return teu4_madd_pre<S>(a, q); else return te_madd<S>(a, q); }
  static AVRF_DI te_ext finish(const acc_t &a) { if constexpr (U) return teu4_to_ext<S>(a); else return a; }
};
// k * P, k a plain integer of `nbits` bits: fixed 4-bit windows over a 16-entry per-lane table (private memory).
// (Binary double-and-add diverges on every bit, so a wave pays doubling + addition for all 253 of them; the window
// form pays 4 doublings + 1 addition per nibble.)
template <class S> AVRF_DN te_ext te_smul(te_pre p, fp k, int nbits) {
  te_ext tab[16];
  tab[0] = te_identity<S>(); tab[1] = te_from_pre<S>(p);
  for (int i = 2; i < 16; i++) tab[i] = te_madd<S>(tab[i - 1], p);
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = (nbits + 3) / 4 - 1; w >= 0; w--) {
    acc = CH::template dbln<4>(acc);
    uint32_t d = (k.v[w >> 3] >> (4 * (w & 7))) & 15u;
    if (d) acc = CH::add(acc, tab[d]);
  }
  return CH::finish(acc);
}
// ---- the same with the window table in the CONTEXT'S WORKSPACE instead of private memory.
// A per-lane array indexed by a runtime digit cannot live in registers; the compiler puts it in scratch, whose layout
// interleaves the lanes dword by dword -- lane L's entry d then touches 32 different 256-byte rows per lookup, and round 2's
// PMC pass measured 250 x the algorithmic bytes (3.5 GB fetched per 65 536 proofs) with the kernels at one wave per SIMD.
// Here item j owns ITEM_TAB_SLOTS consecutive 128-byte entries: a lookup reads two whole 64-byte lines, a table build writes
// them once, and nothing of the table counts against the kernel's private segment.  (One wave's writes followed by its own
// reads of the same addresses are ordered by the memory pipeline.)
enum { ITEM_TAB_SLOTS = 40 };    // 16 + 16 + 4 entries of the three-term verifier chain, rounded up
template <class S> AVRF_DN te_ext te_smul_ws(te_ext *tab, te_pre p, fp k, int nbits) {
  te_ext cur = te_from_pre<S>(p);
  store_ext(tab + 1, cur);
  for (int i = 2; i < 16; i++) { cur = te_madd<S>(cur, p); store_ext(tab + i, cur); }
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = (nbits + 3) / 4 - 1; w >= 0; w--) {
    const uint32_t d = (k.v[w >> 3] >> (4 * (w & 7))) & 15u;
    te_ext e; if (d) e = load_ext(tab + d);                      // issued ahead of the doublings
    acc = CH::template dbln<4>(acc);
    if (d) acc = CH::add(acc, e);
  }
  return CH::finish(acc);
}
// a*P + b*Q, joint 2-bit windows, table i*P + j*Q (i, j < 4) in the workspace
template <class S> AVRF_DN te_ext te_smul2_ws(te_ext *tab, te_pre p, fp a, te_pre q, fp b, int nbits) {
  te_ext row = te_identity<S>();
  for (int j = 0; j < 4; j++) {
    if (j) row = te_madd<S>(row, q);
    te_ext cur = row;
    if (j) store_ext(tab + 4 * j, cur);
    for (int i = 1; i < 4; i++) { cur = te_madd<S>(cur, p); store_ext(tab + 4 * j + i, cur); }
  }
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = (nbits + 1) / 2 - 1; w >= 0; w--) {
    const uint32_t da = (a.v[w >> 4] >> (2 * (w & 15))) & 3u, db = (b.v[w >> 4] >> (2 * (w & 15))) & 3u;
    te_ext e; if (da | db) e = load_ext(tab + 4 * db + da);
    acc = CH::template dbln<2>(acc);
    if (da | db) acc = CH::add(acc, e);
  }
  return CH::finish(acc);
}
// Fixed-base table (one per context, built by k_fixed_table): tab[(base * 32 + w) * 256 + d] = d * 2^(8w) * P,
// base 0 = the suite generator G, 1 = BLINDING_BASE; d = 0 unused.  k * P is then at most 32 mixed additions.
enum { FIXED_G = 0, FIXED_B = 1, FIXED_TABLE_POINTS = 2 * 32 * 256 };
template <class S> AVRF_DN te_ext te_smul_fixed(const te_pre *tab, int base, fp k) {
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  const te_pre *t = tab + (size_t)base * 32 * 256;
  for (int w = 0; w < 32; w++) {
    uint32_t d = (k.v[w >> 2] >> (8 * (w & 3))) & 255u;
    if (d) acc = CH::madd(acc, load_pre(t + w * 256 + d));
  }
  return CH::finish(acc);
}
// a*P + b*Q (Shamir's trick), a/b plain integers of nbits bits: joint 2-bit windows, table i*P + j*Q (i, j < 4)
template <class S> AVRF_DN te_ext te_smul2(te_pre p, fp a, te_pre q, fp b, int nbits) {
  te_ext tab[16];
  tab[0] = te_identity<S>();
  for (int i = 1; i < 4; i++) tab[i] = te_madd<S>(tab[i - 1], p);              // i*P
  for (int j = 1; j < 4; j++) for (int i = 0; i < 4; i++) tab[4 * j + i] = te_madd<S>(tab[4 * (j - 1) + i], q);
  using CH = Chain<S>;
  typename CH::acc_t acc = CH::identity();
  for (int w = (nbits + 1) / 2 - 1; w >= 0; w--) {
    acc = CH::template dbln<2>(acc);
    uint32_t da = (a.v[w >> 4] >> (2 * (w & 15))) & 3u, db = (b.v[w >> 4] >> (2 * (w & 15))) & 3u;
    if (da | db) acc = CH::add(acc, tab[4 * db + da]);
  }
  return CH::finish(acc);
}
template <class S> AVRF_DI void store_xy(uint8_t *out, const te_aff &a) {
  using Fq = typename S::Fq;
  fp_store_le(out, fp_from_mont<Fq>(a.x)); fp_store_le(out + 32, fp_from_mont<Fq>(a.y));
}
template <class S, class T> AVRF_DI void absorb_point_mont(T &h, const te_aff &a) {
  using Fq = typename S::Fq;
  absorb_point_xy<S>(h, fp_from_mont<Fq>(a.x), fp_from_mont<Fq>(a.y));
}
template <class S> AVRF_DI bool ext_eq_aff(const te_ext &p, const te_pre &q) {   // p == q ?
  using Fq = typename S::Fq;
  if constexpr (S::SW_NATIVE) {                                                  // X = x ZZ, Y = y ZZZ; infinity only equals infinity
    const bool qi = fp_is_zero(q.x) && fp_is_zero(q.y);
    if (fp_is_zero(p.t) || qi) return fp_is_zero(p.t) && qi;
    return fp_eq(p.x, fp_mul<Fq>(q.x, p.t)) && fp_eq(p.y, fp_mul<Fq>(q.y, p.z));
  }
  return fp_eq(p.x, fp_mul<Fq>(q.x, p.z)) && fp_eq(p.y, fp_mul<Fq>(q.y, p.z));
}

}  // namespace avrf
