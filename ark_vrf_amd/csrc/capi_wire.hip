// capi_wire.hip -- the wire-format flavour of the C ABI: the same verifier / prover entry points taking the
// reference's `serialize_compressed` encodings (32-byte points, proofs exactly as `CanonicalSerialize` writes them) with a
// `validate` flag, and ring::Prover::prove / ring::Verifier::verify as single calls.  Pure composition of the entry points
// of capi.hip / ring.hip: points are decompressed (validated) by avrf_points_decompress on the device, then the xy entry
// point runs.  No kernels of its own.
//
//   thin proof      R(32) || s(32)                          src/thin.rs:43-48
//   tiny proof      c(16) || s(32)                          src/tiny.rs:60-78
//   pedersen proof  Yb(32) || R(32) || Ok(32) || s || sb    src/pedersen.rs:69-75      (160 bytes)
//   ring-VRF proof  pedersen proof || ring proof            src/ring.rs:160-166        (752 / 640 bytes)
#include "../../include/avrf.h"
#include <string.h>
#include <thread>
#include <vector>

extern "C" int avrf_ctx_suite_(avrf_ctx *c);

namespace {

// decompress `count` points given as (pointer, stride) records into contiguous xy; returns AVRF_OK or the failing status
struct Gather {
  size_t L;                                  // avrf_point_len of the context's suite
  std::vector<uint8_t> comp;
  explicit Gather(size_t l) : L(l) {}
  void add(const uint8_t *p, size_t n, size_t stride) { for (size_t i = 0; i < n; i++) comp.insert(comp.end(), p + i * stride, p + i * stride + L); }
};
int decompress_all(avrf_ctx *ctx, const Gather &g, int validate, std::vector<uint8_t> &xy, std::vector<int32_t> *per_point = nullptr) {
  const size_t n = g.comp.size() / g.L;
  xy.assign(n * 64, 0);
  std::vector<int32_t> st(n ? n : 1, 0);
  int rc = avrf_points_decompress(ctx, n, g.comp.data(), xy.data(), validate, st.data());
  if (rc != AVRF_OK) return rc;
  if (per_point) { *per_point = st; return AVRF_OK; }
  for (size_t i = 0; i < n; i++) if (st[i]) return AVRF_INVALID_DATA;
  return AVRF_OK;
}
size_t sum_counts(const uint32_t *c, size_t n) { size_t t = 0; for (size_t i = 0; i < n; i++) t += c[i]; return t; }

// shared by the thin / tiny / pedersen wire verifiers: kind 1 thin (64-byte proofs, 1 point), 3 tiny (48, 0 points), 2 pedersen (160, 3)
// pp0_xy / pp0_st (optional): xy and decode status of every item's FIRST proof point (the Pedersen key commitment Yb, which
// ring::Verifier also needs as the ring proof's instance) -- so that a caller does not decompress it a second time
// (Until the single-launch MSM -- msm.hip k_msm_tiny_bits -- a "batch" of <= 64 items went through the per-item
// verifiers here, 0.6 ms against 1.5 ms for the Pippenger chain on one item; the batch verifier now takes 0.30 ms for one item,
// 0.32 for eight, 0.46 for 64, against 0.52-0.55 ms per per-item call, so a batch is a batch whatever its size: SMALL_BATCH = 0.)
constexpr size_t SMALL_BATCH = 0;
// The two phases of a wire verification, separate so that a caller can run the verification proper beside other work (the ring
// half of avrf_ring_vrf_verify): wire_prepare decompresses (and validates) every point and lays the proofs out in the x || y form;
// wire_finish runs the verifier and folds the points' decode statuses into the items' statuses.
struct WirePrep {
  int kind = 0; bool batch = false, small = false; size_t n = 0, tot = 0, ppts = 0;
  const uint32_t *io_counts = nullptr, *ad_lens = nullptr; const uint8_t *ads = nullptr;
  std::vector<uint8_t> xy, px; std::vector<int32_t> pst, small_status;
  const uint8_t *x_pks = nullptr, *x_ios = nullptr;
};
int wire_prepare(avrf_ctx *ctx, int kind, bool batch, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                 const uint32_t *ad_lens, const uint8_t *proofs, int validate, bool have_status_out, WirePrep &w,
                 std::vector<uint8_t> *pp0_xy = nullptr, std::vector<int32_t> *pp0_st = nullptr) {
  w.small = batch && n <= SMALL_BATCH && kind != 3;
  if (w.small) { w.small_status.assign(n ? n : 1, 0); batch = false; have_status_out = true; }
  if (!ctx || (n && (!io_counts || !ad_lens || !proofs)) || (n && kind != 2 && !pks) || (!batch && n && !have_status_out)) return AVRF_ERR_BAD_ARG;
  w.kind = kind; w.batch = batch; w.n = n; w.io_counts = io_counts; w.ad_lens = ad_lens; w.ads = ads;
  if (!n) return AVRF_OK;
  const size_t tot = sum_counts(io_counts, n);
  if (tot && !ios) return AVRF_ERR_BAD_ARG;
  w.tot = tot;
  const size_t L = avrf_point_len(avrf_ctx_suite_(ctx));
  const size_t ppts = kind == 1 ? 1 : kind == 3 ? 0 : 3, plen = ppts * L + (kind == 1 ? 32 : kind == 3 ? 48 : 64), xlen = kind == 1 ? 96 : kind == 3 ? 48 : 256;
  w.ppts = ppts;
  Gather g(L);
  if (kind != 2) g.add(pks, n, L);
  g.add(ios, 2 * tot, L);
  for (size_t p = 0; p < ppts; p++) g.add(proofs + L * p, n, plen);
  int rc = decompress_all(ctx, g, validate, w.xy, (batch && !pp0_st) ? nullptr : &w.pst);
  if (rc != AVRF_OK) return rc;
  w.x_pks = w.xy.data(); w.x_ios = w.xy.data() + (kind != 2 ? n * 64 : 0);
  const uint8_t *x_pp = w.x_ios + 2 * tot * 64;
  if (pp0_xy && ppts) pp0_xy->assign(x_pp, x_pp + n * 64);
  if (pp0_st && ppts) pp0_st->assign(w.pst.begin() + ((kind != 2 ? n : 0) + 2 * tot), w.pst.begin() + ((kind != 2 ? n : 0) + 2 * tot + n));
  if (batch && pp0_st) for (size_t i = 0; i < w.pst.size(); i++) if (w.pst[i]) return AVRF_INVALID_DATA;
  w.px.resize(n * xlen);
  for (size_t j = 0; j < n; j++) {
    for (size_t p = 0; p < ppts; p++) memcpy(&w.px[j * xlen + 64 * p], x_pp + (p * n + j) * 64, 64);
    memcpy(&w.px[j * xlen + 64 * ppts], proofs + j * plen + L * ppts, plen - L * ppts);
  }
  return AVRF_OK;
}
int wire_finish(avrf_ctx *ctx, WirePrep &w, int32_t *status_out) {
  const size_t n = w.n, tot = w.tot, ppts = w.ppts; const int kind = w.kind;
  if (!n) return AVRF_OK;
  if (w.small) status_out = w.small_status.data();
  if (w.batch) {
    if (kind == 1) return avrf_thin_batch_verify(ctx, n, w.x_pks, w.x_ios, w.io_counts, w.ads, w.ad_lens, w.px.data());
    return avrf_pedersen_batch_verify(ctx, n, w.x_ios, w.io_counts, w.ads, w.ad_lens, w.px.data());
  }
  int rc;
  if (kind == 1) rc = avrf_thin_verify(ctx, n, w.x_pks, w.x_ios, w.io_counts, w.ads, w.ad_lens, w.px.data(), status_out);
  else if (kind == 3) rc = avrf_tiny_verify(ctx, n, w.x_pks, w.x_ios, w.io_counts, w.ads, w.ad_lens, w.px.data(), status_out);
  else rc = avrf_pedersen_verify(ctx, n, w.x_ios, w.io_counts, w.ads, w.ad_lens, w.px.data(), status_out);
  if (rc != AVRF_OK) return rc;
  // a point that does not decode (or fails validation) makes ITS item InvalidData
  const std::vector<int32_t> &pst = w.pst;
  size_t at = 0, io_at = 0;
  if (kind != 2) { for (size_t j = 0; j < n; j++) if (pst[j]) status_out[j] = AVRF_INVALID_DATA; at = n; }
  for (size_t j = 0; j < n; j++) { for (size_t k = 0; k < 2 * (size_t)w.io_counts[j]; k++) if (pst[at + io_at + k]) status_out[j] = AVRF_INVALID_DATA; io_at += 2 * w.io_counts[j]; }
  at += 2 * tot;
  for (size_t p = 0; p < ppts; p++) for (size_t j = 0; j < n; j++) if (pst[at + p * n + j]) status_out[j] = AVRF_INVALID_DATA;
  if (w.small) {                                                       // BatchVerifier's verdict: InvalidData before VerificationFailure
    int worst = AVRF_OK;
    for (size_t j = 0; j < n; j++) { if (status_out[j] == AVRF_INVALID_DATA) return AVRF_INVALID_DATA; if (status_out[j]) worst = AVRF_VERIFICATION_FAILURE; }
    return worst;
  }
  return AVRF_OK;
}
int verify_wire(avrf_ctx *ctx, int kind, bool batch, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out,
                std::vector<uint8_t> *pp0_xy = nullptr, std::vector<int32_t> *pp0_st = nullptr) {
  if (batch && !pp0_xy && !pp0_st && (kind == 1 || kind == 2)) {
    // BatchVerifier from wire bytes: decompression on the device straight into the staged buffers (capi.hip ctx_stage_wire), then the run
    int rc = kind == 1 ? avrf_thin_batch_stage_wire(ctx, n, pks, ios, io_counts, ads, ad_lens, proofs, validate)
                       : avrf_pedersen_batch_stage_wire(ctx, n, ios, io_counts, ads, ad_lens, proofs, validate);
    if (rc != AVRF_OK) return rc;
    return kind == 1 ? avrf_thin_batch_run(ctx) : avrf_pedersen_batch_run(ctx);
  }
  WirePrep w;
  int rc = wire_prepare(ctx, kind, batch, n, pks, ios, io_counts, ads, ad_lens, proofs, validate, status_out != nullptr, w, pp0_xy, pp0_st);
  if (rc != AVRF_OK || !n) return rc;
  return wire_finish(ctx, w, status_out);
}

}  // namespace

extern "C" {

int avrf_thin_batch_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                const uint32_t *ad_lens, const uint8_t *proofs, int validate) {
  return verify_wire(ctx, 1, true, n, pks, ios, io_counts, ads, ad_lens, proofs, validate, nullptr);
}
int avrf_thin_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                          const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out) {
  return verify_wire(ctx, 1, false, n, pks, ios, io_counts, ads, ad_lens, proofs, validate, status_out);
}
int avrf_tiny_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                          const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out) {
  return verify_wire(ctx, 3, false, n, pks, ios, io_counts, ads, ad_lens, proofs, validate, status_out);
}
int avrf_pedersen_batch_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                    const uint32_t *ad_lens, const uint8_t *proofs, int validate) {
  return verify_wire(ctx, 2, true, n, nullptr, ios, io_counts, ads, ad_lens, proofs, validate, nullptr);
}
int avrf_pedersen_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                              const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out) {
  return verify_wire(ctx, 2, false, n, nullptr, ios, io_counts, ads, ad_lens, proofs, validate, status_out);
}

// ring::Prover::prove (src/ring.rs:211-226) for n provers of one ring: Pedersen proof + ring proof for its blinding, serialised
// as ring::Proof (src/ring.rs:160-166)
int avrf_ring_vrf_prove(avrf_ctx *ctx, avrf_ring_key *key, size_t ring_proof_len, size_t n, const uint8_t *sks, const uint32_t *key_index,
                        const uint8_t *ios_xy, const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, int blinding_mode,
                        uint8_t *proofs_out) {
  if (!ctx || !key || (n && (!sks || !key_index || !io_counts || !ad_lens || !proofs_out))) return AVRF_ERR_BAD_ARG;
  // the ring proof's length is a property of the key's setup: the argument only states the caller's layout and must agree
  avrf_ring_setup *setup = avrf_ring_key_setup(key);
  if (!setup || avrf_ring_setup_suite(setup) != avrf_ctx_suite_(ctx) || ring_proof_len != avrf_ring_proof_len(setup)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  const size_t L = avrf_point_len(avrf_ctx_suite_(ctx)), pedlen = 3 * L + 64;
  std::vector<uint8_t> ped(n * 256), blind(n * 32), rp(n * ring_proof_len), comp(n * 3 * L), pts(n * 3 * 64);
  int rc = avrf_pedersen_prove(ctx, n, sks, nullptr, ios_xy, io_counts, ads, ad_lens, ped.data(), blind.data());
  if (rc != AVRF_OK) return rc;
  rc = avrf_ring_prove(key, n, key_index, blind.data(), blinding_mode, rp.data());
  if (rc != AVRF_OK) return rc;
  for (size_t j = 0; j < n; j++) memcpy(&pts[j * 192], &ped[j * 256], 192);
  rc = avrf_points_compress(ctx, 3 * n, pts.data(), comp.data());
  if (rc != AVRF_OK) return rc;
  const size_t plen = pedlen + ring_proof_len;
  for (size_t j = 0; j < n; j++) {
    uint8_t *o = proofs_out + j * plen;
    memcpy(o, &comp[j * 3 * L], 3 * L); memcpy(o + 3 * L, &ped[j * 256 + 192], 64); memcpy(o + pedlen, &rp[j * ring_proof_len], ring_proof_len);
  }
  return AVRF_OK;
}

// ring::Verifier::verify for every item (each != 0: per-item statuses, src/ring.rs:228-247) or ring::BatchVerifier over all of
// them (each == 0: one status, src/ring.rs:693-735): Pedersen half + ring half on the key commitment Yb of the Pedersen proof
int avrf_ring_vrf_verify(avrf_ctx *ctx, avrf_ring_setup *setup, size_t n, const uint8_t *ring_commitments, size_t n_rings, const uint32_t *ring_of_item,
                         const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                         int validate, int each, int32_t *status_out) {
  if (!ctx || !setup || (n && (!ring_commitments || !n_rings || !io_counts || !ad_lens || !proofs)) || (each && n && !status_out)) return AVRF_ERR_BAD_ARG;
  if (avrf_ring_setup_suite(setup) != avrf_ctx_suite_(ctx)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  const size_t L = avrf_point_len(avrf_ctx_suite_(ctx)), pedlen = 3 * L + 64;
  const size_t rlen = avrf_ring_proof_len(setup), plen = pedlen + rlen, tot = sum_counts(io_counts, n);
  if (tot && !ios) return AVRF_ERR_BAD_ARG;
  std::vector<uint8_t> ped(n * pedlen), rp(n * rlen);
  for (size_t j = 0; j < n; j++) { memcpy(&ped[j * pedlen], proofs + j * plen, pedlen); memcpy(&rp[j * rlen], proofs + j * plen + pedlen, rlen); }
  // the Pedersen half first: it decompresses (and validates) every point of the Pedersen proofs ONCE, including Yb, the ring
  // verifier's instance
  // (the two halves only share Yb: once the points are decompressed, the Pedersen verifier runs on a thread of its own -- the
  // context's stream and buffers -- beside the ring half on the setup's stream and the host pool)
  std::vector<uint8_t> yb; std::vector<int32_t> yst;
  std::vector<int32_t> s1(n), s2(n);
  WirePrep w;
  int rc = wire_prepare(ctx, 2, !each, n, nullptr, ios, io_counts, ads, ad_lens, ped.data(), validate, each != 0, w, &yb, &yst);
  if (rc != AVRF_OK) return rc;
  int rc_ped = AVRF_OK;
  std::thread ped_half; bool threaded = true;
  try { ped_half = std::thread([&] { rc_ped = wire_finish(ctx, w, s1.data()); }); } catch (...) { threaded = false; }
  if (!each) rc = avrf_ring_batch_verify(setup, n, ring_commitments, n_rings, ring_of_item, yb.data(), rp.data());
  else rc = avrf_ring_verify_each(setup, n, ring_commitments, n_rings, ring_of_item, yb.data(), rp.data(), s2.data());
  if (threaded) ped_half.join(); else rc_ped = wire_finish(ctx, w, s1.data());
  if (rc_ped != AVRF_OK) return rc_ped;
  if (!each) return rc;
  if (rc != AVRF_OK) return rc;
  for (size_t j = 0; j < n; j++) status_out[j] = (yst[j] || s1[j] == AVRF_INVALID_DATA || s2[j] == AVRF_INVALID_DATA) ? AVRF_INVALID_DATA : (s1[j] || s2[j]) ? AVRF_VERIFICATION_FAILURE : AVRF_OK;
  return AVRF_OK;
}

}  // extern "C"
