// avrf.hpp -- header-only C++17 mirror of the reference's scheme-layer interface for the
// accelerated path, on top of the C ABI (avrf.h).  The reference is Rust; with no Rust toolchain
// in this image the host side above the boundary is C++ with the reference's names, argument
// meaning and error behaviour, so call sites read like the reference's own tests:
//
//   ark_vrf::Suite                         -> avrf::Suite            (src/lib.rs:177-250)
//   ark_vrf::{Secret, Public, VrfIo}       -> avrf::Secret / Public / VrfIo   (src/lib.rs:258-635)
//   ark_vrf::Error                         -> avrf::Error            (src/lib.rs:135-147)
//   tiny::{Proof, Prover, Verifier}                    -> avrf::tiny::...     (src/tiny.rs:48-214)
//   thin::{Proof, Prover, Verifier, BatchVerifier}     -> avrf::thin::...     (src/thin.rs:43-326)
//   pedersen::{Proof, Prover, Verifier, BatchVerifier} -> avrf::pedersen::... (src/pedersen.rs:69-426)
//   ring::{RingSetup, RingProverKey, RingVerifierKey, RingProver, RingVerifier, Proof, Prover, Verifier,
//          BatchVerifier}                             -> avrf::ring::...     (src/ring.rs:160-735)
//
// Differences forced by the device: provers/verifiers take *batches* of independent items (one GPU
// lane each); single-item calls are batches of one.  Points are 64-byte affine x||y (see avrf.h).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "avrf.h"

namespace avrf {

enum class Error : int { VerificationFailure = 1, InvalidData = 2, RingCapacityExceeded = 3, SrsLookupFailed = 4 };
// Result<(), Error>: 0 = Ok(())
using Status = int;

using Point = std::array<uint8_t, 64>;   // affine x || y, LE32 each
using Scalar = std::array<uint8_t, 32>;  // LE32 canonical

struct VrfIo { Point input, output; };   // src/lib.rs:615-619

// trait Suite: a context binds the suite parameterisation to one GPU stream.
class Suite {
 public:
  enum Id { BandersnatchSha512Ell2 = AVRF_SUITE_BANDERSNATCH_SHA512_ELL2, BabyJubJubSha512Tai = AVRF_SUITE_BABYJUBJUB_SHA512_TAI,
            JubJubSha512Tai = AVRF_SUITE_JUBJUB_SHA512_TAI, Ed25519Sha512Tai = AVRF_SUITE_ED25519_SHA512_TAI,
            BandersnatchSwSha512Tai = AVRF_SUITE_BANDERSNATCH_SW_SHA512_TAI,
            BandersnatchShake128Ell2 = AVRF_SUITE_BANDERSNATCH_SHAKE128_ELL2, TestingSha256Tai = AVRF_SUITE_TESTING_SHA256_TAI,
            Secp256r1Sha256Tai = AVRF_SUITE_SECP256R1_SHA256_TAI };
  // whose job is Validate::Yes? 0: the caller's (typed-point contract of the reference), 1: on-curve check, 2: + subgroup check
  void set_validation(int level) { if (avrf_ctx_set_validation(ctx_, level) != AVRF_OK) throw std::invalid_argument("avrf: validation level"); }
  explicit Suite(Id id, int device = 0) {
    if (avrf_ctx_create(id, device, &ctx_) != AVRF_OK) throw std::runtime_error("avrf: no MI355X device (there is no CPU fallback)");
  }
  ~Suite() { avrf_ctx_destroy(ctx_); }
  Suite(const Suite &) = delete;
  Suite &operator=(const Suite &) = delete;
  avrf_ctx *ctx() const { return ctx_; }

 private:
  avrf_ctx *ctx_ = nullptr;
};

// Input::new(data) (src/lib.rs:440-444): hash-to-curve on the device; throws where the reference returns None
inline Point input_new(const Suite &s, const std::string &data) {
  Point p; uint32_t len = (uint32_t)data.size(); int32_t st = 0;
  if (avrf_hash_to_curve(s.ctx(), 1, (const uint8_t *)data.data(), &len, p.data(), &st) != AVRF_OK || st != 0) throw std::invalid_argument("avrf: hash to curve");
  return p;
}

struct Public { Point point; };          // src/lib.rs:408
struct Secret {                          // src/lib.rs:258
  Scalar scalar; Public public_key;
  // Secret::from_scalar (src/lib.rs:331-334): pk = sk * G on the device
  static Secret from_scalar(const Suite &s, const Scalar &sk) {
    Secret r; r.scalar = sk;
    if (avrf_scalar_mul_base(s.ctx(), 1, sk.data(), r.public_key.point.data()) != AVRF_OK) throw std::invalid_argument("avrf: bad scalar");
    return r;
  }
  // Secret::from_seed (src/lib.rs:346-369): the scalar is derived from the seed by the suite's transcript, on the device
  static Secret from_seed(const Suite &s, const std::array<uint8_t, 32> &seed) {
    Secret r;
    if (avrf_secret_from_seed(s.ctx(), 1, seed.data(), r.scalar.data(), r.public_key.point.data()) != AVRF_OK) throw std::invalid_argument("avrf: from_seed");
    return r;
  }
  // Secret::output / vrf_io (src/lib.rs:391-401)
  VrfIo vrf_io(const Suite &s, const Point &input) const {
    VrfIo io; io.input = input;
    if (avrf_scalar_mul(s.ctx(), 1, scalar.data(), input.data(), io.output.data()) != AVRF_OK) throw std::invalid_argument("avrf: bad input");
    return io;
  }
};

// Output::hash::<32> (src/lib.rs:605-609): the VRF output bytes of an output point
inline std::array<uint8_t, 32> output_hash(const Suite &s, const Point &output) {
  std::array<uint8_t, 32> h{};
  if (avrf_output_hash(s.ctx(), 1, output.data(), 32, h.data()) != AVRF_OK) throw std::invalid_argument("avrf: output hash");
  return h;
}

namespace detail {
// flattens {ios per item, ad per item} into the C-ABI layout
struct Packed {
  std::vector<uint8_t> ios, ads; std::vector<uint32_t> io_counts, ad_lens;
  void push(const std::vector<VrfIo> &item_ios, const std::string &ad) {
    for (const VrfIo &io : item_ios) { ios.insert(ios.end(), io.input.begin(), io.input.end()); ios.insert(ios.end(), io.output.begin(), io.output.end()); }
    io_counts.push_back((uint32_t)item_ios.size());
    ads.insert(ads.end(), ad.begin(), ad.end()); ad_lens.push_back((uint32_t)ad.size());
  }
};
}  // namespace detail

namespace tiny {

struct Proof { std::array<uint8_t, 16> c; Scalar s; };     // src/tiny.rs:48-53  (wire: c(16) || s(32), src/tiny.rs:60-78)

// tiny::Prover::prove for Secret (src/tiny.rs:163-176)
inline Proof prove(const Suite &su, const Secret &sk, const std::vector<VrfIo> &ios, const std::string &ad) {
  detail::Packed p; p.push(ios, ad);
  uint8_t out[48];
  if (avrf_tiny_prove(su.ctx(), 1, sk.scalar.data(), sk.public_key.point.data(), p.ios.data(), p.io_counts.data(), p.ads.data(), p.ad_lens.data(), out) != AVRF_OK)
    throw std::invalid_argument("avrf: tiny prove");
  Proof pr; std::copy(out, out + 16, pr.c.begin()); std::copy(out + 16, out + 48, pr.s.begin());
  return pr;
}
// tiny::Verifier::verify for Public (src/tiny.rs:178-214)
inline Status verify(const Suite &su, const Public &pk, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {
  detail::Packed p; p.push(ios, ad);
  uint8_t pr[48]; std::copy(proof.c.begin(), proof.c.end(), pr); std::copy(proof.s.begin(), proof.s.end(), pr + 16);
  int32_t st = 0;
  int rc = avrf_tiny_verify(su.ctx(), 1, pk.point.data(), p.ios.data(), p.io_counts.data(), p.ads.data(), p.ad_lens.data(), pr, &st);
  return rc != AVRF_OK ? rc : st;
}

}  // namespace tiny

namespace thin {

struct Proof { Point r; Scalar s; };     // src/thin.rs:43-48  (wire: R_xy || s)

// thin::Prover::prove for Secret (src/thin.rs:111-129)
inline Proof prove(const Suite &su, const Secret &sk, const std::vector<VrfIo> &ios, const std::string &ad) {
  detail::Packed p; p.push(ios, ad);
  uint8_t out[96];
  if (avrf_thin_prove(su.ctx(), 1, sk.scalar.data(), sk.public_key.point.data(), p.ios.data(), p.io_counts.data(),
                      p.ads.data(), p.ad_lens.data(), out) != AVRF_OK) throw std::invalid_argument("avrf: thin prove");
  Proof pr; std::copy(out, out + 64, pr.r.begin()); std::copy(out + 64, out + 96, pr.s.begin()); return pr;
}
// thin::Verifier::verify for Public (src/thin.rs:131-165): 0 = Ok, else Error
inline Status verify(const Suite &su, const Public &pk, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {
  detail::Packed p; p.push(ios, ad);
  uint8_t pr[96]; std::copy(proof.r.begin(), proof.r.end(), pr); std::copy(proof.s.begin(), proof.s.end(), pr + 64);
  int32_t st = 0;
  if (avrf_thin_verify(su.ctx(), 1, pk.point.data(), p.ios.data(), p.io_counts.data(), p.ads.data(), p.ad_lens.data(), pr, &st) != AVRF_OK)
    throw std::runtime_error("avrf: thin verify");
  return st;
}

// thin::BatchVerifier (src/thin.rs:188-326): new / push / verify
class BatchVerifier {
 public:
  explicit BatchVerifier(const Suite &su) : su_(su) {}
  void push(const Public &pk, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {   // :234-243
    pks_.insert(pks_.end(), pk.point.begin(), pk.point.end());
    packed_.push(ios, ad);
    proofs_.insert(proofs_.end(), proof.r.begin(), proof.r.end()); proofs_.insert(proofs_.end(), proof.s.begin(), proof.s.end());
    n_++;
  }
  Status verify() const {                                                                                     // :257-325
    return avrf_thin_batch_verify(su_.ctx(), n_, pks_.data(), packed_.ios.data(), packed_.io_counts.data(), packed_.ads.data(),
                                  packed_.ad_lens.data(), proofs_.data());
  }
  size_t len() const { return n_; }

 private:
  const Suite &su_; size_t n_ = 0;
  std::vector<uint8_t> pks_, proofs_; detail::Packed packed_;
  friend class VerifierPool;
};

// Many thin::BatchVerifier::verify calls in flight on one device (avrf_pool, include/avrf.h): submit() hands a filled
// BatchVerifier over and returns a ticket at once, wait() blocks for that batch's verdict.  The BatchVerifier (its buffers)
// must stay alive and unchanged until wait() has returned for its ticket.
class VerifierPool {
 public:
  VerifierPool(int suite, int device, int slots = 8, int lanes = 4, int threads = 2, int hash_group = 8) {
    if (avrf_pool_create(suite, device, 1, slots, lanes, 0, threads, hash_group, &p_) != AVRF_OK) throw std::runtime_error("avrf: pool create");
  }
  ~VerifierPool() { avrf_pool_destroy(p_); }
  VerifierPool(const VerifierPool &) = delete;
  VerifierPool &operator=(const VerifierPool &) = delete;
  uint64_t submit(const BatchVerifier &b) {
    uint64_t t = 0;
    if (avrf_pool_submit(p_, b.n_, b.pks_.data(), b.packed_.ios.data(), b.packed_.io_counts.data(), b.packed_.ads.data(), b.packed_.ad_lens.data(),
                         b.proofs_.data(), &t) != AVRF_OK) throw std::runtime_error("avrf: pool submit");
    return t;
  }
  Status wait(uint64_t ticket) {
    int st = 0;
    if (avrf_pool_wait(p_, ticket, &st) != AVRF_OK) throw std::runtime_error("avrf: pool wait");
    return st;
  }
 private:
  avrf_pool *p_ = nullptr;
};

}  // namespace thin

namespace pedersen {

struct Proof { Point pk_com, r, ok; Scalar s, sb; };   // src/pedersen.rs:69-75 (wire: Yb_xy || R_xy || Ok_xy || s || sb)

inline void to_wire(const Proof &p, uint8_t out[256]) {
  std::copy(p.pk_com.begin(), p.pk_com.end(), out); std::copy(p.r.begin(), p.r.end(), out + 64); std::copy(p.ok.begin(), p.ok.end(), out + 128);
  std::copy(p.s.begin(), p.s.end(), out + 192); std::copy(p.sb.begin(), p.sb.end(), out + 224);
}
// pedersen::Prover::prove (src/pedersen.rs:136-186): returns (proof, blinding)
inline std::pair<Proof, Scalar> prove(const Suite &su, const Secret &sk, const std::vector<VrfIo> &ios, const std::string &ad) {
  detail::Packed p; p.push(ios, ad);
  uint8_t out[256]; Scalar bl;
  if (avrf_pedersen_prove(su.ctx(), 1, sk.scalar.data(), sk.public_key.point.data(), p.ios.data(), p.io_counts.data(), p.ads.data(),
                          p.ad_lens.data(), out, bl.data()) != AVRF_OK) throw std::invalid_argument("avrf: pedersen prove");
  Proof pr;
  std::copy(out, out + 64, pr.pk_com.begin()); std::copy(out + 64, out + 128, pr.r.begin()); std::copy(out + 128, out + 192, pr.ok.begin());
  std::copy(out + 192, out + 224, pr.s.begin()); std::copy(out + 224, out + 256, pr.sb.begin());
  return {pr, bl};
}
// pedersen::Verifier::verify (src/pedersen.rs:188-249)
inline Status verify(const Suite &su, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {
  detail::Packed p; p.push(ios, ad);
  uint8_t pr[256]; to_wire(proof, pr);
  int32_t st = 0;
  if (avrf_pedersen_verify(su.ctx(), 1, p.ios.data(), p.io_counts.data(), p.ads.data(), p.ad_lens.data(), pr, &st) != AVRF_OK)
    throw std::runtime_error("avrf: pedersen verify");
  return st;
}
// pedersen::BatchVerifier (src/pedersen.rs:303-426)
class BatchVerifier {
 public:
  explicit BatchVerifier(const Suite &su) : su_(su) {}
  void push(const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {                        // :324-326
    packed_.push(ios, ad);
    uint8_t pr[256]; to_wire(proof, pr); proofs_.insert(proofs_.end(), pr, pr + 256); n_++;
  }
  Status verify() const {                                                                                     // :341-426
    return avrf_pedersen_batch_verify(su_.ctx(), n_, packed_.ios.data(), packed_.io_counts.data(), packed_.ads.data(),
                                      packed_.ad_lens.data(), proofs_.data());
  }

 private:
  const Suite &su_; size_t n_ = 0;
  std::vector<uint8_t> proofs_; detail::Packed packed_;
};

}  // namespace pedersen

namespace ring {

using RingCommitment = std::vector<uint8_t>;   // 3 x G1 compressed (144 B BLS12-381 / 96 B BN254), src/ring.rs:125-127
using RingBareProof = std::vector<uint8_t>;    // RingBareProof, compressed serialisation (592 B / 480 B)

// ring::Proof (src/ring.rs:160-166)
struct Proof { pedersen::Proof pedersen_proof; RingBareProof ring_proof; };

// RingSetup (src/ring.rs:340-464): PCS parameters + ring context, SRS resident on the device
class RingSetup {
 public:
  // RingSetup::from_pcs_params (src/ring.rs:380-393): `srs` = URS { powers_in_g1, powers_in_g2 }, serialize_uncompressed
  static Status from_pcs_params(const Suite &su, size_t ring_size, const std::vector<uint8_t> &srs, RingSetup *out) {
    out->reset();
    return avrf_ring_setup_load(su.ctx(), srs.data(), srs.size(), ring_size, &out->h_);
  }
  // RingSetup::from_seed (src/ring.rs:359-366): the deterministic setup of a seed, derived as the reference derives it
  // (avrf_ring_setup_from_seed; unpinned by any reference vector, see include/avrf.h)
  static Status from_seed(const Suite &su, size_t ring_size, const std::array<uint8_t, 32> &seed, RingSetup *out) {
    out->reset();
    return avrf_ring_setup_from_seed(su.ctx(), seed.data(), ring_size, &out->h_);
  }
  // Kzg::setup with an explicit trapdoor (what from_seed / from_rand reach, src/ring.rs:359-374); g1 / g2: generator
  // entries in the URS encoding
  static std::vector<uint8_t> generate_pcs_params(const Suite &su, int suite_id, size_t ring_size, const Scalar &tau, const std::vector<uint8_t> &g1,
                                                  const std::vector<uint8_t> &g2) {
    const size_t n_g1 = avrf_ring_pcs_domain_size(suite_id, ring_size);
    std::vector<uint8_t> out(8 + n_g1 * g1.size() + 8 + 2 * g2.size()); size_t len = 0;
    if (avrf_ring_srs_generate(su.ctx(), tau.data(), g1.data(), g2.data(), n_g1, out.data(), out.size(), &len) != AVRF_OK)
      throw std::invalid_argument("avrf: srs generate");
    out.resize(len);
    return out;
  }
  RingSetup() = default;
  ~RingSetup() { reset(); }
  RingSetup(const RingSetup &) = delete;
  RingSetup &operator=(const RingSetup &) = delete;
  size_t max_ring_size() const { return avrf_ring_max_ring_size(h_); }       // src/ring.rs:298-300
  size_t proof_len() const { return avrf_ring_proof_len(h_); }
  avrf_ring_setup *handle() const { return h_; }

 private:
  void reset() { if (h_) avrf_ring_setup_free(h_); h_ = nullptr; }
  avrf_ring_setup *h_ = nullptr;
};

// RingProverKey + RingVerifierKey of one ring (`ring_proof::index`, src/ring.rs:399-417)
class RingKey {
 public:
  // RingSetup::prover_key / verifier_key: Err(RingCapacityExceeded) when pks.len() > max_ring_size
  static Status index(const RingSetup &setup, const std::vector<Public> &pks, RingKey *out) {
    out->reset();
    std::vector<uint8_t> xy; for (const Public &p : pks) xy.insert(xy.end(), p.point.begin(), p.point.end());
    out->commitment_.assign(avrf_ring_commitment_len(setup.handle()), 0);
    out->setup_ = &setup;
    return avrf_ring_index(setup.handle(), xy.data(), pks.size(), &out->h_, out->commitment_.data());
  }
  RingKey() = default;
  ~RingKey() { reset(); }
  RingKey(const RingKey &) = delete;
  RingKey &operator=(const RingKey &) = delete;
  const RingCommitment &commitment() const { return commitment_; }          // RingVerifierKey::commitment
  avrf_ring_key *handle() const { return h_; }
  const RingSetup &setup() const { return *setup_; }

 private:
  void reset() { if (h_) avrf_ring_key_free(h_); h_ = nullptr; }
  avrf_ring_key *h_ = nullptr; const RingSetup *setup_ = nullptr; RingCommitment commitment_;
};

// RingSetup::prover(prover_key, key_index) (src/ring.rs:419-426)
struct RingProver { const RingKey *key; uint32_t key_index; bool hiding = false; };
// RingSetup::verifier(verifier_key) / verifier_key_from_commitment (src/ring.rs:428-440,477-521)
struct RingVerifier { const RingSetup *setup; RingCommitment commitment; };

// ring::Prover::prove for Secret (src/ring.rs:211-226): Pedersen proof, then the ring proof of its blinding
inline Proof prove(const Suite &su, const Secret &sk, const std::vector<VrfIo> &ios, const std::string &ad, const RingProver &prover) {
  auto pp = pedersen::prove(su, sk, ios, ad);
  Proof pr; pr.pedersen_proof = pp.first;
  pr.ring_proof.assign(prover.key->setup().proof_len(), 0);
  if (avrf_ring_prove(prover.key->handle(), 1, &prover.key_index, pp.second.data(), prover.hiding ? 1 : 0, pr.ring_proof.data()) != AVRF_OK)
    throw std::invalid_argument("avrf: ring prove");
  return pr;
}
// ring::Verifier::verify for Public (src/ring.rs:228-247): Pedersen verify, then the ring proof of the key commitment
inline Status verify(const Suite &su, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof, const RingVerifier &verifier) {
  Status st = pedersen::verify(su, ios, ad, proof.pedersen_proof);
  if (st) return st;
  return avrf_ring_batch_verify(verifier.setup->handle(), 1, verifier.commitment.data(), 1, nullptr, proof.pedersen_proof.pk_com.data(),
                                proof.ring_proof.data());
}

// ring::BatchVerifier (src/ring.rs:682-735): proofs from one or more rings sharing the SRS
class BatchVerifier {
 public:
  BatchVerifier(const Suite &su, const RingVerifier &ring_verifier) : su_(su), setup_(ring_verifier.setup), ped_(su) {}
  // push (src/ring.rs:713-723)
  void push(const RingVerifier &verifier, const std::vector<VrfIo> &ios, const std::string &ad, const Proof &proof) {
    uint32_t ring = 0;
    for (; ring < rings_.size(); ring++) if (rings_[ring] == verifier.commitment) break;
    if (ring == rings_.size()) rings_.push_back(verifier.commitment);
    ring_of_.push_back(ring);
    ped_.push(ios, ad, proof.pedersen_proof);
    inst_.insert(inst_.end(), proof.pedersen_proof.pk_com.begin(), proof.pedersen_proof.pk_com.end());
    proofs_.insert(proofs_.end(), proof.ring_proof.begin(), proof.ring_proof.end());
  }
  // verify (src/ring.rs:729-735): Pedersen batch first, then the ring batch
  Status verify() const {
    Status st = ped_.verify();
    if (st) return st;
    std::vector<uint8_t> coms; for (const RingCommitment &c : rings_) coms.insert(coms.end(), c.begin(), c.end());
    return avrf_ring_batch_verify(setup_->handle(), ring_of_.size(), coms.data(), rings_.size(), ring_of_.data(), inst_.data(), proofs_.data());
  }

 private:
  const Suite &su_; const RingSetup *setup_;
  pedersen::BatchVerifier ped_;
  std::vector<RingCommitment> rings_; std::vector<uint32_t> ring_of_;
  std::vector<uint8_t> inst_, proofs_;
};

}  // namespace ring
}  // namespace avrf
