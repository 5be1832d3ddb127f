/*
 * avrf.h -- C ABI of the MI355X-native batched VRF engine (libavrf.so).
 *
 * Drop-in boundary for the hot path of davxy/ark-vrf.  The reference has no
 * FFI of its own (`#![deny(unsafe_code)]`, src/lib.rs:95); each entry point below names the
 * reference interface it replaces (paths relative to the reference repo).  A Rust shim
 * re-implementing `thin::{Prover,Verifier,BatchVerifier}` / `pedersen::{...}` for
 * `Secret<S>` / `Public<S>` binds exactly these symbols.
 *
 * Conventions
 *  - plain pointers and sizes; the caller owns every buffer; handles are opaque.
 *  - scalars: 32-byte little-endian canonical integers (< group order), as
 *    `CanonicalSerialize` writes `ScalarField` (src/testing.rs:25-34).
 *  - "xy" points: 64 bytes, LE32(x) || LE32(y), canonical (non-Montgomery) affine
 *    twisted-Edwards coordinates, identity = (0, 1).  This is what a caller holding
 *    arkworks `Affine { x, y }` values has after deserialisation.
 *  - "compressed" points: 32 bytes, ark-serialize compressed form (LE32(y), bit 255 = x sign).
 *  - status codes mirror `ark_vrf::Error` (src/lib.rs:135-147).
 *  - every call runs on the context's own HIP stream; one context per host thread.
 *  - no CPU fallback: creating a context fails (AVRF_ERR_NO_DEVICE) without a gfx950 device.
 *  - host requirement: the library's host code is built for x86-64-v3 (AVX2, BMI2 -- the sequential SHA-512 weight transcript
 *    runs twice as fast with rorx / mulx); every MI355X host platform has it, an older CPU dies with SIGILL at load, not with
 *    a status code.  (The optional 8-lane multi-buffer hash additionally needs AVX-512 and IS checked at run time.)
 *  - short-Weierstrass suite (AVRF_SUITE_SECP256R1_SHA256_TAI): xy points are the curve's affine points, identity = 64 zero bytes.
 */
#ifndef AVRF_H
#define AVRF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  AVRF_OK = 0,                     /* Ok(())                                   */
  AVRF_VERIFICATION_FAILURE = 1,   /* Error::VerificationFailure               */
  AVRF_INVALID_DATA = 2,           /* Error::InvalidData                       */
  AVRF_RING_CAPACITY_EXCEEDED = 3, /* Error::RingCapacityExceeded              */
  AVRF_SRS_LOOKUP_FAILED = 4,      /* Error::SrsLookupFailed                   */
  AVRF_ERR_NO_DEVICE = -1,         /* no HIP device / kernels not loadable     */
  AVRF_ERR_BAD_ARG = -2
};

enum {
  AVRF_SUITE_BANDERSNATCH_SHA512_ELL2 = 0, /* src/suites/bandersnatch.rs:62-105 */
  AVRF_SUITE_BABYJUBJUB_SHA512_TAI = 1,    /* src/suites/baby_jubjub.rs:56-95   */
  AVRF_SUITE_JUBJUB_SHA512_TAI = 2,        /* src/suites/jubjub.rs:56-95 (ring proofs over BLS12-381) */
  AVRF_SUITE_ED25519_SHA512_TAI = 3,       /* src/suites/ed25519.rs:44-66 (Tiny / Thin / Pedersen; no RingSuite: avrf_ring_* -> BAD_ARG) */
  AVRF_SUITE_BANDERSNATCH_SHAKE128_ELL2 = 5, /* src/suites/bandersnatch_shake128.rs: suite 0's curve, Shake128Transcript, expand_message_xof */
  AVRF_SUITE_TESTING_SHA256_TAI = 6,        /* src/suites/testing.rs: the crate's own test suite (edwards25519, HashTranscript<Sha256>; no ring) */
  AVRF_SUITE_SECP256R1_SHA256_TAI = 7,      /* src/suites/secp256r1.rs:49-70: NIST P-256 (a = -3 short Weierstrass, cofactor 1, 256-bit fields with the
                                             * top bit set), HashTranscript<Sha256>; Tiny / Thin / Pedersen, no RingSuite.  xy points are the curve's own
                                             * affine points, the identity is the all-zero 64 bytes; compressed form: 33 bytes (avrf_point_len) */
  AVRF_SUITE_BANDERSNATCH_SW_SHA512_TAI = 4 /* src/suites/bandersnatch_sw.rs:60-112: Bandersnatch in its short-Weierstrass presentation.
                                             * xy points are the TEMapping (src/utils/te_sw_map.rs) of the suite's SWAffine; the compressed
                                             * form (avrf_points_*, *_wire) is its 33-byte serialize_compressed: see avrf_point_len */
};

typedef struct avrf_ctx avrf_ctx;

/* Library / build identification (no GPU needed). */
const char *avrf_version(void);
/* Number of visible HIP devices (0 when there is none; never initialises a context). */
int avrf_device_count(void);
/* How host threads wait for `device`: on = 1 sleeps in the driver until the stream is done (hipDeviceScheduleBlockingSync), on = 0
 * restores the runtime's default (a yielding spin).  A process that runs many contexts under a CPU quota wants 1: a spinning waiter
 * burns the cores the weight hashes of the other contexts need.  Device-wide, so the caller decides -- once, early, before the
 * contexts are created (that is the tested use); AVRF_OK / AVRF_ERR_NO_DEVICE. */
int avrf_device_set_blocking_sync(int device, int on);

/* One engine instance: suite parameterisation (trait Suite, src/lib.rs:177-250) + one HIP
 * stream + device workspace on `device`. */
int avrf_ctx_create(int suite, int device, avrf_ctx **out);
void avrf_ctx_destroy(avrf_ctx *ctx);

/* Input validation of the verifier entry points (SURVEY.md 8b "Input validation").  The reference's verifiers ASSUME points
 * on the curve and in the prime-order subgroup (src/thin.rs:78-94, src/pedersen.rs:103-120): its typed points have passed
 * CanonicalDeserialize with Validate::Yes / the checked constructors (src/lib.rs:410-433,440-444,471-494).  This ABI takes raw
 * coordinates, so the caller says who validates:
 *   level 0 (default)  "_unchecked": the caller guarantees on-curve subgroup points (e.g. they came from
 *                      avrf_points_decompress(validate = 1) or from an arkworks value); off-curve input gives undefined verdicts;
 *   level 1            every pk / I/O / proof point of avrf_{thin,pedersen}_{verify,batch_verify,batch_run,batch_challenges} must
 *                      be on the curve, else AVRF_INVALID_DATA (per item for the *_verify calls) before any equation is evaluated;
 *   level 2            additionally r * P = O for every point (one 253-bit scalar multiplication per point: opt-in).
 * The identity checks of the verifiers (src/thin.rs:140-149,266-271, src/pedersen.rs:204-213,348-353) are unconditional.
 *
 * PRIME-ORDER-SUBGROUP MEMBERSHIP IS A HARD PRECONDITION of every prove / verify / batch-verify entry point at levels 0 and 1
 * and of the wire flavour with validate = 0 -- exactly where the reference says "the caller must ensure ... otherwise the
 * behaviour is undefined" (src/thin.rs:78-94).  The kernels exploit it: scalars are reduced mod r before they meet a point
 * (one-pair forms (s z) I, (c z) O) and Bandersnatch multiplies through the GLV endomorphism, k P = k1 P + k2 psi(P) (glv.h),
 * both of which equal the reference's literal products only inside the subgroup.  For an on-curve point with a 2- or
 * 4-torsion component the verdict is unspecified and MAY DIFFER from the reference's: a consensus-critical caller that does
 * not hold validated points must use level 2 / validate = 1, which answers AVRF_INVALID_DATA for such a point before any
 * equation (tests/test_gpu_wire.py::test_torsion_points).  avrf_scalar_mul and avrf_msm_te compute literal products for
 * any curve point. */
int avrf_ctx_set_validation(avrf_ctx *ctx, int level);

/* <S::Affine as AffineRepr>::Group::msm_unchecked(&bases, &scalars)
 * (call sites src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411).
 * bases_xy: n x 64, scalars: n x 32, out_xy: normalised result (64 bytes).
 * "unchecked": no subgroup check; off-range coordinates / scalars give AVRF_INVALID_DATA. */
int avrf_msm_te(avrf_ctx *ctx, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t out_xy[64]);
/* The zero-copy flavour of the same call: everything as arkworks keeps it IN MEMORY -- bases n x Affine { x, y },
 * each coordinate an Fp<MontBackend, 4> (four little-endian u64 limbs, Montgomery form, R = 2^256); scalars n x ScalarField in
 * the same form; result Montgomery x || y.  A shim can pass `bases.as_ptr()` / `scalars.as_ptr()` of the slices handed to
 * `msm_unchecked` (the crate's Affine / Fp structs are plain limb arrays; the shim asserts their size: 64 / 32 bytes).
 * A limb value >= the modulus gives AVRF_INVALID_DATA.  The canonical-bytes flavour above remains the tested contract of the
 * other entry points (it is what the reference's vectors pin). */
int avrf_msm_te_mont(avrf_ctx *ctx, size_t n, const uint8_t *bases_mont_xy, const uint8_t *scalars_mont, uint8_t out_mont_xy[64]);

/* <E::G1 as VariableBaseMSM>::msm on the suite's pairing curve (BLS12-381 for Bandersnatch, BN254 for
 * Baby-JubJub) -- the KZG commit/open MSMs inside w3f-ring-proof reached from src/ring.rs:220,404,416,731.
 * bases_xy: n x (2*FQ) bytes, canonical little-endian x || y (FQ = 48 / 32), all-zero = point at infinity;
 * scalars: n x 32 (< r of the pairing curve); out_xy: same point format. */
int avrf_g1_msm(avrf_ctx *ctx, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t *out_xy);

/* thin::BatchVerifier::{new, push*, verify}  (src/thin.rs:188-326).
 * n items; item j has io_counts[j] VRF I/O pairs; ios_xy holds sum(io_counts) pairs as
 * input_xy(64) || output_xy(64); ads holds the concatenated additional-data strings with
 * lengths ad_lens[j]; proofs: R_xy(64) || s(32) per item.
 * Returns AVRF_OK, AVRF_INVALID_DATA (identity pk / io point, or malformed encodings),
 * or AVRF_VERIFICATION_FAILURE.  Empty batch => AVRF_OK (src/thin.rs:262-264). */
int avrf_thin_batch_verify(avrf_ctx *ctx, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy,
                           const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens,
                           const uint8_t *proofs);

/* Two-phase form of the same call for callers that keep a batch resident in HBM
 * (bench.py times avrf_thin_batch_run only): stage copies the batch to the device,
 * run performs prepare + verify on the staged batch and may be called repeatedly. */
int avrf_thin_batch_stage(avrf_ctx *ctx, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy,
                          const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens,
                          const uint8_t *proofs);
int avrf_thin_batch_run(avrf_ctx *ctx);
/* The run of a staged batch (Thin or Pedersen) in three non-overlapping calls, for a host thread that keeps several contexts
 * in flight: BatchVerifier::verify (src/thin.rs:257-325, src/pedersen.rs:341-426) is device work, then the sequential
 * weight transcript on the host (src/thin.rs:274-279), then device work again.
 *   avrf_batch_run_begin  enqueues validation + the prepare kernel + the copies back on the context's stream and returns;
 *   avrf_batch_run_hash   waits for them, returns AVRF_INVALID_DATA as the one-call form would, hashes the weight transcript
 *                         on the calling thread, enqueues the terms kernel and the MSM, returns without waiting;
 *   avrf_batch_run_end    waits for the MSM and returns the verdict (AVRF_OK / AVRF_VERIFICATION_FAILURE).
 * Results are those of avrf_thin_batch_run / avrf_pedersen_batch_run (which are these three in a row).  Between begin and
 * end the run owns the context: EVERY other entry point that takes the context (staging, MSMs, challenges / partials, scalar
 * multiplications, the point codecs, the per-item provers / verifiers, the avrf_ring_* calls of setups created from it)
 * returns AVRF_ERR_BAD_ARG and leaves the run undisturbed; _begin / _hash / _end out of order return AVRF_ERR_BAD_ARG too;
 * an error status from _hash or _end ends the run (the context is idle again). */
int avrf_batch_run_begin(avrf_ctx *ctx);
int avrf_batch_run_hash(avrf_ctx *ctx);
int avrf_batch_run_end(avrf_ctx *ctx);

/* ---- Ownership of host buffers; page-locked memory (SURVEY.md 8b "Ownership").
 * Every entry point copies what it needs out of the caller's buffers before it returns, EXCEPT avrf_pool_submit (below), whose
 * copies are left in flight.  From ordinary (pageable) memory a copy to the device goes through the runtime's bounce buffers
 * at ~10 GB/s of a host core; from page-locked memory it is a DMA transfer that costs no host time and overlaps kernels.
 * avrf_host_alloc returns page-locked memory for the batch buffers (pks / ios / ads / proofs / counts); avrf_host_register
 * pins memory the caller already owns (page-aligned ranges are cheapest; unregister before freeing it). */
int avrf_host_alloc(size_t bytes, void **out);
/* NUMA placement (a two-socket host with several GPUs): the pool's worker threads run on the CPUs of their
 * device's NUMA node, and avrf_host_alloc allocates its pages from the node of the CURRENT device (hipGetDevice) -- both inside
 * the caller's affinity mask, both read from sysfs (/sys/bus/pci/devices/<bdf>/numa_node), both off with AVRF_POOL_NUMA=0.
 * avrf_numa_cpus_of_pci is the lookup itself (node_out: -1 = unknown; returns the CPUs written to cpus_out);
 * avrf_pool_numa_node the node a pool's workers were bound to (-1: not bound). */
int avrf_numa_cpus_of_pci(const char *pci_bdf, const char *sysfs_root_dir, int32_t *cpus_out, size_t cap, int32_t *node_out);
void avrf_host_free(void *p);
int avrf_host_register(void *p, size_t bytes);
int avrf_host_unregister(void *p);

/* ---- avrf_pool: many BatchVerifier::verify jobs in flight on one device (thin::BatchVerifier src/thin.rs:188-326 when
 * kind = 1, pedersen::BatchVerifier src/pedersen.rs:303-426 when kind = 2) -- the form a service that verifies batch after
 * batch wants, and the one bench.py measures.  The reference's verify() is one call on one thread; here a batch is device
 * work, 3.5 ms of a host core for the sequential weight transcript (src/thin.rs:274-279), then 0.6 ms of device work, so
 * throughput comes from overlapping many batches, and host time per batch is what bounds a node with several GPUs.
 *   n_slots    batches that can be staged at once (inputs + transcript records, ~30 MB each at 65 536 items);
 *   n_lanes    streams + MSM workspaces (~150 MB each) the MSM chains run on; waiting batches hold no lane;
 *   lane_depth chains queued behind each other on one lane (0 = 3): a host thread that has just hashed a group of transcripts
 *              enqueues ALL their chains at once and goes back to hashing while the device works through them (few streams,
 *              deep queues: more than ~20 streams in flight slow the device down);
 *   n_threads  native host threads; each owns a share of the slots and lanes and is driven by completion events;
 *   hash_group transcripts a thread hashes TOGETHER: 1 = one scalar SHA-512 chain at a time (lowest latency; right when the
 *              process has >= ~4 cores per GPU), 8 / 16 = the AVX-512 multi-buffer code (3.5-5 x the hashes per core-second at
 *              the price of latency; right for 1-3 cores per GPU; falls back to 1 without AVX-512).
 * avrf_pool_submit hands over one batch (arguments as avrf_thin_batch_stage; pks_xy NULL for kind 2) and returns a ticket
 * at once; THE BUFFERS MUST STAY UNTOUCHED UNTIL avrf_pool_wait HAS RETURNED FOR THAT TICKET (the copies are asynchronous).
 * It blocks while every slot is in flight and returns AVRF_ERR_BAD_ARG when every slot holds an uncollected verdict.
 * avrf_pool_wait blocks until the verdict of `ticket` is out: *status = AVRF_OK / AVRF_VERIFICATION_FAILURE /
 * AVRF_INVALID_DATA exactly as avrf_thin_batch_verify would return for that batch.  avrf_pool_resubmit runs the batch of a
 * collected ticket again (from_host = 0: the copy resident in HBM; 1: staged again from the same host buffers).
 * avrf_pool_cycle is the measurement loop: every slot that holds a collected batch is run again and again by the workers
 * themselves until the region consists of whole blocks of `steps_block` runs and has lasted `min_seconds` (0: one block;
 * max_steps bounds it), no foreign-function call per step; returns the runs made, how many returned something else than
 * `expect_status`, and the time from the first launch to the last verdict.
 * avrf_pool_stats: out[0..4] host-thread CPU microseconds spent in staging + prepare launches / collecting / hashing /
 * terms + MSM launches / folding, out[6] hash groups, out[7] transcripts hashed, out[8] times a thread slept, out[9] / out[10]
 * total milliseconds / launches of the dominant kernel (k_accumulate, HIP events on the lanes' streams); n_out >= 11. */
typedef struct avrf_pool avrf_pool;
int avrf_pool_create(int suite, int device, int kind, int n_slots, int n_lanes, int lane_depth, int n_threads, int hash_group, avrf_pool **out);
void avrf_pool_destroy(avrf_pool *pool);
int avrf_pool_set_validation(avrf_pool *pool, int level);
int avrf_pool_submit(avrf_pool *pool, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                     const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, uint64_t *ticket);
/* avrf_pool_submit_wire: the same job from wire bytes (`serialize_compressed` points; thin proofs R || s, pedersen proofs
 * Yb || R || Ok || s || sb as CanonicalSerialize writes them): decompression, and with validate != 0 the non-identity and
 * prime-order-subgroup checks of Validate::Yes (src/lib.rs:410-433), run on the device while the batch is staged; a point that
 * fails makes the verdict AVRF_INVALID_DATA.  Resubmission and the cycle mode work on such slots as on any other. */
int avrf_pool_submit_wire(avrf_pool *pool, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts,
                          const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int validate, uint64_t *ticket);
int avrf_pool_wait(avrf_pool *pool, uint64_t ticket, int *status);
int avrf_pool_resubmit(avrf_pool *pool, uint64_t ticket, int from_host, uint64_t *new_ticket);
int avrf_pool_cycle(avrf_pool *pool, int from_host, uint64_t steps_block, double min_seconds, uint64_t max_steps, int expect_status,
                    uint64_t *steps_done, uint64_t *mismatches, double *seconds);
int avrf_pool_stats(avrf_pool *pool, int reset, double *out, size_t n_out);
int avrf_pool_numa_node(avrf_pool *pool);

/* pedersen::BatchVerifier split the same way (src/pedersen.rs:341-426): stage the shard with avrf_pedersen_batch_stage,
 * avrf_pedersen_batch_challenges -> n_shard x 16 bytes, avrf_batch_weight_seed(pedersen = 1) over all items
 * (resp = s || sb, 64 bytes per item), avrf_pedersen_batch_partial(seed, first_index) -> the shard's partial point. */
int avrf_pedersen_batch_challenges(avrf_ctx *ctx, uint8_t *c_out);
int avrf_pedersen_batch_partial(avrf_ctx *ctx, const uint8_t seed64[64], uint64_t first_index, uint8_t out_xy[64]);

/* One BatchVerifier split over several GPUs; see ark_vrf_amd/dist.py.
 * The weights of src/thin.rs:274-289 depend on ALL items, so: every rank stages its shard and calls
 * avrf_thin_batch_challenges (per-item c_j, 16 bytes each); the c_j and response scalars are
 * all-gathered; avrf_batch_weight_seed hashes them (host, sequential) into the 64-byte stream seed;
 * avrf_thin_batch_partial returns the MSM of the shard's terms (incl. its share of the G term) under
 * that seed; the partial points are all-gathered and added with avrf_points_sum; the batch verifies
 * iff the sum is the identity (0, 1). */
int avrf_thin_batch_challenges(avrf_ctx *ctx, uint8_t *c_out);
int avrf_batch_weight_seed(int suite, int pedersen, size_t n, const uint8_t *c16, const uint8_t *resp, uint8_t seed_out[64]);
/* The same for `count` <= 8 batches in one pass: the chains of different batches are independent, so their transcripts are
 * hashed together, one batch per lane of an AVX-512 register (host_sha512_mb.h: 3.6 x the hashes per core-second at twice
 * the latency of one scalar chain).  Inside avrf_*_batch_run the same code runs behind AVRF_HASH_THREADS service threads
 * (default 0; for hosts that grant a GPU fewer cores than its contexts have transcripts to hash -- bench.py turns it on
 * from the CPU quota).  seeds_out: count x 64 bytes.  AVRF_ERR_NO_DEVICE when the host CPU lacks AVX-512. */
int avrf_batch_weight_seeds_x8(int suite, int pedersen, int count, const size_t *n, const uint8_t *const *c16, const uint8_t *const *resp, uint8_t *seeds_out);
/* SHA-512 of `count` <= 8 contiguous messages through that multi-buffer code (the form avrf_*_batch_run hands to the service:
 * prefix || records in one buffer); digests_out: count x 64 bytes. */
int avrf_sha512_x8(int count, const uint8_t *const *msgs, const size_t *lens, uint8_t *digests_out);
/* the same for `count` <= 16 messages through the pool's form (two interleaved groups of eight above eight messages) */
int avrf_sha512_x16(int count, const uint8_t *const *msgs, const size_t *lens, uint8_t *digests_out);
/* (AVRF_ERR_BAD_ARG unless avrf_thin_batch_challenges succeeded on the CURRENT staging: re-staging or any other call that
 * stages -- avrf_thin_verify, a prover -- invalidates the challenges) */
int avrf_thin_batch_partial(avrf_ctx *ctx, const uint8_t seed[64], uint64_t first_index, uint8_t out_xy[64]);
int avrf_points_sum(int suite, size_t k, const uint8_t *points_xy, uint8_t out_xy[64]);

/* Exposes the MSM the last avrf_thin_batch_run / avrf_pedersen_batch_run built
 * (bases_xy: n_terms x 64, scalars: n_terms x 32) so tests can compare weights and terms with
 * the oracle bit for bit.  Either output pointer may be NULL; returns the number of terms. */
size_t avrf_batch_last_terms(avrf_ctx *ctx, uint8_t *bases_xy, uint8_t *scalars);

/* Last-call timing breakdown in microseconds (host wall clock):
 * [0] total, [1] device prepare (hash), [2] host weight transcript, [3] scalars, [4] msm, [5] finish.
 * Time spent inside the calls: with avrf_batch_run_begin / _hash / _end the gaps between the three calls are not counted. */
void avrf_last_timing(avrf_ctx *ctx, double out[8]);

/* Device-side timing of the dominant kernel (MSM bucket accumulation), measured with HIP events
 * recorded on the context's stream around every launch since the last reset: total milliseconds,
 * number of launches, and the plan of the last MSM {window bits c, windows, buckets/window, lanes/bucket}. */
int avrf_kernel_stats(avrf_ctx *ctx, int reset, double *accum_ms_total, uint64_t *accum_launches, int32_t plan[4]);

/* ---- Independent items (the per-item Prover / Verifier traits).  Two device forms behind every entry point below, same
 * results: calls of up to 2 048 items with ONE I/O pair each (and a given public key, for the provers) on the twisted-Edwards
 * suites spread an item over 32 lanes -- latency of a call 0.5-1.1 ms whatever n (Thin 0.54 / 0.72 ms, Pedersen 0.52 / 1.03 ms);
 * ONE item per call goes through the MSM engine instead (its equations / its commitments as small MSMs in one launch,
 * the doubling chains folded on the host): Thin 0.28 ms verifying, 0.36 ms proving; Pedersen 0.35 / 0.69 ms; up to 30 pairs; everything else runs one lane per item (2-3.4 ms per call up to 65 536 items: 19-25 M items/s). */

/* thin::Prover::prove for a batch of independent (sk, ios, ad)  (src/thin.rs:111-129).
 * sks: n x 32; pks_xy: the cached `Secret::public` (n x 64) or NULL to derive sk*G on the device;
 * ios as above; proofs_out: n x 96 (R_xy || s). */
int avrf_thin_prove(avrf_ctx *ctx, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
                    const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out);

/* thin::Verifier::verify for a batch of independent items (src/thin.rs:131-165);
 * status_out[j] receives the per-item status.  Returns AVRF_OK when the call itself ran. */
int avrf_thin_verify(avrf_ctx *ctx, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                     const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out);

/* tiny::Prover::prove / tiny::Verifier::verify (src/tiny.rs:163-214) for batches of independent items: the Thin construction
 * under scheme tag 0x00 with the challenge kept in the proof, proof = LE16(c) || LE32(s) = 48 bytes (Proof's compressed
 * serialisation, src/tiny.rs:60-78).  No batch verifier exists for this scheme (src/tiny.rs:1-6).  Arguments as for
 * avrf_thin_prove / avrf_thin_verify; proofs(_out): n x 48. */
int avrf_tiny_prove(avrf_ctx *ctx, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
                    const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out);
int avrf_tiny_verify(avrf_ctx *ctx, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                     const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out);

/* pedersen::BatchVerifier::{new, push*, verify} (src/pedersen.rs:303-426).
 * proofs: Yb_xy(64) || R_xy(64) || Ok_xy(64) || s(32) || sb(32) = 256 bytes per item. */
int avrf_pedersen_batch_verify(avrf_ctx *ctx, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                               const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs);
int avrf_pedersen_batch_stage(avrf_ctx *ctx, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                              const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs);
int avrf_pedersen_batch_run(avrf_ctx *ctx);

/* pedersen::Prover::prove (src/pedersen.rs:136-186) for a batch; pks_xy as for avrf_thin_prove;
 * proofs_out: n x 256, blindings_out: n x 32 (may be NULL). */
int avrf_pedersen_prove(avrf_ctx *ctx, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
                        const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out,
                        uint8_t *blindings_out);

/* pedersen::Verifier::verify (src/pedersen.rs:188-249) for a batch of independent items. */
int avrf_pedersen_verify(avrf_ctx *ctx, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                         const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out);

/* ---- Ring VRF (src/ring.rs).  Handles own device-resident state; free them with *_free. ----
 *
 * RingSetup::from_pcs_params (src/ring.rs:380-393) on an arkworks `URS` file (`serialize_uncompressed`,
 * e.g. data/srs/bls12-381-srs-2-11-uncompressed-zcash.bin): parses and uploads the first 3N+1 G1 powers,
 * keeps the two G2 powers, builds the PIOP domain for `ring_size` (src/ring.rs:810-843).
 * Returns AVRF_RING_CAPACITY_EXCEEDED when the SRS is too short for the ring size. */
typedef struct avrf_ring_setup avrf_ring_setup;
typedef struct avrf_ring_key avrf_ring_key;
int avrf_ring_setup_load(avrf_ctx *ctx, const uint8_t *srs, size_t srs_len, size_t ring_size, avrf_ring_setup **out);
void avrf_ring_setup_free(avrf_ring_setup *setup);
/* Kzg::setup as reached from RingSetup::from_rand / from_seed (src/ring.rs:359-374), with the trapdoor tau (32 B LE,
 * < r) and the generators given explicitly (g1: one `powers_in_g1` entry, g2: one `powers_in_g2` entry, both in the
 * URS serialize_uncompressed encoding, e.g. the first entries of an existing SRS file).  Writes the URS bytes
 * { tau^i g1, i < n_g1 ; g2, tau g2 } that avrf_ring_setup_load reads; *out_len = bytes needed (also when out is too
 * small -> AVRF_ERR_BAD_ARG).  n_g1 = avrf_ring_pcs_domain_size(suite, ring_size) for a ring of that size.
 * Insecure by construction (the caller knows tau) -- exactly like the reference's from_seed; for tests and benches. */
int avrf_ring_srs_generate(avrf_ctx *ctx, const uint8_t *tau, const uint8_t *g1, const uint8_t *g2, size_t n_g1,
                           uint8_t *out, size_t out_cap, size_t *out_len);
/* RingSetup::from_seed(ring_size, seed) (src/ring.rs:359-366): the deterministic setup of a 32-byte seed as the reference derives it --
 * Transcript::new(SUITE_ID), absorb_raw(seed), to_rng() (src/utils/transcript.rs:61-92), then Kzg::setup = URS::generate: tau = Fr::rand,
 * g1 = G1::rand, g2 = G2::rand from that stream -- so that two parties sharing only the seed hold the same SRS (README.md:181 of the
 * reference).  PARITY UNPINNED: the reference holds no vector of a seeded setup; the samplers (ark-ff Fp::rand, ark-ec
 * Projective::rand, rand 0.8 bool, w3f-pcs draw order) are restated from their published code, here and in oracle/ring_py.py
 * srs_from_seed, and the two restatements are tested against each other.  Insecure by construction (the seed gives tau), like the
 * reference's. */
int avrf_ring_setup_from_seed(avrf_ctx *ctx, const uint8_t *seed, size_t ring_size, avrf_ring_setup **out);
/* (avrf_ring_setup_load also accepts the serialize_compressed form of the same object: G1 / G2 points are decompressed on
 * the host, curve membership checked, no G2 subgroup check -- the SRS is trusted-setup material, src/ring.rs:466-474.)
 * CanonicalSerialize for RingSetup (= its PcsParams, truncated to the 3N+1 powers the setup keeps; src/ring.rs:484-521) and
 * for RingBuilderPcsParams (Vec<G1Affine>: the SRS in Lagrangian form L_i(tau) g1, i < N; src/ring.rs:523-529, obtained in the
 * reference from RingSetup::verifier_key_builder), compress = 0 / 1 for ark-serialize's two modes.  *out_len = bytes needed
 * (also when `out` is NULL or too small -> AVRF_ERR_BAD_ARG). */
int avrf_ring_setup_serialize(avrf_ring_setup *setup, int compress, uint8_t *out, size_t out_cap, size_t *out_len);
/* Verifier-only setup (src/ring.rs:466-482, verifier_key_from_commitment: "verifier-only users: no SRS required"):
 * avrf_ring_pcs_verifier_params_serialize writes RingSetup::pcs_verifier_params() (src/ring.rs:435) =
 * RawKzgVerifierKey { g1, g2, tau_in_g2 } in either ark-serialize mode; avrf_ring_verifier_setup_load builds from those bytes
 * (points validated) a setup handle that serves avrf_ring_batch_verify / avrf_ring_verify_each / avrf_ring_vrf_verify /
 * avrf_ring_pairing_check and nothing that needs the SRS (index, prove, builder, setup serialisation ->
 * AVRF_SRS_LOOKUP_FAILED). */
int avrf_ring_pcs_verifier_params_serialize(avrf_ring_setup *setup, int compress, uint8_t *out, size_t out_cap, size_t *out_len);
int avrf_ring_verifier_setup_load(avrf_ctx *ctx, const uint8_t *params, size_t params_len, size_t ring_size, avrf_ring_setup **out);
int avrf_ring_builder_params_serialize(avrf_ring_setup *setup, int compress, uint8_t *out, size_t out_cap, size_t *out_len);
size_t avrf_ring_pcs_domain_size(int suite, size_t ring_size);   /* pcs_domain_size, src/ring.rs:810-817 */
size_t avrf_ring_max_ring_size(const avrf_ring_setup *setup);   /* RingContext::max_ring_size, src/ring.rs:298-300 */
size_t avrf_ring_domain_size(const avrf_ring_setup *setup);     /* piop_domain_size, src/ring.rs:819-821 */
size_t avrf_ring_proof_len(const avrf_ring_setup *setup);       /* 592 (BLS12-381) / 480 (BN254) */
size_t avrf_ring_commitment_len(const avrf_ring_setup *setup);  /* 144 / 96 */
/* the suite a setup was loaded for (-1: NULL) and the setup a prover key was indexed on (owned by the caller as before) */
int avrf_ring_setup_suite(const avrf_ring_setup *setup);
avrf_ring_setup *avrf_ring_key_setup(const avrf_ring_key *key);
/* how the prover's KZG commitments are laid out as fixed-base MSMs (for op counts in reports): out = { window bits and rows of
 * the SRS window table, window bits and rows of the Lagrange-basis witness table (0 until the first proof builds it) }.  A prove call of
 * 64 or more proofs builds, once per device and SRS, the tables of ALL multiples of both base sets in the HBM that is free (DESIGN.md;
 * AVRF_RING_TABLE_GB, default 232, AVRF_RING_DIRECT=0 to keep the bucket form); from then on the figures are those tables'. */
int avrf_ring_setup_plan(const avrf_ring_setup *setup, int32_t out[4]);

/* RingSetup::prover_key / verifier_key -> ring_proof::index (src/ring.rs:399-417): fixed columns of the
 * ring `pks_xy` (n_keys x 64) and their three KZG commitments.  commitment_out (may be NULL) receives the
 * RingCommitment in its compressed serialisation (3 x G1), i.e. the `ring_pks_com` of the reference vectors.
 * AVRF_RING_CAPACITY_EXCEEDED if n_keys > max ring size. */
int avrf_ring_index(avrf_ring_setup *setup, const uint8_t *pks_xy, size_t n_keys, avrf_ring_key **out, uint8_t *commitment_out);
void avrf_ring_key_free(avrf_ring_key *key);

/* VerifierKeyBuilder (src/ring.rs:539-637): incremental construction of a ring commitment.  `new` = the ring of padding
 * points (VerifierKeyBuilder::new; the Lagrange-basis SRS the reference passes as RingBuilderPcsParams / SrsLookup is
 * derived from the setup's SRS on the device), `append` adds keys in order (AVRF_RING_CAPACITY_EXCEEDED and nothing
 * appended when they do not fit, AVRF_INVALID_DATA for a non-canonical coordinate), `finalize` writes the compressed
 * RingCommitment -- equal to avrf_ring_index's for the same key list. */
typedef struct avrf_ring_vk_builder avrf_ring_vk_builder;
int avrf_ring_vk_builder_new(avrf_ring_setup *setup, avrf_ring_vk_builder **out);
void avrf_ring_vk_builder_free(avrf_ring_vk_builder *builder);
size_t avrf_ring_vk_builder_free_slots(const avrf_ring_vk_builder *builder);          /* src/ring.rs:584-586 */
int avrf_ring_vk_builder_append(avrf_ring_vk_builder *builder, const uint8_t *pks_xy, size_t n);   /* :598-622 */
int avrf_ring_vk_builder_finalize(const avrf_ring_vk_builder *builder, uint8_t *commitment_out);   /* :625-627 */

/* RingProver::prove (the `ring_prover.prove(blinding)` half of ring::Prover::prove, src/ring.rs:219-221) for n
 * proofs over one ring: key_index[i] is the prover's position in the ring, blindings[i] the secret blinding
 * returned by avrf_pedersen_prove.  blinding_mode 0 = RingContext::new_without_blinding (deterministic,
 * reproduces the reference vectors); 1 = hiding (the reference's default RingContext): the last 3 rows of every
 * witness column are fresh uniformly random field elements (getrandom(2)), so proofs are not reproducible.  proofs_out: n x avrf_ring_proof_len bytes (the RingBareProof in its
 * compressed serialisation); the full ring-VRF proof is the Pedersen proof followed by it (src/ring.rs:160-166). */
int avrf_ring_prove(avrf_ring_key *key, size_t n, const uint32_t *key_index, const uint8_t *blindings, int blinding_mode,
                    uint8_t *proofs_out);

/* RingVerifier::verify (n = 1) and the multi-ring RingBatchVerifier::{push, verify} (src/ring.rs:242,682-735):
 * verifies n bare ring proofs (compressed serialisation, avrf_ring_proof_len bytes each) for the key
 * commitments instances_xy (n x 64: the Pedersen proof's Yb, src/ring.rs:237-241) against ring commitments
 * (n_rings x avrf_ring_commitment_len, cf. verifier_key_from_commitment src/ring.rs:477-482); ring_of_item[i]
 * selects the ring of item i (NULL = all items use ring 0).  One randomised check: two G1 MSMs on the GPU and
 * a 2-pairing check.  A complete ring-VRF (batch) verification is avrf_pedersen_(batch_)verify on the Pedersen
 * halves plus this call (src/ring.rs:236-242,729-735).  Returns AVRF_OK / AVRF_VERIFICATION_FAILURE /
 * AVRF_INVALID_DATA (undecodable point or scalar). */
int avrf_ring_batch_verify(avrf_ring_setup *setup, size_t n, const uint8_t *ring_commitments, size_t n_rings,
                           const uint32_t *ring_of_item, const uint8_t *instances_xy, const uint8_t *ring_proofs);

/* n x ring::Verifier::verify (src/ring.rs:228-247, the ring half): the same inputs as avrf_ring_batch_verify, but every proof is
 * checked on its own and gets its own status (status_out[i] = AVRF_OK / AVRF_VERIFICATION_FAILURE / AVRF_INVALID_DATA), so a
 * bad proof is identified instead of failing the batch: per proof two small G1 linear combinations and one 2-pairing check,
 * all on the device (Miller loops + final exponentiations of all proofs in one kernel, pairing.hip).  Few proofs are latency
 * cases -- one wave of the pairing kernel needs 11.7 ms whatever it carries -- so up to 16 checks are finished on the host pool
 * from the device's G1 sums (tabulated G2 lines, 1.0 ms per check and core), and n = 1 runs as a batch of one (1.7 ms; the
 * reference: 3.24 ms).  Same statuses either way.  A ring commitment that does not decode fails the call (AVRF_INVALID_DATA). */
int avrf_ring_verify_each(avrf_ring_setup *setup, size_t n, const uint8_t *ring_commitments, size_t n_rings, const uint32_t *ring_of_item,
                          const uint8_t *instances_xy, const uint8_t *ring_proofs, int32_t *status_out);

/* The pairing half of the KZG verifier on the device (arkworks `Pairing::multi_pairing` + final exponentiation as reached
 * from RingVerifier::verify, src/ring.rs:242): for i < n,  ok_out[i] = 1 iff  e(A_i, g2) * e(B_i, tau g2) == 1  with (g2, tau g2)
 * the two `powers_in_g2` of the setup's SRS.  a_xy / b_xy: n G1 points each, canonical little-endian x || y (48+48 bytes
 * BLS12-381, 32+32 BN254; all-zero = infinity), assumed on the curve (they are outputs of the verifier's own MSMs).  Miller
 * loops and final exponentiations run as one kernel, an Fp12 element spread over 16 lanes (pairing.hip); the line tables of
 * the two G2 arguments are built on the device at first use (up to 16 checks: on the host pool, see avrf_ring_verify_each).
 * AVRF_INVALID_DATA for a coordinate >= p. */
int avrf_ring_pairing_check(avrf_ring_setup *setup, size_t n, const uint8_t *a_xy, const uint8_t *b_xy, int32_t *ok_out);

/* Input::new(data) = Suite::data_to_point (src/lib.rs:440-444 via src/utils/hash_to_curve.rs:34-100) for n messages:
 * Elligator2 with expand_message_xmd(SHA-512) for Bandersnatch, try-and-increment for Baby-JubJub.  data = the
 * messages concatenated, data_lens[i] their lengths.  out_xy: n x 64; status_out[i] = 0, or 2 (InvalidData) where
 * try-and-increment found no point (the reference returns None). */
int avrf_hash_to_curve(avrf_ctx *ctx, size_t n, const uint8_t *data, const uint32_t *data_lens, uint8_t *out_xy, int32_t *status_out);

/* CanonicalSerialize / CanonicalDeserialize of curve points, batched on the device
 * (ark-serialize compressed form; checked constructors src/lib.rs:410-494).
 * decompress: in n x L -> out n x 64; status_out[j] = AVRF_OK / AVRF_INVALID_DATA; L = avrf_point_len(suite) = 32 for the
 * twisted-Edwards suites (LE32(y), sign of x in bit 255), 33 for the short-Weierstrass presentation (LE32(x) || flags; the
 * xy side is then the TEMapping of the point, src/utils/te_sw_map.rs:32-47).
 * validate != 0 additionally requires prime-order-subgroup membership and non-identity. */
size_t avrf_point_len(int suite);
int avrf_points_decompress(avrf_ctx *ctx, size_t n, const uint8_t *in, uint8_t *out_xy, int validate, int32_t *status_out);
/* Output::hash::<N> (src/lib.rs:605-609 -> Suite::point_to_hash, src/utils/common.rs:290-305): the VRF output bytes `beta` of n
 * output points, hash_len = N <= 64 bytes each.  Secret::from_seed (src/lib.rs:346-369) for n 32-byte seeds: the secret scalars
 * (LE32) and, when pks_xy_out is not NULL, their public keys (Secret::from_scalar).  The device copies of the seeds and scalars are
 * zeroed before the call returns (the reference zeroizes seed and sk, src/lib.rs:367-368); sks_out is the caller's to scrub. */
int avrf_output_hash(avrf_ctx *ctx, size_t n, const uint8_t *points_xy, size_t hash_len, uint8_t *out);
int avrf_secret_from_seed(avrf_ctx *ctx, size_t n, const uint8_t *seeds, uint8_t *sks_out, uint8_t *pks_xy_out);
int avrf_points_compress(avrf_ctx *ctx, size_t n, const uint8_t *in_xy, uint8_t *out);

/* Secret::from_scalar / Secret::output: sk*G and sk*I for a batch (src/lib.rs:331-334,391-393). */
int avrf_scalar_mul_base(avrf_ctx *ctx, size_t n, const uint8_t *sks, uint8_t *out_xy);
int avrf_scalar_mul(avrf_ctx *ctx, size_t n, const uint8_t *scalars, const uint8_t *points_xy, uint8_t *out_xy);

/* ---- Wire-format flavour: the reference's `serialize_compressed` encodings and a `validate` flag. ----
 * Points are L-byte compressed, L = avrf_point_len(suite) (pks: n x L; ios: per pair input(L) || output(L)); proofs are the
 * reference's byte strings: thin R(L) || s(32) (src/thin.rs:43-48), tiny c(16) || s(32) (src/tiny.rs:60-78), pedersen
 * Yb || R || Ok || s || sb = 3 L + 64 (src/pedersen.rs:69-75), ring-VRF = pedersen proof || ring proof = 752 / 640 at L = 32,
 * 755 for Bandersnatch-SW (src/ring.rs:160-166).
 * validate = 0: CanonicalDeserialize with Validate::No (the point must decode, i.e. lie on the curve);
 * validate = 1: Validate::Yes as in the checked constructors (src/lib.rs:410-433): also prime-order subgroup and not the identity.
 * A point that fails gives AVRF_INVALID_DATA (for its item in the per-item calls) before any equation is evaluated.
 * These calls decompress on the device (avrf_points_decompress) and then run the xy entry points above: the xy flavour is
 * the fast path for callers that hold deserialised points, as the reference's own benches do (benches/thin.rs:46-90). */
/* Staging from wire bytes for the three-call run (avrf_*_batch_stage_wire, then avrf_thin_batch_run / avrf_pedersen_batch_run or
 * avrf_batch_run_begin / _hash / _end): every point is decompressed (and validated) on the device straight into the context's
 * staged buffers; AVRF_INVALID_DATA when a point does not decode / fails validation (nothing is staged then).  The *_batch_verify_wire
 * entry points below are exactly stage_wire + run. */
int avrf_thin_batch_stage_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                               const uint32_t *ad_lens, const uint8_t *proofs, int validate);
int avrf_pedersen_batch_stage_wire(avrf_ctx *ctx, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                   const uint32_t *ad_lens, const uint8_t *proofs, int validate);
int avrf_thin_batch_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                const uint32_t *ad_lens, const uint8_t *proofs, int validate);
int avrf_thin_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                          const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out);
int avrf_tiny_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                          const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out);
int avrf_pedersen_batch_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                    const uint32_t *ad_lens, const uint8_t *proofs, int validate);
int avrf_pedersen_verify_wire(avrf_ctx *ctx, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                              const uint32_t *ad_lens, const uint8_t *proofs, int validate, int32_t *status_out);

/* ring::Prover::prove (src/ring.rs:211-226) as ONE call for n provers of the ring behind `key` (avrf_pedersen_prove +
 * avrf_ring_prove + serialisation): sks n x 32, key_index[i] = position of prover i's key in the ring, ios_xy as for
 * avrf_pedersen_prove; proofs_out: n x (160 + ring_proof_len) bytes, ring::Proof's compressed serialisation.
 * ring_proof_len must equal avrf_ring_proof_len(avrf_ring_key_setup(key)) -- it states the caller's buffer layout and is
 * checked, not trusted (AVRF_ERR_BAD_ARG otherwise, also when ctx and the key's setup are of different suites). */
int avrf_ring_vrf_prove(avrf_ctx *ctx, avrf_ring_key *key, size_t ring_proof_len, size_t n, const uint8_t *sks, const uint32_t *key_index,
                        const uint8_t *ios_xy, const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, int blinding_mode,
                        uint8_t *proofs_out);
/* ring::Verifier::verify for every item (each != 0: status_out[i] per proof; src/ring.rs:228-247) or ring::BatchVerifier over
 * all items (each == 0: the return value is the batch's status; src/ring.rs:693-735).  ios wire-format as above; proofs: n x
 * (160 + avrf_ring_proof_len(setup)).  Pedersen half on the context, ring half on the setup (same suite, else AVRF_ERR_BAD_ARG). */
int avrf_ring_vrf_verify(avrf_ctx *ctx, avrf_ring_setup *setup, size_t n, const uint8_t *ring_commitments, size_t n_rings, const uint32_t *ring_of_item,
                         const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs,
                         int validate, int each, int32_t *status_out);

#ifdef __cplusplus
}
#endif
#endif
