// capi.hip -- the C ABI of libavrf.so (include/avrf.h): contexts, staging, orchestration.
//
// Host-side counterpart of the reference's scheme layer for the accelerated path
// (src/thin.rs:188-326 BatchVerifier; src/pedersen.rs:303-426).  All group/field work is
// launched on the context's HIP stream; the host only (1) runs the sequential weight
// transcript (host_sha512.h), (2) finishes the MSM's O(256)-step window Horner (host_te.h).
#include "capi_internal.h"
#include "host_sha512.h"
#include "host_shake128.h"
#include "host_sha256.h"
#include "host_sha512_mb.h"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include "proto_dev.h"
#include "vrf_batch.h"
#include "suite_dispatch.h"
#include <chrono>
#include <stdlib.h>
#include <string.h>

using namespace avrf;

namespace avrf {
double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace avrf

// A run opened by avrf_batch_run_begin owns the context's stream, staged buffers and MSM workspace until avrf_batch_run_end
// (or an error) closes it: every other entry point that touches them refuses with AVRF_ERR_BAD_ARG meanwhile (include/avrf.h).
static inline bool ctx_busy(const avrf_ctx *c) { return c->run_phase != 0; }

// fixed-base tables for the provers: built once per context, on the context's stream
static int ensure_fixed(avrf_ctx *c) {
  if (c->fixed_ready) return AVRF_OK;
  if (c->d_fixed.ensure((size_t)2 * 32 * 256 * sizeof(te_pre_raw)) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  launch_fixed_table(c->suite, c->d_fixed.as<te_pre_raw>(), c->stream);
  c->fixed_ready = true;
  return AVRF_OK;
}

static BatchDev batch_of(avrf_ctx *c) {
  BatchDev b;
  b.pks_xy = c->d_pks.as<uint8_t>(); b.ios_xy = c->d_ios.as<uint8_t>(); b.io_off = c->d_io_off.as<uint32_t>();
  b.ads = c->d_ads.as<uint8_t>(); b.ad_off = c->d_ad_off.as<uint32_t>(); b.proofs = c->d_proofs.as<uint8_t>();
  b.sks = c->d_sks.as<uint8_t>(); b.n = (uint32_t)c->n;
  b.fixed = (const te_pre *)c->d_fixed.p;
  b.tabs = (te_ext *)c->d_tabs.p; b.first = 0;     // per-item window tables (sized by per_item_chunks)
  b.weights = nullptr; b.records = nullptr;
  return b;
}

// The independent per-item kernels (vrf_single.hip) keep ITEM_TAB_SLOTS window-table entries of 128 bytes per item (5 KB) in
// the context's workspace.  A call of any size walks its items in chunks of at most ITEM_CHUNK (two residency rounds of the
// chip: 65 536 lanes at one wave per SIMD), launched back to back on the context's stream, so the workspace is bounded by
// ITEM_CHUNK x 5 KB = 671 MB whatever n is; `launch` receives the BatchDev of one chunk (first .. n).
static constexpr size_t ITEM_CHUNK = 131072;
template <class F> static int per_item_chunks(avrf_ctx *c, bool with_pks, F launch) {
  if (int fs = ensure_fixed(c)) return fs;
  const size_t chunk = c->n < ITEM_CHUNK ? c->n : ITEM_CHUNK;
  if (c->d_tabs.ensure(chunk * (size_t)ITEM_TAB_SLOTS * sizeof(te_ext_raw)) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  BatchDev b = batch_of(c);
  if (!with_pks) b.pks_xy = nullptr;
  for (size_t i = 0; i < c->n; i += chunk) {
    b.first = (uint32_t)i; b.n = (uint32_t)(c->n - i < chunk ? c->n : i + chunk);
    launch(b);
  }
  return AVRF_OK;
}

// Validate::Yes over every point of the staged batch (pk, I/O pairs, proof points) when the context asks for it;
// rec_status (device, n x int32) receives 2 for items with a bad point (per-item verifiers), else NULL.
static void validate_staged(avrf_ctx *c, int kind, int32_t *d_rec_status) {
  if (c->validate <= 0 || !c->n) return;
  uint32_t *fl = c->d_flags.as<uint32_t>();
  const uint32_t n = (uint32_t)c->n;
  if (kind != 2 && c->d_pks.p) launch_validate_xy(c->suite, c->d_pks.as<uint8_t>(), 64, 1, n, c->validate, fl, d_rec_status, c->stream);
  if (kind == 1) launch_validate_xy(c->suite, c->d_proofs.as<uint8_t>(), 96, 1, n, c->validate, fl, d_rec_status, c->stream);
  else if (kind == 2) launch_validate_xy(c->suite, c->d_proofs.as<uint8_t>(), 256, 3, n, c->validate, fl, d_rec_status, c->stream);
  if (c->tot_io) {
    if (!d_rec_status) launch_validate_xy(c->suite, c->d_ios.as<uint8_t>(), 128, 2, (uint32_t)c->tot_io, c->validate, fl, nullptr, c->stream);
    else {   // per-item status: the I/O pairs of item j are records io_off[j] .. io_off[j+1]; uniform M = 1 is the common case
      const uint32_t *io_off = c->h_io.as<uint32_t>();
      bool uniform = c->tot_io == c->n;
      for (size_t j = 0; uniform && j < c->n; j++) uniform = io_off[j] == j;
      if (uniform) launch_validate_xy(c->suite, c->d_ios.as<uint8_t>(), 128, 2, n, c->validate, fl, d_rec_status, c->stream);
      else launch_validate_xy(c->suite, c->d_ios.as<uint8_t>(), 128, 2, (uint32_t)c->tot_io, c->validate, fl, d_rec_status, c->stream,
                              c->d_io_off.as<uint32_t>(), n);     // one launch: every lane looks its item up in the staged offsets
    }
  }
}

namespace avrf {
int ctx_create(int suite, int device, bool lane_owner, avrf_ctx **out) {
  if (!out || suite < 0 || suite >= AVRF_N_SUITES) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return AVRF_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  avrf_ctx *c = new avrf_ctx();
  c->suite = suite; c->device = device; c->lane_owner = lane_owner;
  if (lane_owner) {
    if (hipStreamCreateWithFlags(&c->own.stream, hipStreamNonBlocking) != hipSuccess) { delete c; return AVRF_ERR_NO_DEVICE; }
    c->stream = c->own.stream;
  }
  if (c->d_flags.ensure(64) != hipSuccess || c->h_flags.ensure(64) != hipSuccess) { avrf_ctx_destroy(c); return AVRF_ERR_NO_DEVICE; }
  *out = c;
  return AVRF_OK;
}
}  // namespace avrf

extern "C" {

int avrf_ctx_set_validation(avrf_ctx *c, int level) {
  if (!c || level < 0 || level > 2) return AVRF_ERR_BAD_ARG;
  c->validate = level;
  return AVRF_OK;
}

// accessors for the other translation units of the library (ring.hip)
hipStream_t avrf_ctx_stream_(avrf_ctx *c) { return c->stream; }
int avrf_ctx_suite_(avrf_ctx *c) { return c->suite; }
int avrf_ctx_device_(avrf_ctx *c) { return c->device; }
int avrf_ctx_busy_(avrf_ctx *c) { return c && ctx_busy(c); }

const char *avrf_version(void) { return "avrf 0.3 (gfx950; tiny/thin/pedersen/ring VRF over Bandersnatch, Baby-JubJub, JubJub; device pairing)"; }

int avrf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int avrf_device_set_blocking_sync(int device, int on) {
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AVRF_ERR_NO_DEVICE; }
  const hipError_t e = hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) (void)hipGetLastError();          // the runtime's last-error slot is sticky: later calls of this thread poll it
  return e == hipSuccess ? AVRF_OK : AVRF_ERR_NO_DEVICE;
}

int avrf_ctx_create(int suite, int device, avrf_ctx **out) { return ctx_create(suite, device, true, out); }

void avrf_ctx_destroy(avrf_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->lane_owner) (void)hipStreamSynchronize(c->own.stream);
  DevBuf *bufs[] = {&c->d_pks, &c->d_ios, &c->d_io_off, &c->d_ads, &c->d_ad_off, &c->d_proofs, &c->d_sks, &c->d_c, &c->d_z,
                    &c->d_flags, &c->d_misc, &c->d_out, &c->d_status, &c->d_fixed, &c->d_weights, &c->d_rec, &c->d_tabs};
  for (DevBuf *b : bufs) b->release();
  c->h_c.release(); c->h_flags.release(); c->h_io.release(); c->h_msg.release();
  c->own.release();                                                    // stream + workspace (nothing for a pool slot)
  delete c;
}

static int finish_point(avrf_ctx *c, const HostExt &r, uint8_t out_xy[64]) {
  with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; HostTe<S>::to_affine_bytes(r, out_xy); });
  return AVRF_OK;
}
static bool point_is_identity(avrf_ctx *c, const HostExt &r) {
  return with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; return HostTe<S>::is_identity(r); });
}
static bool scalar_in_range(int suite, const uint8_t *s) {
  H256 v; memcpy(v.l, s, 32);
  return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return !HostField<typename S::Fr>::geq_p(v); });
}

int avrf_msm_te(avrf_ctx *c, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t out_xy[64]) {
  if (!c || !out_xy || (n && (!bases_xy || !scalars))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  for (size_t i = 0; i < n; i++) if (!scalar_in_range(c->suite, scalars + 32 * i)) return AVRF_INVALID_DATA;
  HostExt r;
  if (n) {
    HIP_TRY(c->d_misc.ensure(n * 64)); HIP_TRY(c->L->d_scalars.ensure(n * 32)); HIP_TRY(c->L->d_pre.ensure(n * sizeof(te_pre_raw)));
    HIP_TRY(hipMemcpyAsync(c->d_misc.p, bases_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->L->d_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    launch_pre_from_affine(c->suite, c->d_misc.as<uint8_t>(), n, c->L->d_pre.as<te_pre_raw>(), c->d_flags.as<uint32_t>(), 0, c->stream);
    HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  }
  c->staged_kind = 0;
  if (int e = guarded([&] { return msm_te_device(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), n, c->L->ws, c->stream, &r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  if (n && *c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
  return finish_point(c, r, out_xy);
}

// The same MSM on arkworks' IN-MEMORY values (SURVEY.md 8b "zero-copy option"): bases = n x Affine { x, y } with each coordinate
// an Fp<MontBackend, 4> (four little-endian u64 limbs, Montgomery form R = 2^256), scalars = n x ScalarField in the same form
// -- what `msm_unchecked(&[Affine], &[ScalarField])` is handed (src/thin.rs:319) -- and the result as Montgomery x || y.  The
// canonical <-> Montgomery conversions of avrf_msm_te disappear on the host side; the device converts the scalars (one
// multiplication each) and takes the bases as they are.
int avrf_msm_te_mont(avrf_ctx *c, size_t n, const uint8_t *bases_mont_xy, const uint8_t *scalars_mont, uint8_t out_mont_xy[64]) {
  if (!c || !out_mont_xy || (n && (!bases_mont_xy || !scalars_mont))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HostExt r;
  if (n) {
    HIP_TRY(c->d_misc.ensure(n * 64)); HIP_TRY(c->L->d_scalars.ensure(n * 32)); HIP_TRY(c->L->d_pre.ensure(n * sizeof(te_pre_raw)));
    HIP_TRY(hipMemcpyAsync(c->d_misc.p, bases_mont_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->L->d_scalars.p, scalars_mont, n * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    launch_pre_from_affine(c->suite, c->d_misc.as<uint8_t>(), n, c->L->d_pre.as<te_pre_raw>(), c->d_flags.as<uint32_t>(), 0, c->stream, 1);
    launch_scalars_from_mont(c->suite, c->L->d_scalars.as<uint32_t>(), n, c->d_flags.as<uint32_t>(), c->stream);
    HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  }
  c->staged_kind = 0;
  if (int e = guarded([&] { return msm_te_device(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), n, c->L->ws, c->stream, &r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  if (n && *c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
  uint8_t canon[64];
  finish_point(c, r, canon);
  with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; using Fq = HostField<typename S::Fq>;
    Fq::store_le(out_mont_xy, Fq::to_mont(Fq::load_le(canon))); Fq::store_le(out_mont_xy + 32, Fq::to_mont(Fq::load_le(canon + 32))); });
  return AVRF_OK;
}

// G1 MSM of the suite's pairing curve (KZG commit / open): bases as canonical little-endian x || y
// (48+48 bytes BLS12-381, 32+32 bytes BN254; all-zero = infinity), scalars LE32 < r.
int avrf_g1_msm(avrf_ctx *c, size_t n, const uint8_t *bases_xy, const uint8_t *scalars, uint8_t *out_xy) {
  if (!c || !out_xy || (n && (!bases_xy || !scalars))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  const int curve = pairing_curve_of(c->suite);
  const size_t fqb = curve == 0 ? 48 : 32;
  for (size_t i = 0; i < n; i++) {
    H256 v; memcpy(v.l, scalars + 32 * i, 32);
    bool ok = curve == 0 ? !HostField<FqBandersnatch>::geq_p(v) : !HostField<FqBabyJubJub>::geq_p(v);   // Fr of the pairing curve
    if (!ok) return AVRF_INVALID_DATA;
  }
  c->staged_kind = 0;
  if (n) {
    HIP_TRY(c->d_misc.ensure(n * 2 * fqb)); HIP_TRY(c->L->d_scalars.ensure(n * 32)); HIP_TRY(c->L->d_pre.ensure(n * 2 * fqb));
    HIP_TRY(hipMemcpyAsync(c->d_misc.p, bases_xy, n * 2 * fqb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->L->d_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    launch_g1_bases(curve, c->d_misc.as<uint8_t>(), n, c->L->d_pre.as<uint32_t>(), c->d_flags.as<uint32_t>(), c->stream);
    HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  }
  if (int e = guarded([&] { return msm_g1_device(curve, c->L->d_pre.as<uint32_t>(), c->L->d_scalars.as<uint32_t>(), n, c->L->ws, c->stream, out_xy) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  if (n && *c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
  return AVRF_OK;
}

// ---------------------------------------------------------------- staging

// kind: 1 thin (pks + 96-byte proofs), 2 pedersen (256-byte proofs); proofs/pks/sks may be NULL for provers
// wait = false (pool.hip): the copies are left in flight on c->stream -- from buffers of avrf_host_alloc they are DMA transfers
// that cost no host time; the caller's buffers must stay untouched until the batch's verdict is out
}  // extern "C"
namespace avrf {
static bool suite_host_weights(int suite) { return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)S::HOST_WEIGHTS; }); }
int ctx_stage(avrf_ctx *c, int kind, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
              const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, bool wait) {
  if (!c || c->run_phase) return AVRF_ERR_BAD_ARG;              // (a run in flight owns the staged buffers)
  if (n && (!io_counts || !ad_lens)) return AVRF_ERR_BAD_ARG;
  if (n > 0x0fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0; c->n = n; c->tot_io = 0; c->n_terms = 0; c->stage_gen++; c->wire_pending = false;
  if (n == 0) { c->staged_kind = kind; return AVRF_OK; }
  HIP_TRY(c->h_io.ensure((n + 1) * 8));
  uint32_t *io_off = c->h_io.as<uint32_t>(), *ad_off = io_off + (n + 1);
  uint64_t a = 0, b = 0;
  for (size_t j = 0; j < n; j++) { io_off[j] = (uint32_t)a; ad_off[j] = (uint32_t)b; a += io_counts[j]; b += ad_lens[j]; }
  if (a > 0x1fffffffULL || b > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  if ((a && !ios_xy) || (b && !ads)) return AVRF_ERR_BAD_ARG;
  io_off[n] = (uint32_t)a; ad_off[n] = (uint32_t)b;
  c->tot_io = (size_t)a;
  HIP_TRY(c->d_io_off.ensure((n + 1) * 4)); HIP_TRY(c->d_ad_off.ensure((n + 1) * 4));
  HIP_TRY(c->d_ios.ensure(a * 128 + 16)); HIP_TRY(c->d_ads.ensure(b + 16));
  HIP_TRY(hipMemcpyAsync(c->d_io_off.p, io_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_ad_off.p, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (a) HIP_TRY(hipMemcpyAsync(c->d_ios.p, ios_xy, a * 128, hipMemcpyHostToDevice, c->stream));
  if (b) HIP_TRY(hipMemcpyAsync(c->d_ads.p, ads, b, hipMemcpyHostToDevice, c->stream));
  const size_t psz = kind == 1 ? 96 : kind == 3 ? 48 : 256;
  if (pks_xy) { HIP_TRY(c->d_pks.ensure(n * 64)); HIP_TRY(hipMemcpyAsync(c->d_pks.p, pks_xy, n * 64, hipMemcpyHostToDevice, c->stream)); }
  if (sks) { HIP_TRY(c->d_sks.ensure(n * 32)); HIP_TRY(hipMemcpyAsync(c->d_sks.p, sks, n * 32, hipMemcpyHostToDevice, c->stream)); }
  if (proofs) {
    HIP_TRY(c->d_proofs.ensure(n * psz)); HIP_TRY(hipMemcpyAsync(c->d_proofs.p, proofs, n * psz, hipMemcpyHostToDevice, c->stream));
    if (kind == 3) { HIP_TRY(hipStreamSynchronize(c->stream)); c->staged_kind = kind; return AVRF_OK; }   // Tiny: no batch verifier
    if (suite_host_weights(c->suite)) {       // a sponge transcript absorbs the responses on the host; counter-mode ones hash the device's records
      const size_t rsz = kind == 1 ? 32 : 64, roff = kind == 1 ? 64 : 192;
      c->h_resp.resize(n * rsz);
      for (size_t j = 0; j < n; j++) memcpy(&c->h_resp[rsz * j], proofs + psz * j + roff, rsz);
    }
    c->n_terms = kind == 1 ? 2 * n + 2 * c->tot_io + 1 : 5 * n + 2;
    HIP_TRY(c->d_c.ensure(n * 16)); HIP_TRY(c->h_c.ensure(n * 16));
    HIP_TRY(c->d_z.ensure(kind == 1 ? c->tot_io * 16 + 16 : n * 128));
    if (c->lane_owner) { HIP_TRY(c->L->d_scalars.ensure(c->n_terms * 32)); HIP_TRY(c->L->d_pre.ensure(c->n_terms * sizeof(te_pre_raw))); HIP_TRY(c->L->d_gpart.ensure(((n + 127) / 128) * 64 + 64)); }
  }
  if (wait) HIP_TRY(hipStreamSynchronize(c->stream));
  c->staged_kind = kind;
  return AVRF_OK;
}
// The wire flavour of the batch verifiers' staging (src/thin.rs:78-94, src/pedersen.rs:106-134 deserialise, then
// `push`): the caller's `serialize_compressed` bytes go to the device as they are, every point is decompressed (validate: + not the
// identity, + prime-order subgroup, src/lib.rs:410-433) STRAIGHT INTO the context's staged x || y buffers, the proofs' scalars are
// copied beside their points -- no decompressed byte crosses PCIe or a host core.  A point that fails makes the batch InvalidData
// here, before any equation (as the reference's deserialisation would).  kind 1 thin (pk, R || s), 2 pedersen (Yb, R, Ok || s || sb).
// wait = false (pool.hip): nothing is waited for -- the flag is read by batch_collect once the stream's work has completed
int ctx_stage_wire(avrf_ctx *c, int kind, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                   const uint32_t *ad_lens, const uint8_t *proofs, int validate, bool wait) {
  if (!c || c->run_phase || (kind != 1 && kind != 2)) return AVRF_ERR_BAD_ARG;
  if (n && (!io_counts || !ad_lens || !proofs || (kind == 1 && !pks))) return AVRF_ERR_BAD_ARG;
  if (n > 0x0fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0; c->n = n; c->tot_io = 0; c->n_terms = 0; c->stage_gen++; c->wire_pending = false;
  if (n == 0) { c->staged_kind = kind; return AVRF_OK; }
  HIP_TRY(c->h_io.ensure((n + 1) * 8));
  uint32_t *io_off = c->h_io.as<uint32_t>(), *ad_off = io_off + (n + 1);
  uint64_t a = 0, b = 0;
  for (size_t j = 0; j < n; j++) { io_off[j] = (uint32_t)a; ad_off[j] = (uint32_t)b; a += io_counts[j]; b += ad_lens[j]; }
  if (a > 0x1fffffffULL || b > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  if ((a && !ios) || (b && !ads)) return AVRF_ERR_BAD_ARG;
  io_off[n] = (uint32_t)a; ad_off[n] = (uint32_t)b;
  c->tot_io = (size_t)a;
  const size_t L = (size_t)point_len_of(c->suite), ppts = kind == 1 ? 1 : 3, tail = kind == 1 ? 32 : 64, plen = ppts * L + tail, psz = kind == 1 ? 96 : 256;
  const size_t w_pks = kind == 1 ? n * L : 0, w_ios = 2 * (size_t)a * L, w_pr = n * plen;
  HIP_TRY(c->d_io_off.ensure((n + 1) * 4)); HIP_TRY(c->d_ad_off.ensure((n + 1) * 4));
  HIP_TRY(c->d_ios.ensure(a * 128 + 16)); HIP_TRY(c->d_ads.ensure(b + 16)); HIP_TRY(c->d_proofs.ensure(n * psz));
  if (kind == 1) HIP_TRY(c->d_pks.ensure(n * 64));
  HIP_TRY(c->d_misc.ensure(w_pks + w_ios + w_pr + 64)); HIP_TRY(c->d_status.ensure(64));
  uint8_t *dw = c->d_misc.as<uint8_t>();
  HIP_TRY(hipMemcpyAsync(c->d_io_off.p, io_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_ad_off.p, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (b) HIP_TRY(hipMemcpyAsync(c->d_ads.p, ads, b, hipMemcpyHostToDevice, c->stream));
  if (w_pks) HIP_TRY(hipMemcpyAsync(dw, pks, w_pks, hipMemcpyHostToDevice, c->stream));
  if (w_ios) HIP_TRY(hipMemcpyAsync(dw + w_pks, ios, w_ios, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dw + w_pks + w_ios, proofs, w_pr, hipMemcpyHostToDevice, c->stream));
  uint32_t *d_flag = c->d_status.as<uint32_t>(), *h_flag = c->h_flags.as<uint32_t>() + 2;
  HIP_TRY(hipMemsetAsync(d_flag, 0, 4, c->stream));
  if (kind == 1) launch_decompress_strided(c->suite, dw, (uint32_t)L, (uint32_t)n, c->d_pks.as<uint8_t>(), 64, validate, d_flag, c->stream);
  launch_decompress_strided(c->suite, dw + w_pks, (uint32_t)L, (uint32_t)(2 * a), c->d_ios.as<uint8_t>(), 64, validate, d_flag, c->stream);
  const uint8_t *dpr = dw + w_pks + w_ios;
  for (size_t p = 0; p < ppts; p++)
    launch_decompress_strided(c->suite, dpr + L * p, (uint32_t)plen, (uint32_t)n, c->d_proofs.as<uint8_t>() + 64 * p, (uint32_t)psz, validate, d_flag, c->stream);
  HIP_TRY(hipMemcpy2DAsync(c->d_proofs.as<uint8_t>() + 64 * ppts, psz, dpr + L * ppts, plen, tail, n, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(h_flag, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  if (suite_host_weights(c->suite)) {
    const size_t rsz = kind == 1 ? 32 : 64, roff = ppts * L;
    c->h_resp.resize(n * rsz);
    for (size_t j = 0; j < n; j++) memcpy(&c->h_resp[rsz * j], proofs + plen * j + roff, rsz);
  }
  c->n_terms = kind == 1 ? 2 * n + 2 * c->tot_io + 1 : 5 * n + 2;
  HIP_TRY(c->d_c.ensure(n * 16)); HIP_TRY(c->h_c.ensure(n * 16));
  HIP_TRY(c->d_z.ensure(kind == 1 ? c->tot_io * 16 + 16 : n * 128));
  if (c->lane_owner) { HIP_TRY(c->L->d_scalars.ensure(c->n_terms * 32)); HIP_TRY(c->L->d_pre.ensure(c->n_terms * sizeof(te_pre_raw))); HIP_TRY(c->L->d_gpart.ensure(((n + 127) / 128) * 64 + 64)); }
  if (!wait) { c->wire_pending = true; c->staged_kind = kind; return AVRF_OK; }
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  if (*h_flag) return AVRF_INVALID_DATA;
  c->staged_kind = kind;
  return AVRF_OK;
}
}  // namespace avrf
extern "C" {
int avrf_thin_batch_stage_wire(avrf_ctx *c, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                               const uint32_t *ad_lens, const uint8_t *proofs, int validate) {
  if (!c || ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  return avrf::ctx_stage_wire(c, 1, n, pks, ios, io_counts, ads, ad_lens, proofs, validate, true);
}
int avrf_pedersen_batch_stage_wire(avrf_ctx *c, size_t n, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                                   const uint32_t *ad_lens, const uint8_t *proofs, int validate) {
  if (!c || ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  return avrf::ctx_stage_wire(c, 2, n, nullptr, ios, io_counts, ads, ad_lens, proofs, validate, true);
}
static int stage(avrf_ctx *c, int kind, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
                 const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  return ctx_stage(c, kind, n, sks, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs, true);
}
// the per-item entry points wait for the stream before they return: their staging copies need no wait of their own
static int stage_nowait(avrf_ctx *c, int kind, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy,
                        const uint32_t *io_counts, const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  return ctx_stage(c, kind, n, sks, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs, false);
}

int avrf_thin_batch_stage(avrf_ctx *c, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                          const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  if (n && (!pks_xy || !proofs)) return AVRF_ERR_BAD_ARG;
  return stage(c, 1, n, nullptr, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs);
}
int avrf_pedersen_batch_stage(avrf_ctx *c, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                              const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  if (n && !proofs) return AVRF_ERR_BAD_ARG;
  return stage(c, 2, n, nullptr, nullptr, ios_xy, io_counts, ads, ad_lens, proofs);
}

// ---- the weight transcripts of the contexts in flight, hashed together (host_sha512_mb.h)
static void weight_digest_scalar(WeightJob &j) {
  HostSha512 h;
  if (j.msg) { h.update(j.msg, j.msg_len); h.final(j.digest); return; }
  h.update(j.prefix, j.prefix_len);
  uint8_t rec[96]; memset(rec, 0, sizeof rec);
  for (size_t k = 0; k < j.n; k++) { memcpy(rec, j.c16 + 16 * k, 16); memcpy(rec + 32, j.resp + j.rsz * k, j.rsz); h.update(rec, 32 + j.rsz); }
  h.final(j.digest);
}
namespace {
// OPT-IN (AVRF_HASH_THREADS = number of workers; default 0 = every context hashes its own transcript on its own thread).
// Contexts hand their transcript to a small pool; a worker takes up to eight pending ones and advances them in the eight lanes
// of a 512-bit register (a lone request goes through the scalar code).  Measured on the bench host (16-CPU quota): one 8-lane
// pass takes ~10 ms against 4.4 ms for one scalar chain, i.e. 3.6x the hashes per core-second but twice the latency per batch;
// the contexts form a closed loop, so with 16 of them the longer wait costs more than the saved cycles return (55-68 M/s with
// 2-4 workers against 76-81 M/s), and 48 contexts with 8 workers only draw level (74 M/s).  Kept for hosts with fewer cores per GPU.
class WeightHashService {
 public:
  static WeightHashService &get() { static WeightHashService s; return s; }
  bool enabled() const { return !workers_.empty(); }
  void run(WeightJob &job) {
    Item it{&job, false};
    std::unique_lock<std::mutex> lk(m_);
    q_.push_back(&it);
    cv_work_.notify_one();
    cv_done_.wait(lk, [&] { return it.done; });
  }
 private:
  struct Item { WeightJob *job; bool done; };
  WeightHashService() {
    int n = 0;
    if (const char *e = getenv("AVRF_HASH_THREADS")) n = atoi(e);
    if (n < 0) n = 0; if (n > 16) n = 16;
    if (!sha512_mb_available()) n = 0;
    for (int i = 0; i < n; i++) workers_.emplace_back([this] { loop(); });
  }
  ~WeightHashService() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_work_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void loop() {
    for (;;) {
      Item *take[8]; int cnt = 0;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (stop_) return;
        if (q_.size() == 1) cv_work_.wait_for(lk, std::chrono::microseconds(40), [&] { return stop_ || q_.size() >= 2; });   // companions on their way?
        if (stop_) return;
        while (cnt < 8 && !q_.empty()) { take[cnt++] = q_.front(); q_.pop_front(); }
      }
      if (cnt == 1) weight_digest_scalar(*take[0]->job);
      else if (cnt > 1) { WeightJob *jobs[8]; for (int i = 0; i < cnt; i++) jobs[i] = take[i]->job; sha512_weights_x8(jobs, cnt); }
      { std::lock_guard<std::mutex> lk(m_); for (int i = 0; i < cnt; i++) take[i]->done = true; }
      cv_done_.notify_all();
    }
  }
  std::mutex m_; std::condition_variable cv_work_, cv_done_; std::deque<Item *> q_; std::vector<std::thread> workers_; bool stop_ = false;
};
}  // namespace

// shared tail of both batch verifiers in three phases (run_phase): 1 = validation + prepare kernel + copies back enqueued,
// 2 = weight transcript hashed on the host and terms + MSM enqueued, 0 = idle.  batch_run walks all three; the
// avrf_batch_run_begin / _hash / _end entry points let one host thread keep several contexts in flight (hash one context's
// transcript while the others' kernels run) instead of parking a thread per context; the pool's workers (pool.hip) call the
// pieces themselves (batch_collect / batch_seed / batch_launch) and hash several contexts' transcripts together.
}  // extern "C"
namespace avrf {
static int host_stream_kind(int suite) { return with_suite(suite, [&](auto tag_) { using S = typename decltype(tag_)::type; return S::XOF_SHAKE ? 1 : S::TR_SHA256 ? 2 : 0; }); }
static size_t batch_prefix(int suite, uint8_t prefix[64]) {
  size_t pl = 0;
  with_suite(suite, [&](auto tag_) { using S = typename decltype(tag_)::type; memcpy(prefix, S::SUITE_ID, S::SUITE_ID_LEN); pl = S::SUITE_ID_LEN; });
  prefix[pl++] = DS_BATCH_VERIFY;
  return pl;
}
bool batch_host_weights(const avrf_ctx *c) { return host_stream_kind(c->suite) != 0; }

int batch_begin(avrf_ctx *c, int kind) {
  if (!c || c->staged_kind != kind || c->run_phase != 0) return AVRF_ERR_BAD_ARG;
  if (c->n == 0) { c->run_phase = 1; return AVRF_OK; }                 // src/thin.rs:262-264, src/pedersen.rs:343-345
  if (!c->n_terms) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->run_t0 = now_us();
  const size_t n = c->n;
  BatchDev b = batch_of(c);
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  validate_staged(c, kind, nullptr);
  const int host_stream = host_stream_kind(c->suite);
  // the weight transcript's message, prefix || records: the prepare kernel writes the records, one copy brings them back
  const size_t recsz = kind == 1 ? 64 : 96;
  uint8_t prefix[64]; const size_t pl = batch_prefix(c->suite, prefix);
  if (!host_stream) {
    HIP_TRY(c->d_rec.ensure(n * recsz)); HIP_TRY(c->h_msg.ensure(pl + n * recsz));
    b.records = c->d_rec.as<uint8_t>();
  }
  if (kind == 1) launch_thin_prepare(c->suite, b, c->d_c.as<uint32_t>(), c->d_z.as<uint32_t>(), c->d_flags.as<uint32_t>(), c->stream);
  else launch_ped_prepare(c->suite, b, c->d_c.as<uint32_t>(), c->d_z.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream);
  if (!host_stream) {
    memcpy(c->h_msg.p, prefix, pl);
    HIP_TRY(hipMemcpyAsync(c->h_msg.as<uint8_t>() + pl, c->d_rec.p, n * recsz, hipMemcpyDeviceToHost, c->stream));
    c->h_msg_len = pl + n * recsz;
  } else HIP_TRY(hipMemcpyAsync(c->h_c.p, c->d_c.p, n * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  c->run_begin_us = now_us() - c->run_t0;
  c->run_phase = 1;
  return AVRF_OK;
}

// the prepare kernel and the copies back have completed (the caller waited for c->stream or for an event behind them)
int batch_collect(avrf_ctx *c, int kind) {
  if (!c || c->staged_kind != kind || c->run_phase != 1) return AVRF_ERR_BAD_ARG;
  if (c->n && *c->h_flags.as<uint32_t>()) { c->run_phase = 0; return AVRF_INVALID_DATA; }   // src/thin.rs:266-271, src/pedersen.rs:348-353
  if (c->wire_pending) {                                               // staged from wire bytes without waiting (ctx_stage_wire): a point that failed to decode / validate
    c->wire_pending = false;
    if (c->n && c->h_flags.as<uint32_t>()[2]) { c->run_phase = 0; c->staged_kind = 0; return AVRF_INVALID_DATA; }
  }
  return AVRF_OK;
}

// weight transcript (src/thin.rs:274-279, src/pedersen.rs:361-367):
//   new(SUITE_ID); absorb [0x50]; per item absorb LE32(c) || LE32(s) [|| LE32(sb)]
// counter-mode transcripts: SHA-512 of prefix || records -> digest.  Sponge / SHA-256 transcripts (Shake128Transcript: the
// weights are the sponge's OUTPUT STREAM, 16 bytes per item, 32 for Pedersen -- sequential, so the host squeezes it and the
// terms kernel reads it from HBM instead of deriving block j / 4 from a seed; HashTranscript<Sha256> of the test suite takes
// the same route): the stream goes to c->h_weights and the digest is not used.
int batch_seed(avrf_ctx *c, int kind, uint8_t digest[64]) {
  const size_t n = c->n;
  memset(digest, 0, 64);
  if (!n) return AVRF_OK;
  const int host_stream = host_stream_kind(c->suite);
  if (host_stream) {
    const uint8_t tag = DS_BATCH_VERIFY;
    const uint8_t *cs = c->h_c.as<uint8_t>();
    const size_t rsz = kind == 1 ? 32 : 64, wsz = kind == 1 ? 16 : 32;
    uint8_t rec[96]; memset(rec, 0, sizeof rec);
    c->h_weights.resize(n * wsz);
    auto run = [&](auto &h) {
      with_suite(c->suite, [&](auto tag_) { using S = typename decltype(tag_)::type; h.update(S::SUITE_ID, S::SUITE_ID_LEN); });
      h.update(&tag, 1);
      for (size_t j = 0; j < n; j++) { memcpy(rec, cs + 16 * j, 16); memcpy(rec + 32, &c->h_resp[rsz * j], rsz); h.update(rec, 32 + rsz); }
      h.squeeze_copy(c->h_weights.data(), n * wsz);
    };
    if (host_stream == 1) { HostShake128 h; run(h); } else { HostSha256 h; run(h); }
    return AVRF_OK;
  }
  WeightJob job; job.prefix = nullptr; job.prefix_len = 0; job.c16 = nullptr; job.resp = nullptr; job.n = n; job.rsz = 0;
  job.msg = c->h_msg.as<uint8_t>(); job.msg_len = c->h_msg_len;
  WeightHashService &svc = WeightHashService::get();
  if (svc.enabled()) svc.run(job); else weight_digest_scalar(job);
  memcpy(digest, job.digest, 64);
  return AVRF_OK;
}

int batch_launch(avrf_ctx *c, int kind, const uint8_t digest[64]) {
  if (!c || c->staged_kind != kind || c->run_phase != 1) return AVRF_ERR_BAD_ARG;
  if (c->n == 0) { c->run_phase = 2; return AVRF_OK; }
  c->run_phase = 0;                                                    // an error below leaves the context idle
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = c->n;
  BatchDev b = batch_of(c);
  if (c->unit_weights) {                                               // (kind 1 only) w_j = 1: the sum IS the item's own equation
    c->h_weights.assign(n * 16, 0);
    for (size_t j = 0; j < n; j++) c->h_weights[16 * j] = 1;
    HIP_TRY(c->d_weights.ensure(n * 16));
    HIP_TRY(hipMemcpyAsync(c->d_weights.p, c->h_weights.data(), n * 16, hipMemcpyHostToDevice, c->stream));
    b.weights = c->d_weights.as<uint8_t>();
  } else if (!host_stream_kind(c->suite)) b.records = c->d_rec.as<uint8_t>();
  else {
    const size_t wsz = kind == 1 ? 16 : 32;
    HIP_TRY(c->d_weights.ensure(n * wsz));
    HIP_TRY(hipMemcpyAsync(c->d_weights.p, c->h_weights.data(), n * wsz, hipMemcpyHostToDevice, c->stream));
    b.weights = c->d_weights.as<uint8_t>();
  }
  Seed64 seed;
  for (int i = 0; i < 8; i++) { uint64_t v; memcpy(&v, digest + 8 * i, 8); seed.w[i] = __builtin_bswap64(v); }
  HIP_TRY(c->L->d_scalars.ensure(c->n_terms * 32)); HIP_TRY(c->L->d_pre.ensure(c->n_terms * sizeof(te_pre_raw))); HIP_TRY(c->L->d_gpart.ensure(((n + 127) / 128) * 64 + 64));
  const double t2 = now_us();
  if (kind == 1) launch_thin_terms(c->suite, b, seed, 0, c->d_c.as<uint32_t>(), c->d_z.as<uint32_t>(), c->L->d_scalars.as<uint32_t>(),
                                   c->L->d_pre.as<te_pre_raw>(), c->L->d_gpart.as<uint32_t>(), (uint32_t)c->n_terms, c->stream);
  else launch_ped_terms(c->suite, b, seed, 0, c->d_c.as<uint32_t>(), c->d_z.as<uint8_t>(), c->L->d_scalars.as<uint32_t>(),
                        c->L->d_pre.as<te_pre_raw>(), c->L->d_gpart.as<uint32_t>(), (uint32_t)c->n_terms, c->stream);
  const double t3 = now_us();
  if (int e = guarded([&] { return msm_te_enqueue(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), c->n_terms, c->L->ws, c->stream, c->pend) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  c->timing[3] = t3 - t2;
  c->run_msm_us = now_us() - t3;
  c->run_phase = 2;
  return AVRF_OK;
}

int batch_end(avrf_ctx *c, int kind) {
  if (!c || c->staged_kind != kind || c->run_phase != 2) return AVRF_ERR_BAD_ARG;
  c->run_phase = 0;
  if (c->n == 0) return AVRF_OK;
  HIP_TRY(hipSetDevice(c->device));
  const double t3 = now_us();
  HostExt r;
  if (int e = guarded([&] { return msm_te_finish(c->suite, c->L->ws, c->stream, &r, c->pend) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  double t4 = now_us();
  int st = point_is_identity(c, r) ? AVRF_OK : AVRF_VERIFICATION_FAILURE;   // src/thin.rs:319-322, src/pedersen.rs:420-423
  double t5 = now_us();
  c->timing[4] = c->run_msm_us + (t4 - t3); c->timing[5] = t5 - t4;
  c->timing[0] = c->timing[1] + c->timing[2] + c->timing[3] + c->timing[4] + c->timing[5];   // time spent IN the three calls
  return st;
}
}  // namespace avrf
extern "C" {

// the second of the three calls: wait for the prepare kernel, hash on the calling thread, enqueue terms + MSM
static int batch_hash(avrf_ctx *c, int kind) {
  if (!c || c->staged_kind != kind || c->run_phase != 1) return AVRF_ERR_BAD_ARG;
  if (c->n == 0) { c->run_phase = 2; return AVRF_OK; }
  HIP_TRY(hipSetDevice(c->device));
  const double tw = now_us();
  if (hipStreamSynchronize(c->stream) != hipSuccess) { c->run_phase = 0; (void)hipGetLastError(); return AVRF_ERR_NO_DEVICE; }
  const double t1 = now_us();
  if (int e = batch_collect(c, kind)) return e;
  uint8_t digest[64];
  if (int e = batch_seed(c, kind, digest)) { c->run_phase = 0; return e; }
  const double t2 = now_us();
  c->timing[1] = c->run_begin_us + (t1 - tw); c->timing[2] = t2 - t1;
  return batch_launch(c, kind, digest);
}

static int batch_run(avrf_ctx *c, int kind) {
  if (int e = batch_begin(c, kind)) return e;
  if (int e = batch_hash(c, kind)) return e;
  return batch_end(c, kind);
}

int avrf_thin_batch_run(avrf_ctx *c) { return batch_run(c, 1); }
int avrf_pedersen_batch_run(avrf_ctx *c) { return batch_run(c, 2); }
// (defensive: a run can only be open on a staged batch -- every entry point that would un-stage it is refused while
// run_phase != 0 -- but should the two ever disagree the run is closed rather than leaving the context refusing every call)
static int run_call(avrf_ctx *c, int (*phase)(avrf_ctx *, int)) {
  if (!c) return AVRF_ERR_BAD_ARG;
  if (!c->staged_kind) {
    if (c->run_phase) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); c->run_phase = 0; c->L->ws.pending_armed = false; }
    return AVRF_ERR_BAD_ARG;
  }
  return phase(c, c->staged_kind);
}
int avrf_batch_run_begin(avrf_ctx *c) { return run_call(c, batch_begin); }
int avrf_batch_run_hash(avrf_ctx *c) { return run_call(c, batch_hash); }
int avrf_batch_run_end(avrf_ctx *c) { return run_call(c, batch_end); }

int avrf_thin_batch_verify(avrf_ctx *c, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                           const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  int st = avrf_thin_batch_stage(c, n, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs);
  return st ? st : avrf_thin_batch_run(c);
}
int avrf_pedersen_batch_verify(avrf_ctx *c, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                               const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs) {
  int st = avrf_pedersen_batch_stage(c, n, ios_xy, io_counts, ads, ad_lens, proofs);
  return st ? st : avrf_pedersen_batch_run(c);
}

// ---------------------------------------------------------------- one batch split over several GPUs

// weight transcript of src/thin.rs:274-279 / src/pedersen.rs:361-367 over ALL items of a batch (host only)
int avrf_batch_weight_seed(int suite, int pedersen, size_t n, const uint8_t *c16, const uint8_t *resp, uint8_t seed_out[64]) {
  if (suite < 0 || suite >= AVRF_N_SUITES || !seed_out || (n && (!c16 || !resp))) return AVRF_ERR_BAD_ARG;
  // a sponge transcript has no seed to hand to the shards (its weight stream is sequential): the split-one-batch mode is for the
  // counter-mode (HashTranscript) suites; whole batches shard over GPUs for every suite
  if (with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)S::HOST_WEIGHTS; })) return AVRF_ERR_BAD_ARG;
  HostSha512 h;
  with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; h.update(S::SUITE_ID, S::SUITE_ID_LEN); });
  const uint8_t tag = DS_BATCH_VERIFY; h.update(&tag, 1);
  const size_t rsz = pedersen ? 64 : 32;
  uint8_t rec[96]; memset(rec, 0, sizeof rec);
  for (size_t j = 0; j < n; j++) { memcpy(rec, c16 + 16 * j, 16); memcpy(rec + 32, resp + rsz * j, rsz); h.update(rec, 32 + rsz); }
  h.final(seed_out);
  return AVRF_OK;
}

// the same for up to eight batches at once through the multi-buffer hash (host_sha512_mb.h); AVRF_ERR_NO_DEVICE when the host
// CPU has no AVX-512 (the library then hashes every transcript on its context's thread)
int avrf_batch_weight_seeds_x8(int suite, int pedersen, int count, const size_t *n, const uint8_t *const *c16, const uint8_t *const *resp, uint8_t *seeds_out) {
  if (suite < 0 || suite >= AVRF_N_SUITES || count < 1 || count > 8 || !n || !c16 || !resp || !seeds_out) return AVRF_ERR_BAD_ARG;
  if (with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)S::HOST_WEIGHTS; })) return AVRF_ERR_BAD_ARG;
  if (!sha512_mb_available()) return AVRF_ERR_NO_DEVICE;
  uint8_t prefix[64]; size_t pl = 0;
  with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; memcpy(prefix, S::SUITE_ID, S::SUITE_ID_LEN); pl = S::SUITE_ID_LEN; });
  prefix[pl++] = DS_BATCH_VERIFY;
  WeightJob jobs[8]; WeightJob *pj[8];
  for (int i = 0; i < count; i++) {
    if (n[i] && (!c16[i] || !resp[i])) return AVRF_ERR_BAD_ARG;
    jobs[i].prefix = prefix; jobs[i].prefix_len = pl; jobs[i].c16 = c16[i]; jobs[i].resp = resp[i]; jobs[i].n = n[i]; jobs[i].rsz = pedersen ? 64 : 32; pj[i] = &jobs[i];
  }
  sha512_weights_x8(pj, count);
  for (int i = 0; i < count; i++) memcpy(seeds_out + 64 * i, jobs[i].digest, 64);
  return AVRF_OK;
}

// SHA-512 of up to eight contiguous messages through the same multi-buffer code -- the form avrf_*_batch_run hands to the
// hash service (prefix || records, one buffer per batch); exported so that a test can hold it against an independent SHA-512
int avrf_sha512_x8(int count, const uint8_t *const *msgs, const size_t *lens, uint8_t *digests_out) {
  if (count < 1 || count > 8 || !msgs || !lens || !digests_out) return AVRF_ERR_BAD_ARG;
  if (!sha512_mb_available()) return AVRF_ERR_NO_DEVICE;
  static const uint8_t empty = 0;
  WeightJob jobs[8]; WeightJob *pj[8];
  for (int i = 0; i < count; i++) {
    if (lens[i] && !msgs[i]) return AVRF_ERR_BAD_ARG;
    jobs[i].msg = lens[i] ? msgs[i] : &empty; jobs[i].msg_len = lens[i]; pj[i] = &jobs[i];
  }
  sha512_weights_x8(pj, count);
  for (int i = 0; i < count; i++) memcpy(digests_out + 64 * i, jobs[i].digest, 64);
  return AVRF_OK;
}

// the same for up to sixteen messages through the form the pool uses (host_sha512_mb.h sha512_weights_x16: more than eight lanes
// run as two groups of eight in one interleaved round loop)
int avrf_sha512_x16(int count, const uint8_t *const *msgs, const size_t *lens, uint8_t *digests_out) {
  if (count < 1 || count > 16 || !msgs || !lens || !digests_out) return AVRF_ERR_BAD_ARG;
  if (!sha512_mb16_available()) return AVRF_ERR_NO_DEVICE;
  static const uint8_t empty = 0;
  WeightJob jobs[16]; WeightJob *pj[16];
  for (int i = 0; i < count; i++) {
    if (lens[i] && !msgs[i]) return AVRF_ERR_BAD_ARG;
    jobs[i].msg = lens[i] ? msgs[i] : &empty; jobs[i].msg_len = lens[i]; pj[i] = &jobs[i];
  }
  sha512_weights_x16(pj, count);
  for (int i = 0; i < count; i++) memcpy(digests_out + 64 * i, jobs[i].digest, 64);
  return AVRF_OK;
}

// prepare (src/thin.rs:209-226) on the staged shard: per-item challenges, 16 bytes each
int avrf_thin_batch_challenges(avrf_ctx *c, uint8_t *c_out) {
  if (!c || c->staged_kind != 1 || (c->n && !c_out)) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (c->n == 0) return AVRF_OK;
  HIP_TRY(hipSetDevice(c->device));
  BatchDev b = batch_of(c);
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  validate_staged(c, 1, nullptr);
  launch_thin_prepare(c->suite, b, c->d_c.as<uint32_t>(), c->d_z.as<uint32_t>(), c->d_flags.as<uint32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(c_out, c->d_c.p, c->n * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (*c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
  c->chal_gen = c->stage_gen;                                         // the challenges in d_c / d_z belong to THIS staging
  return AVRF_OK;
}

// MSM of the staged shard's terms under the GLOBAL weight stream `seed`; the shard's first item has
// global index first_index.  Includes the shard's share (G, -sum w s z0) of the shared-generator term,
// so the partial points of all shards add up to the batch MSM of src/thin.rs:319.
int avrf_thin_batch_partial(avrf_ctx *c, const uint8_t seed64[64], uint64_t first_index, uint8_t out_xy[64]) {
  if (!c || c->staged_kind != 1 || !seed64 || !out_xy) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  // a sponge / SHA-256 transcript has no seed to hand to the shards (avrf_batch_weight_seed refuses those suites too)
  if (with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)S::HOST_WEIGHTS; })) return AVRF_ERR_BAD_ARG;
  if (c->n && c->chal_gen != c->stage_gen) return AVRF_ERR_BAD_ARG;     // avrf_thin_batch_challenges has not run on this staging (or it failed)
  HIP_TRY(hipSetDevice(c->device));
  HostExt r;
  if (c->n == 0) { r = with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; return HostTe<S>::identity(); }); return finish_point(c, r, out_xy); }
  Seed64 seed;
  for (int i = 0; i < 8; i++) { uint64_t v; memcpy(&v, seed64 + 8 * i, 8); seed.w[i] = __builtin_bswap64(v); }
  BatchDev b = batch_of(c);
  launch_thin_terms(c->suite, b, seed, first_index, c->d_c.as<uint32_t>(), c->d_z.as<uint32_t>(), c->L->d_scalars.as<uint32_t>(),
                    c->L->d_pre.as<te_pre_raw>(), c->L->d_gpart.as<uint32_t>(), (uint32_t)c->n_terms, c->stream);
  if (int e = guarded([&] { return msm_te_device(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), c->n_terms, c->L->ws, c->stream, &r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  return finish_point(c, r, out_xy);
}

// The same two steps for pedersen::BatchVerifier (src/pedersen.rs:341-426): challenges of the staged shard, then the MSM
// of its 5 n_shard + 2 terms under the global weight stream (the shard's shares of the G and BLINDING_BASE terms included).
int avrf_pedersen_batch_challenges(avrf_ctx *c, uint8_t *c_out) {
  if (!c || c->staged_kind != 2 || (c->n && !c_out)) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (c->n == 0) return AVRF_OK;
  HIP_TRY(hipSetDevice(c->device));
  BatchDev b = batch_of(c);
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  validate_staged(c, 2, nullptr);
  launch_ped_prepare(c->suite, b, c->d_c.as<uint32_t>(), c->d_z.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(c_out, c->d_c.p, c->n * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (*c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
  c->chal_gen = c->stage_gen;                                         // the challenges in d_c / d_z belong to THIS staging
  return AVRF_OK;
}
int avrf_pedersen_batch_partial(avrf_ctx *c, const uint8_t seed64[64], uint64_t first_index, uint8_t out_xy[64]) {
  if (!c || c->staged_kind != 2 || !seed64 || !out_xy) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  // a sponge / SHA-256 transcript has no seed to hand to the shards (avrf_batch_weight_seed refuses those suites too)
  if (with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)S::HOST_WEIGHTS; })) return AVRF_ERR_BAD_ARG;
  if (c->n && c->chal_gen != c->stage_gen) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  HostExt r;
  if (c->n == 0) { r = with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; return HostTe<S>::identity(); }); return finish_point(c, r, out_xy); }
  Seed64 seed;
  for (int i = 0; i < 8; i++) { uint64_t v; memcpy(&v, seed64 + 8 * i, 8); seed.w[i] = __builtin_bswap64(v); }
  BatchDev b = batch_of(c);
  launch_ped_terms(c->suite, b, seed, first_index, c->d_c.as<uint32_t>(), c->d_z.as<uint8_t>(), c->L->d_scalars.as<uint32_t>(),
                   c->L->d_pre.as<te_pre_raw>(), c->L->d_gpart.as<uint32_t>(), (uint32_t)c->n_terms, c->stream);
  if (int e = guarded([&] { return msm_te_device(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), c->n_terms, c->L->ws, c->stream, &r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  return finish_point(c, r, out_xy);
}

// sum of k affine points on the host (combining per-GPU partial MSM results)
int avrf_points_sum(int suite, size_t k, const uint8_t *points_xy, uint8_t out_xy[64]) {
  if (suite < 0 || suite >= AVRF_N_SUITES || !out_xy || (k && !points_xy)) return AVRF_ERR_BAD_ARG;
  bool bad = false;
  with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; using T = HostTe<S>;
    HostExt acc = T::identity(), p;
    for (size_t i = 0; i < k && !bad; i++) { if (!T::from_affine_bytes(points_xy + 64 * i, &p)) bad = true; else acc = T::add(acc, p); }
    if (!bad) T::to_affine_bytes(acc, out_xy); });
  if (bad) return AVRF_INVALID_DATA;
  return AVRF_OK;
}

size_t avrf_batch_last_terms(avrf_ctx *c, uint8_t *bases_xy, uint8_t *scalars) {
  if (!c || !c->staged_kind || !c->n_terms || ctx_busy(c)) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return 0;
  size_t k = c->n_terms;
  if (scalars) { if (hipMemcpy(scalars, c->L->d_scalars.p, k * 32, hipMemcpyDeviceToHost) != hipSuccess) return 0; }
  if (bases_xy) {
    std::vector<te_pre_raw> pre(k);
    if (hipMemcpy(pre.data(), c->L->d_pre.p, k * sizeof(te_pre_raw), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (size_t i = 0; i < k; i++) {
      H256 x, y; memcpy(x.l, pre[i].w, 32); memcpy(y.l, pre[i].w + 8, 32);
      with_suite(c->suite, [&](auto tag) { using S = typename decltype(tag)::type; x = HostField<typename S::Fq>::from_mont(x); y = HostField<typename S::Fq>::from_mont(y); });
      memcpy(bases_xy + 64 * i, x.l, 32); memcpy(bases_xy + 64 * i + 32, y.l, 32);
    }
  }
  return k;
}

void avrf_last_timing(avrf_ctx *c, double out[8]) {
  if (!c || !out) return;
  for (int i = 0; i < 8; i++) out[i] = c->timing[i];
}

int avrf_kernel_stats(avrf_ctx *c, int reset, double *accum_ms_total, uint64_t *accum_launches, int32_t plan[4]) {
  if (!c) return AVRF_ERR_BAD_ARG;
  if (accum_ms_total) *accum_ms_total = c->L->ws.accum_ms_total;
  if (accum_launches) *accum_launches = c->L->ws.accum_launches;
  if (plan) { plan[0] = c->L->ws.last_plan.c; plan[1] = c->L->ws.last_plan.nwin; plan[2] = c->L->ws.last_plan.nb; plan[3] = c->L->ws.last_plan.lpb; }
  if (reset) { c->L->ws.accum_ms_total = 0; c->L->ws.accum_launches = 0; }
  return AVRF_OK;
}

// ---------------------------------------------------------------- independent per-item calls

static int read_flags(avrf_ctx *c) {
  if (hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) return -1;
  static const bool trace = getenv("AVRF_TRACE_FLAGS") != nullptr;   // which check refused the input: 1 range, 2 identity, 4 scalar, 8 curve
  if (trace && *c->h_flags.as<uint32_t>()) fprintf(stderr, "avrf: input flags 0x%x (suite %d)\n", *c->h_flags.as<uint32_t>(), c->suite);
  return (int)*c->h_flags.as<uint32_t>();
}

// few items with one I/O pair each: the kernels that spread an item over 32 lanes (vrf_single.hip "few items")
static bool wave_shape(const avrf_ctx *c, size_t n, const uint32_t *io_counts) {
  if (!n || n > (size_t)AVRF_WAVE_ITEMS_MAX || c->tot_io != n) return false;
  static const bool off = getenv("AVRF_NO_WAVE_ITEMS") != nullptr;     // (A/B hook)
  if (off) return false;
  for (size_t j = 0; j < n; j++) if (io_counts[j] != 1) return false;
  return true;
}

// ONE thin / tiny proof through the MSM engine (vrf_single.hip k_thin_prove_begin / _end): the terms of R = k G + sum (k z_i) I_i,
// the single-launch MSM with the host's Horner, R back as canonical x || y, challenge and response.  0.71 -> ~0.3 ms
// for one proof; same bytes (any evaluation of R gives the same group element).  The nonce never leaves device memory.
static bool one_as_msm() { static const bool on = getenv("AVRF_NO_ONE_AS_MSM") == nullptr; return on; }   // (A/B hook)
static int prove_one_as_msm(avrf_ctx *c, bool have_pk, bool tiny, uint8_t *proofs_out) {
  const size_t nt = 1 + c->tot_io, plen = tiny ? 48 : 96, sb = thin_prove_state_bytes(c->suite);
  if (!have_pk) { if (int fs = ensure_fixed(c)) return fs; }
  HIP_TRY(c->L->d_scalars.ensure(nt * 32)); HIP_TRY(c->L->d_pre.ensure(nt * sizeof(te_pre_raw))); HIP_TRY(c->d_misc.ensure(sb + 64)); HIP_TRY(c->d_out.ensure(plen));
  BatchDev b = batch_of(c);
  if (!have_pk) b.pks_xy = nullptr;
  uint8_t *d_state = c->d_misc.as<uint8_t>();
  launch_thin_prove_begin(c->suite, b, c->L->d_scalars.as<uint32_t>(), c->L->d_pre.as<te_pre_raw>(), d_state, c->stream, tiny);
  HostExt r;
  if (int e = guarded([&] { return msm_te_device(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), nt, c->L->ws, c->stream, &r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  uint8_t *rxy = c->h_c.as<uint8_t>();                                  // (pinned; the challenges are not in use by a prover)
  finish_point(c, r, rxy);
  HIP_TRY(hipMemcpyAsync(d_state + sb, rxy, 64, hipMemcpyHostToDevice, c->stream));
  launch_thin_prove_end(c->suite, b, d_state, d_state + sb, c->d_out.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream, tiny);
  HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, plen, hipMemcpyDeviceToHost, c->stream));
  const int f = read_flags(c);
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}

// ONE Pedersen proof the same way (vrf_single.hip k_ped_prove_begin / _mid / _end): Yb = pk + bl B as a two-term MSM, then R = k G + kb B
// and Ok = k I_m as two scalar vectors over {G, B, I_0, ..} in one launch; every doubling chain is the host's.  1.03 -> ~0.5 ms.
static int prove_ped_one_as_msm(avrf_ctx *c, bool have_pk, uint8_t *proofs_out, uint8_t *blindings_out) {
  const size_t m = c->tot_io, nt = 2 + m, sb = ped_prove_state_bytes(c->suite), wb = (m * 32 + 63) / 64 * 64 + 64;
  HIP_TRY(c->L->d_scalars.ensure(2 * nt * 32)); HIP_TRY(c->L->d_pre.ensure(nt * sizeof(te_pre_raw)));
  HIP_TRY(c->d_misc.ensure(sb + wb + 192 + 32)); HIP_TRY(c->d_out.ensure(256)); HIP_TRY(c->h_c.ensure(192));
  BatchDev b = batch_of(c);
  if (!have_pk) b.pks_xy = nullptr;
  uint8_t *d_state = c->d_misc.as<uint8_t>(), *d_pts = d_state + sb + wb, *d_blind = d_pts + 192;
  uint32_t *d_wts = reinterpret_cast<uint32_t *>(d_state + sb);
  uint32_t *d_sc = c->L->d_scalars.as<uint32_t>(); te_pre_raw *d_pre = c->L->d_pre.as<te_pre_raw>();
  launch_ped_prove_begin(c->suite, b, d_sc, d_pre, d_state, d_wts, c->stream);
  HostExt r[2];
  if (int e = guarded([&] { return msm_te_device(c->suite, d_pre, d_sc, 2, c->L->ws, c->stream, &r[0]) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  uint8_t *pts = c->h_c.as<uint8_t>();                                  // pinned: Yb | R | Ok
  finish_point(c, r[0], pts);
  HIP_TRY(hipMemcpyAsync(d_pts, pts, 64, hipMemcpyHostToDevice, c->stream));
  launch_ped_prove_mid(c->suite, b, d_sc, d_pre, d_state, d_wts, d_pts, c->stream);
  if (int e = guarded([&] { return msm_te_small_vectors(c->suite, d_pre, d_sc, nt, 2, c->L->ws, c->stream, r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
  finish_point(c, r[0], pts + 64); finish_point(c, r[1], pts + 128);
  HIP_TRY(hipMemcpyAsync(d_pts + 64, pts + 64, 128, hipMemcpyHostToDevice, c->stream));
  launch_ped_prove_end(c->suite, b, d_state, d_pts, c->d_out.as<uint8_t>(), blindings_out ? d_blind : nullptr, c->d_flags.as<uint32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, 256, hipMemcpyDeviceToHost, c->stream));
  if (blindings_out) HIP_TRY(hipMemcpyAsync(blindings_out, d_blind, 32, hipMemcpyDeviceToHost, c->stream));
  const int f = read_flags(c);
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}

int avrf_thin_prove(avrf_ctx *c, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                    const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out) {
  if (n && (!sks || !proofs_out)) return AVRF_ERR_BAD_ARG;
  int st = stage_nowait(c, 1, n, sks, pks_xy, ios_xy, io_counts, ads, ad_lens, nullptr);
  if (st || !n) return st;
  c->staged_kind = 0;
  HIP_TRY(c->d_out.ensure(n * 96));
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  double t0 = now_us();
  if (n == 1 && c->tot_io < 1000 && one_as_msm()) {
    HIP_TRY(c->h_c.ensure(64));
    const int st1 = prove_one_as_msm(c, pks_xy != nullptr, false, proofs_out);
    c->timing[0] = now_us() - t0;
    return st1;
  }
  if (pks_xy && wave_shape(c, n, io_counts)) {
    HIP_TRY(c->d_status.ensure(n * 4)); HIP_TRY(c->h_c.ensure(n * 4));
    if (launch_thin_prove_wave(c->suite, batch_of(c), c->d_out.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->d_status.as<int32_t>(), c->stream)) {
      HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 96, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_c.p, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
      int f = read_flags(c);
      if (f < 0) return AVRF_ERR_NO_DEVICE;
      bool fallback = false;
      for (size_t j = 0; j < n; j++) fallback |= c->h_c.as<int32_t>()[j] == AVRF_WAVE_FALLBACK;
      if (!fallback) { c->timing[0] = now_us() - t0; return f ? AVRF_INVALID_DATA : AVRF_OK; }
      HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));          // a degenerate point somewhere: the lane-per-item kernel takes the call
    }
  }
  if (int e = per_item_chunks(c, pks_xy != nullptr, [&](const BatchDev &b) { launch_thin_prove(c->suite, b, c->d_out.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream); })) return e;
  HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 96, hipMemcpyDeviceToHost, c->stream));
  int f = read_flags(c);
  c->timing[0] = now_us() - t0;
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}

int avrf_thin_verify(avrf_ctx *c, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                     const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out) {
  if (n && (!pks_xy || !proofs || !status_out)) return AVRF_ERR_BAD_ARG;
  int st = stage_nowait(c, 1, n, nullptr, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs);
  if (st || !n) return st;
  HIP_TRY(c->d_status.ensure(n * 4));
  double t0 = now_us();
  // ONE item: its equation  R + c z0 pk + sum_i c z_i O_i - s z0 G - sum_i s z_i I_i == 0  (src/thin.rs:158-161 expanded, the
  // BatchVerifier's sum with the weight w = 1) through the prepare / terms kernels and the single-launch MSM
  // (msm.hip k_msm_tiny_bits): the doubling chain runs on the host's Horner instead of a lone wave -- 0.54 -> 0.28 ms.  Same
  // statuses as the per-item kernels: the flags of the prepare kernel and of the validation are InvalidData, a non-zero sum is
  // VerificationFailure.  Everything is enqueued back to back; the one wait is in batch_end.
  if (n == 1 && one_as_msm() && c->n_terms && c->n_terms <= 2048) {
    uint8_t zero[64] = {0};
    c->unit_weights = true;
    int st = batch_begin(c, 1);
    if (st == AVRF_OK) st = batch_launch(c, 1, zero);
    if (st == AVRF_OK) st = batch_end(c, 1);
    c->unit_weights = false;
    if (st == AVRF_OK || st == AVRF_VERIFICATION_FAILURE) {
      if (*c->h_flags.as<uint32_t>()) st = AVRF_INVALID_DATA;
      status_out[0] = st; c->timing[0] = now_us() - t0;
      return AVRF_OK;
    }
    c->run_phase = 0;
    return st;
  }
  if (wave_shape(c, n, io_counts) && launch_thin_verify_wave(c->suite, batch_of(c), c->d_status.as<int32_t>(), c->stream)) {
    validate_staged(c, 1, c->d_status.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
    bool fallback = false;
    for (size_t j = 0; j < n; j++) fallback |= status_out[j] == AVRF_WAVE_FALLBACK;
    if (!fallback) { c->timing[0] = now_us() - t0; return AVRF_OK; }     // (else a degenerate point somewhere: the lane-per-item kernel takes the call)
  }
  if (int e = per_item_chunks(c, true, [&](const BatchDev &b) { launch_thin_verify(c->suite, b, c->d_status.as<int32_t>(), c->stream); })) return e;
  validate_staged(c, 1, c->d_status.as<int32_t>());                   // Validate::Yes failures overwrite the item's status with InvalidData
  HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  c->timing[0] = now_us() - t0;
  return AVRF_OK;
}

// tiny::Prover::prove / tiny::Verifier::verify (src/tiny.rs:163-214) for batches of independent items
int avrf_tiny_prove(avrf_ctx *c, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                    const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out) {
  if (n && (!sks || !proofs_out)) return AVRF_ERR_BAD_ARG;
  int st = stage(c, 3, n, sks, pks_xy, ios_xy, io_counts, ads, ad_lens, nullptr);
  if (st || !n) return st;
  c->staged_kind = 0;
  HIP_TRY(c->d_out.ensure(n * 48));
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  if (n == 1 && c->tot_io < 1000 && one_as_msm()) { HIP_TRY(c->h_c.ensure(64)); return prove_one_as_msm(c, pks_xy != nullptr, true, proofs_out); }
  if (pks_xy && wave_shape(c, n, io_counts)) {                         // few items: 32 lanes per item (vrf_single.hip)
    HIP_TRY(c->d_status.ensure(n * 4)); HIP_TRY(c->h_c.ensure(n * 4));
    if (launch_thin_prove_wave(c->suite, batch_of(c), c->d_out.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->d_status.as<int32_t>(), c->stream, true)) {
      HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 48, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_c.p, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
      int f = read_flags(c);
      if (f < 0) return AVRF_ERR_NO_DEVICE;
      bool fallback = false;
      for (size_t j = 0; j < n; j++) fallback |= c->h_c.as<int32_t>()[j] == AVRF_WAVE_FALLBACK;
      if (!fallback) return f ? AVRF_INVALID_DATA : AVRF_OK;
      HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    }
  }
  if (int e = per_item_chunks(c, pks_xy != nullptr, [&](const BatchDev &b) { launch_thin_prove(c->suite, b, c->d_out.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream, true); })) return e;
  HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 48, hipMemcpyDeviceToHost, c->stream));
  int f = read_flags(c);
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}
int avrf_tiny_verify(avrf_ctx *c, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                     const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out) {
  if (n && (!pks_xy || !proofs || !status_out)) return AVRF_ERR_BAD_ARG;
  int st = stage(c, 3, n, nullptr, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs);
  if (st || !n) return st;
  HIP_TRY(c->d_status.ensure(n * 4));
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  if (wave_shape(c, n, io_counts) && launch_tiny_verify_wave(c->suite, batch_of(c), c->d_status.as<int32_t>(), c->stream)) {
    validate_staged(c, 3, c->d_status.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
    bool fallback = false;
    for (size_t j = 0; j < n; j++) fallback |= status_out[j] == AVRF_WAVE_FALLBACK;
    if (!fallback) { c->staged_kind = 0; return AVRF_OK; }
  }
  if (int e = per_item_chunks(c, true, [&](const BatchDev &b) { launch_tiny_verify(c->suite, b, c->d_status.as<int32_t>(), c->stream); })) return e;
  validate_staged(c, 3, c->d_status.as<int32_t>());
  HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  c->staged_kind = 0;
  return AVRF_OK;
}

int avrf_pedersen_prove(avrf_ctx *c, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
                        const uint8_t *ads, const uint32_t *ad_lens, uint8_t *proofs_out, uint8_t *blindings_out) {
  if (n && (!sks || !proofs_out)) return AVRF_ERR_BAD_ARG;
  int st = stage_nowait(c, 2, n, sks, pks_xy, ios_xy, io_counts, ads, ad_lens, nullptr);
  if (st || !n) return st;
  c->staged_kind = 0;
  HIP_TRY(c->d_out.ensure(n * 256)); HIP_TRY(c->d_misc.ensure(n * 32));
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  double t0 = now_us();
  if (n == 1 && c->tot_io <= 1000 && one_as_msm()) {
    if (!pks_xy) { if (int fs = ensure_fixed(c)) return fs; }
    const int st1 = prove_ped_one_as_msm(c, pks_xy != nullptr, proofs_out, blindings_out);
    c->timing[0] = now_us() - t0;
    return st1;
  }
  if (pks_xy && wave_shape(c, n, io_counts)) {                         // few items: 32 lanes per item (vrf_single.hip)
    if (int fs = ensure_fixed(c)) return fs;
    HIP_TRY(c->d_status.ensure(n * 4)); HIP_TRY(c->h_c.ensure(n * 4));
    if (launch_ped_prove_wave(c->suite, batch_of(c), c->d_out.as<uint8_t>(), c->d_misc.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->d_status.as<int32_t>(), c->stream)) {
      HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 256, hipMemcpyDeviceToHost, c->stream));
      if (blindings_out) HIP_TRY(hipMemcpyAsync(blindings_out, c->d_misc.p, n * 32, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_c.p, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
      int f = read_flags(c);
      if (f < 0) return AVRF_ERR_NO_DEVICE;
      bool fallback = false;
      for (size_t j = 0; j < n; j++) fallback |= c->h_c.as<int32_t>()[j] == AVRF_WAVE_FALLBACK;
      if (!fallback) { c->timing[0] = now_us() - t0; return f ? AVRF_INVALID_DATA : AVRF_OK; }
      HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    }
  }
  if (int e = per_item_chunks(c, pks_xy != nullptr, [&](const BatchDev &b) { launch_ped_prove(c->suite, b, c->d_out.as<uint8_t>(), c->d_misc.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream); })) return e;
  HIP_TRY(hipMemcpyAsync(proofs_out, c->d_out.p, n * 256, hipMemcpyDeviceToHost, c->stream));
  if (blindings_out) HIP_TRY(hipMemcpyAsync(blindings_out, c->d_misc.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  int f = read_flags(c);
  c->timing[0] = now_us() - t0;
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}

int avrf_pedersen_verify(avrf_ctx *c, size_t n, const uint8_t *ios_xy, const uint32_t *io_counts,
                         const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, int32_t *status_out) {
  if (n && (!proofs || !status_out)) return AVRF_ERR_BAD_ARG;
  int st = stage_nowait(c, 2, n, nullptr, nullptr, ios_xy, io_counts, ads, ad_lens, proofs);
  if (st || !n) return st;
  HIP_TRY(c->d_status.ensure(n * 4));
  double t0 = now_us();
  // ONE item: its two equations (src/pedersen.rs:229-245) as two scalar vectors over the item's seven bases -- the terms kernel run with
  // the weights (t, u) = (1, 0) and (0, 1) -- through the single-launch MSM with the host's two Horners side by side (see
  // avrf_thin_verify): both sums must be the identity, exactly the reference's two checks.  0.52 -> ~0.33 ms.
  if (n == 1 && one_as_msm() && c->n_terms && c->n_terms <= 2048) {
    int st = batch_begin(c, 2);                                        // validation + prepare kernel (challenge, merged pair) + flags copy
    c->run_phase = 0;
    if (st != AVRF_OK) return st;
    const size_t nt = c->n_terms;
    HIP_TRY(c->L->d_scalars.ensure(2 * nt * 32)); HIP_TRY(c->L->d_pre.ensure(nt * sizeof(te_pre_raw))); HIP_TRY(c->L->d_gpart.ensure(2 * 64 + 64));
    c->h_weights.assign(64, 0); c->h_weights[0] = 1; c->h_weights[32 + 16] = 1;                   // (t, u) = (1, 0) | (0, 1)
    HIP_TRY(c->d_weights.ensure(64));
    HIP_TRY(hipMemcpyAsync(c->d_weights.p, c->h_weights.data(), 64, hipMemcpyHostToDevice, c->stream));
    BatchDev b = batch_of(c);
    Seed64 seed; for (int i = 0; i < 8; i++) seed.w[i] = 0;
    for (int v = 0; v < 2; v++) {
      b.weights = c->d_weights.as<uint8_t>() + 32 * v;
      launch_ped_terms(c->suite, b, seed, 0, c->d_c.as<uint32_t>(), c->d_z.as<uint8_t>(), c->L->d_scalars.as<uint32_t>() + (size_t)v * nt * 8,
                       c->L->d_pre.as<te_pre_raw>(), c->L->d_gpart.as<uint32_t>(), (uint32_t)nt, c->stream);
    }
    HostExt r[2];
    if (int e = guarded([&] { return msm_te_small_vectors(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), nt, 2, c->L->ws, c->stream, r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
    status_out[0] = *c->h_flags.as<uint32_t>() ? AVRF_INVALID_DATA : (point_is_identity(c, r[0]) && point_is_identity(c, r[1])) ? AVRF_OK : AVRF_VERIFICATION_FAILURE;
    c->timing[0] = now_us() - t0;
    return AVRF_OK;
  }
  if (wave_shape(c, n, io_counts) && launch_ped_verify_wave(c->suite, batch_of(c), c->d_status.as<int32_t>(), c->stream)) {
    validate_staged(c, 2, c->d_status.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
    bool fallback = false;
    for (size_t j = 0; j < n; j++) fallback |= status_out[j] == AVRF_WAVE_FALLBACK;
    if (!fallback) { c->timing[0] = now_us() - t0; return AVRF_OK; }
  }
  if (int e = per_item_chunks(c, true, [&](const BatchDev &b) { launch_ped_verify(c->suite, b, c->d_status.as<int32_t>(), c->stream); })) return e;
  validate_staged(c, 2, c->d_status.as<int32_t>());
  HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  c->timing[0] = now_us() - t0;
  return AVRF_OK;
}

static int smul_common(avrf_ctx *c, size_t n, const uint8_t *scalars, const uint8_t *points_xy, uint8_t *out_xy) {
  if (!c || (n && (!scalars || !out_xy))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  // A handful of variable-base products (Secret::output = sk * input, src/lib.rs:391-393: 80 us on a CPU core): a lane walks 253
  // doublings in 1.9 ms however few items there are.  Up to 32 go through the single-launch MSM instead, as n scalar vectors over
  // the n bases with the scalars on the diagonal (a zero scalar has no bit sums), the doubling chains folded side by side on the host pool: 0.15 ms for one, ~0.4 ms for 32.
  // The bit sums are the literal product for ANY curve point, like the lane kernel's window form (no endomorphism split).
  if (points_xy && n <= 32 && one_as_msm()) {
    for (size_t i = 0; i < n; i++) if (!scalar_in_range(c->suite, scalars + 32 * i)) return AVRF_INVALID_DATA;
    HIP_TRY(c->d_misc.ensure(n * 64)); HIP_TRY(c->L->d_scalars.ensure(n * n * 32)); HIP_TRY(c->L->d_pre.ensure(n * sizeof(te_pre_raw)));
    std::vector<uint8_t> diag(n * n * 32, 0);
    for (size_t i = 0; i < n; i++) memcpy(&diag[(i * n + i) * 32], scalars + 32 * i, 32);
    HIP_TRY(hipMemcpyAsync(c->d_misc.p, points_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->L->d_scalars.p, diag.data(), diag.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
    launch_pre_from_affine(c->suite, c->d_misc.as<uint8_t>(), n, c->L->d_pre.as<te_pre_raw>(), c->d_flags.as<uint32_t>(), 0, c->stream);
    HIP_TRY(hipMemcpyAsync(c->h_flags.p, c->d_flags.p, 4, hipMemcpyDeviceToHost, c->stream));
    HostExt r[32];
    if (int e = guarded([&] { return msm_te_small_vectors(c->suite, c->L->d_pre.as<te_pre_raw>(), c->L->d_scalars.as<uint32_t>(), n, n, c->L->ws, c->stream, r) ? (int)AVRF_ERR_BAD_ARG : 0; })) return e;
    if (*c->h_flags.as<uint32_t>()) return AVRF_INVALID_DATA;
    for (size_t i = 0; i < n; i++) finish_point(c, r[i], out_xy + 64 * i);
    return AVRF_OK;
  }
  HIP_TRY(c->d_sks.ensure(n * 32)); HIP_TRY(c->d_out.ensure(n * 64));
  HIP_TRY(hipMemcpyAsync(c->d_sks.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  if (points_xy) { HIP_TRY(c->d_misc.ensure(n * 64)); HIP_TRY(hipMemcpyAsync(c->d_misc.p, points_xy, n * 64, hipMemcpyHostToDevice, c->stream)); }
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  if (!points_xy) { if (int fs = ensure_fixed(c)) return fs; }
  launch_smul(c->suite, c->d_sks.as<uint8_t>(), points_xy ? c->d_misc.as<uint8_t>() : nullptr, (uint32_t)n, c->d_out.as<uint8_t>(),
              c->d_flags.as<uint32_t>(), c->d_fixed.as<te_pre_raw>(), c->stream);
  HIP_TRY(hipMemcpyAsync(out_xy, c->d_out.p, n * 64, hipMemcpyDeviceToHost, c->stream));
  int f = read_flags(c);
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  return f ? AVRF_INVALID_DATA : AVRF_OK;
}
int avrf_scalar_mul_base(avrf_ctx *c, size_t n, const uint8_t *sks, uint8_t *out_xy) { return smul_common(c, n, sks, nullptr, out_xy); }
int avrf_scalar_mul(avrf_ctx *c, size_t n, const uint8_t *scalars, const uint8_t *points_xy, uint8_t *out_xy) {
  if (n && !points_xy) return AVRF_ERR_BAD_ARG;
  return smul_common(c, n, scalars, points_xy, out_xy);
}

size_t avrf_point_len(int suite) { return (suite < 0 || suite >= AVRF_N_SUITES) ? 0 : (size_t)point_len_of(suite); }

int avrf_points_decompress(avrf_ctx *c, size_t n, const uint8_t *in, uint8_t *out_xy, int validate, int32_t *status_out) {
  if (!c || (n && (!in || !out_xy || !status_out))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  const size_t pl = (size_t)point_len_of(c->suite);
  HIP_TRY(c->d_misc.ensure(n * pl)); HIP_TRY(c->d_out.ensure(n * 64)); HIP_TRY(c->d_status.ensure(n * 4));
  HIP_TRY(hipMemcpyAsync(c->d_misc.p, in, n * pl, hipMemcpyHostToDevice, c->stream));
  launch_decompress(c->suite, c->d_misc.as<uint8_t>(), (uint32_t)n, c->d_out.as<uint8_t>(), validate, c->d_status.as<int32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(out_xy, c->d_out.p, n * 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  return AVRF_OK;
}
int avrf_hash_to_curve(avrf_ctx *c, size_t n, const uint8_t *data, const uint32_t *data_lens, uint8_t *out_xy, int32_t *status_out) {
  if (!c || (n && (!data_lens || !out_xy || !status_out))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  std::vector<uint32_t> off(n + 1); uint64_t tot = 0;
  for (size_t i = 0; i < n; i++) { off[i] = (uint32_t)tot; tot += data_lens[i]; if (tot > 0xffffffffULL) return AVRF_ERR_BAD_ARG; }
  off[n] = (uint32_t)tot;
  if (tot && !data) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  HIP_TRY(c->d_ads.ensure(tot + 16)); HIP_TRY(c->d_ad_off.ensure((n + 1) * 4)); HIP_TRY(c->d_out.ensure(n * 64)); HIP_TRY(c->d_status.ensure(n * 4));
  if (tot) HIP_TRY(hipMemcpyAsync(c->d_ads.p, data, tot, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_ad_off.p, off.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  launch_hash_to_curve(c->suite, c->d_ads.as<uint8_t>(), c->d_ad_off.as<uint32_t>(), (uint32_t)n, c->d_out.as<uint8_t>(), c->d_status.as<int32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(out_xy, c->d_out.p, n * 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(status_out, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  return AVRF_OK;
}
// Output::hash::<N> of n output points (x || y in, N = hash_len <= 64 bytes each out)
int avrf_output_hash(avrf_ctx *c, size_t n, const uint8_t *points_xy, size_t hash_len, uint8_t *out) {
  if (!c || hash_len < 1 || hash_len > 64 || (n && (!points_xy || !out))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  HIP_TRY(c->d_misc.ensure(n * 64)); HIP_TRY(c->d_out.ensure(n * 64));
  HIP_TRY(hipMemcpyAsync(c->d_misc.p, points_xy, n * 64, hipMemcpyHostToDevice, c->stream));
  launch_output_hash(c->suite, c->d_misc.as<uint8_t>(), (uint32_t)n, (uint32_t)hash_len, c->d_out.as<uint8_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(out, c->d_out.p, n * hash_len, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  return AVRF_OK;
}
// Secret::from_seed for n 32-byte seeds: the secret scalars (LE32, canonical) and, when pks_xy_out is given, the public keys
int avrf_secret_from_seed(avrf_ctx *c, size_t n, const uint8_t *seeds, uint8_t *sks_out, uint8_t *pks_xy_out) {
  if (!c || (n && (!seeds || !sks_out))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  HIP_TRY(c->d_misc.ensure(n * 32)); HIP_TRY(c->d_sks.ensure(n * 32));
  HIP_TRY(hipMemcpyAsync(c->d_misc.p, seeds, n * 32, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_flags.p, 0, 4, c->stream));
  launch_secret_from_seed(c->suite, c->d_misc.as<uint8_t>(), (uint32_t)n, c->d_sks.as<uint8_t>(), c->d_flags.as<uint32_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(sks_out, c->d_sks.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  // the reference zeroizes the seed and the intermediate scalar (src/lib.rs:367-368): the seeds are scrubbed from the scratch buffer
  // behind the kernel, the scalars behind the copy back (d_misc / d_sks are general scratch that later, non-secret calls reuse and
  // copy from).  sks_out is the caller's to scrub, as `Secret` is the caller's in the reference.
  HIP_TRY(hipMemsetAsync(c->d_misc.p, 0, n * 32, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_sks.p, 0, n * 32, c->stream));
  int f = read_flags(c);
  if (f < 0) return AVRF_ERR_NO_DEVICE;
  if (f) return AVRF_INVALID_DATA;
  if (!pks_xy_out) return AVRF_OK;
  const int st = smul_common(c, n, sks_out, nullptr, pks_xy_out);          // (uploads the scalars into d_sks again for the base multiplication)
  if (hipMemsetAsync(c->d_sks.p, 0, n * 32, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return st ? st : (int)AVRF_ERR_NO_DEVICE; }
  return st;
}
int avrf_points_compress(avrf_ctx *c, size_t n, const uint8_t *in_xy, uint8_t *out) {
  if (!c || (n && (!in_xy || !out))) return AVRF_ERR_BAD_ARG;
  if (ctx_busy(c)) return AVRF_ERR_BAD_ARG;
  if (!n) return AVRF_OK;
  if (n > 0x7fffffffULL) return AVRF_ERR_BAD_ARG;
  HIP_TRY(hipSetDevice(c->device));
  c->staged_kind = 0;
  HIP_TRY(c->d_misc.ensure(n * 64)); const size_t pl = (size_t)point_len_of(c->suite);
  HIP_TRY(c->d_out.ensure(n * pl));
  HIP_TRY(hipMemcpyAsync(c->d_misc.p, in_xy, n * 64, hipMemcpyHostToDevice, c->stream));
  launch_compress(c->suite, c->d_misc.as<uint8_t>(), (uint32_t)n, c->d_out.as<uint8_t>(), c->stream);
  HIP_TRY(hipMemcpyAsync(out, c->d_out.p, n * pl, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipGetLastError());
  return AVRF_OK;
}

}  // extern "C"
