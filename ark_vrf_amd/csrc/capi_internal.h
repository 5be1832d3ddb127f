// capi_internal.h -- what the translation units of libavrf.so share about a context (capi.hip, pool.hip): buffers, the
// execution lane (stream + MSM workspace) and the phases of BatchVerifier::verify (src/thin.rs:257-325, src/pedersen.rs:341-426).
// Not part of the C ABI (include/avrf.h).
#pragma once
#include "../../include/avrf.h"
#include "host_te.h"
#include "msm.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <new>
#include <vector>

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "avrf: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return AVRF_ERR_NO_DEVICE; } } while (0)

namespace avrf {

struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return (T *)p; }
};
struct PinBuf {
  void *p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes + 256);
    if (e == hipSuccess) cap = bytes + 256;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return (T *)p; }
};

template <class F> int guarded(F f) {
  try { return f(); }
  catch (const avrf::HipFailure &e) { fprintf(stderr, "avrf: HIP error %s at %s:%d\n", hipGetErrorString(e.err), e.file, e.line); return AVRF_ERR_NO_DEVICE; }
  catch (const std::bad_alloc &) { return AVRF_ERR_NO_DEVICE; }
}

// An execution lane: the stream an MSM chain runs on, its workspace and the term arrays the chain reads.  A context owns one;
// the slots of a pool (pool.hip) borrow the pool's lanes for the MSM phase of their batches, so that many staged batches can
// wait for their weight transcript without each holding a stream and ~150 MB of workspace.  One chain in flight per lane.
struct Lane {
  hipStream_t stream = nullptr;
  int queued = 0;                 // chains enqueued and not yet collected (pool.hip)
  MsmWorkspace ws;
  DevBuf d_scalars, d_pre, d_gpart;
  void release() { ws.release(); d_scalars.release(); d_pre.release(); d_gpart.release(); if (stream) (void)hipStreamDestroy(stream); stream = nullptr; }
};

}  // namespace avrf

struct avrf_ctx {
  int suite = 0, device = 0;
  hipStream_t stream = nullptr;   // where the next piece of work is enqueued: the own lane's stream, or what a pool points it at
  avrf::Lane own; avrf::Lane *L = &own;
  avrf::MsmPending *pend = nullptr;   // pool slots: where the MSM chain's results land (several chains queue on one lane); else the lane's workspace
  bool lane_owner = true;         // false: a pool slot -- no stream or workspace of its own (L and stream are set by the pool)
  // staged batch
  int validate = 0;               // avrf_ctx_set_validation: 0 unchecked (typed-point callers), 1 on-curve, 2 + subgroup
  bool wire_pending = false;              // staged from wire bytes without waiting: batch_collect reads the decode flag (h_flags[2])
  uint64_t stage_gen = 0, chal_gen = 0;   // challenges of *_batch_challenges belong to staging generation chal_gen
  int staged_kind = 0;            // 0 none, 1 thin, 2 pedersen
  size_t n = 0, tot_io = 0, n_terms = 0;
  avrf::DevBuf d_pks, d_ios, d_io_off, d_ads, d_ad_off, d_proofs, d_sks;
  std::vector<uint8_t> h_resp;    // host copy of the response scalars (s [, sb]) for the weight transcript (sponge transcripts only)
  avrf::DevBuf d_rec; avrf::PinBuf h_msg;     // counter-mode transcripts: the records written by the prepare kernel, and prefix || records on the host
  size_t h_msg_len = 0;
  std::vector<uint8_t> h_weights; avrf::DevBuf d_weights;   // sponge transcripts: the squeezed weight stream of the staged batch
  avrf::DevBuf d_c, d_z, d_flags, d_misc, d_out, d_status;
  avrf::DevBuf d_tabs;                               // per-item window tables of the independent prove / verify kernels
  avrf::DevBuf d_fixed; bool fixed_ready = false;   // fixed-base tables of G and BLINDING_BASE (provers, scalar_mul_base), built on first use
  avrf::PinBuf h_c, h_flags, h_io;
  double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int run_phase = 0;              // batch_begin / batch_hash / batch_end
  bool unit_weights = false;      // batch_launch: every item's weight is 1 (avrf_thin_verify runs ONE item as its own equation through the MSM path)
  double run_t0 = 0, run_begin_us = 0, run_msm_us = 0;
};

namespace avrf {
// the phases of a staged batch's run, shared by the three-call ABI (capi.hip) and the pool's workers (pool.hip)
int ctx_create(int suite, int device, bool lane_owner, avrf_ctx **out);
int ctx_stage(avrf_ctx *c, int kind, size_t n, const uint8_t *sks, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts,
              const uint8_t *ads, const uint32_t *ad_lens, const uint8_t *proofs, bool wait);
int ctx_stage_wire(avrf_ctx *c, int kind, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                   const uint32_t *ad_lens, const uint8_t *proofs, int validate, bool wait);   // serialize_compressed bytes in, decompressed on the device
int batch_begin(avrf_ctx *c, int kind);                               // validation + prepare kernel + copies back enqueued on c->stream
int batch_collect(avrf_ctx *c, int kind);                             // (c->stream's work has completed) AVRF_INVALID_DATA for a refused item
bool batch_host_weights(const avrf_ctx *c);                           // sponge / SHA-256 transcript: batch_seed squeezes the weight stream itself
int batch_seed(avrf_ctx *c, int kind, uint8_t digest[64]);            // the weight transcript on the calling thread
int batch_launch(avrf_ctx *c, int kind, const uint8_t digest[64]);    // terms kernel + MSM chain enqueued on c->stream / c->L
int batch_end(avrf_ctx *c, int kind);                                 // waits for the chain, folds, verdict
double now_us();
}  // namespace avrf
