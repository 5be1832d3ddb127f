// pool.hip -- avrf_pool: many BatchVerifier::verify jobs in flight on one device, driven by a few native host threads.
//
// thin::BatchVerifier::verify (src/thin.rs:257-325) and pedersen::BatchVerifier::verify (src/pedersen.rs:341-426) are device
// work, then ONE sequential SHA-512 chain over all items on the host (src/thin.rs:274-279: 3.5 ms of a core per 65 536-item
// batch), then device work again (0.6 ms).  A GPU is therefore fed by MANY batches at once, and what limits a node with
// several GPUs is host time per batch.  The pool is the host side written for that:
//   * SLOTS hold staged batches (inputs, challenges, transcript records: ~30 MB each) and cost no stream and no MSM
//     workspace; LANES (stream + MSM workspace + term arrays, ~150 MB) are borrowed for the MSM phase only -- batches that
//     wait for their hash do not occupy the lanes the device is busy on;
//   * a WORKER thread owns a share of the slots and lanes and never blocks while anything of its own can progress: it
//     launches prepare kernels on its ingest streams, collects the transcripts that have arrived, hashes up to sixteen of
//     them TOGETHER (host_sha512_mb.h: the chains of different batches fill the lanes of the vector unit), launches their MSM
//     chains, folds finished ones (host_te.h) -- all from completion events, sleeping in the driver only when idle;
//   * inputs are read where the caller put them: from buffers of avrf_host_alloc the staging copies are DMA transfers that
//     overlap the other batches' kernels and cost no host time (include/avrf.h "Ownership").
// Verdicts are those of avrf_thin_batch_run / avrf_pedersen_batch_run: the workers call the same phases (capi_internal.h), each
// under `guarded` -- a failed HIP call or allocation inside a worker becomes the batch's status, never an exception off the thread.
#include "capi_internal.h"
#include "suite_dispatch.h"
#include "host_sha512.h"
#include "host_sha512_mb.h"
#include "host_numa.h"
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <thread>
#include <time.h>

using namespace avrf;

namespace {

enum SlotState { S_FREE = 0, S_SUBMITTED, S_BEGUN, S_READY, S_HASHED, S_MSM, S_DONE };

struct Slot {
  avrf_ctx *c = nullptr;          // a lane-less context: staged buffers, challenges, records, pinned transcript
  std::atomic<int> state{S_FREE};
  bool has_batch = false;         // the buffers hold a staged batch (it can be run again without its host sources)
  bool wire = false; int validate = 0;   // the host sources are serialize_compressed bytes (avrf_pool_submit_wire): decompressed on the device while staging
  bool from_host = false;         // stage from the host sources before the run
  bool resident_invalid = false;  // the staged bytes did not decode / validate (a wire batch): runs from the resident state answer InvalidData again
  uint64_t ticket = 0;
  int status = 0;
  hipEvent_t ev = nullptr;        // behind the prepare kernel's copies back, then behind the MSM chain
  Lane *lane = nullptr;
  MsmPending pend;                // what the slot's MSM chain sends back (several chains queue on one lane)
  uint64_t seq = 0;               // order in which the worker enqueued the slot's current phase (oldest is waited for first)
  size_t n = 0;
  const uint8_t *pks = nullptr, *ios = nullptr, *ads = nullptr, *proofs = nullptr;
  const uint32_t *io_counts = nullptr, *ad_lens = nullptr;
  uint8_t digest[64];
};

struct Worker {
  std::thread th;
  std::vector<int> slots;
  std::vector<Lane *> lanes; size_t capacity = 0;       // capacity: chains the worker's lanes hold (lanes x depth)
  hipStream_t ingest = nullptr;
  // thread CPU time by phase (us): begin (staging + prepare launches), collect, hash, launch (terms + MSM chain), end (fold), all
  double cpu_us[6] = {0, 0, 0, 0, 0, 0};
  uint64_t hash_groups = 0, hash_msgs = 0, waits = 0;
};

double thread_cpu_us() {
  timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

}  // namespace

struct avrf_pool {
  int suite = 0, device = 0, kind = 1, group = 1, depth = 1;
  std::vector<Slot> slots; std::vector<Lane> lanes; std::vector<Worker> workers;
  int numa_node = -1; bool numa_bound = false;          // where the workers run (host_numa.h)
  std::mutex m; std::condition_variable cv_work, cv_done;
  bool stop = false;
  uint64_t next_ticket = 1;
  // cycle mode (avrf_pool_cycle): the workers re-run their resident batches until the region is whole blocks and long enough
  bool cycling = false, cyc_closed = false, cyc_from_host = false;
  uint64_t cyc_issued = 0, cyc_target = 0, cyc_block = 1, cyc_done = 0, cyc_mismatch = 0, cyc_cap = 0;
  double cyc_t0 = 0, cyc_min_us = 0, cyc_t_last = 0; int cyc_expect = 0;
  bool take_locked() {                                                 // (m held) reserve the next step of the cycle
    if (!cycling || cyc_closed) return false;
    if (cyc_issued >= cyc_target) {
      if (now_us() - cyc_t0 >= cyc_min_us || cyc_target >= cyc_cap) { cyc_closed = true; return false; }
      cyc_target += cyc_block;
    }
    cyc_issued++;
    return true;
  }
};

namespace {

void scalar_digest(const uint8_t *msg, size_t len, uint8_t out[64]) { HostSha512 h; h.update(msg, len); h.final(out); }

struct Run {
  avrf_pool *P; Worker &W; uint64_t seq = 0;
  Run(avrf_pool *p, Worker &w) : P(p), W(w) {}

  void done(Slot &S, int status) {                                     // the slot's run is over (verdict or error)
    if (S.lane) { S.lane->queued--; S.lane = nullptr; }
    S.c->run_phase = 0; S.c->L = &S.c->own; S.c->stream = nullptr;
    std::lock_guard<std::mutex> lk(P->m);
    if (P->cycling) {
      P->cyc_done++; P->cyc_t_last = now_us();
      if (status != P->cyc_expect) P->cyc_mismatch++;
      S.status = status;
      if (status == P->cyc_expect && P->take_locked()) { S.from_host = P->cyc_from_host; S.state = S_SUBMITTED; }
      else S.state = S_FREE;
    } else { S.status = status; S.state = S_DONE; }
    P->cv_done.notify_all();
  }
  // (a wire batch whose points did not decode leaves no usable staged state -- batch_collect drops it -- but the slot still "holds" that
  // batch: resubmission and cycle mode from the resident state must give the same verdict, not a bad-argument error)
  static void latch(Slot &S, int status) { if (S.wire && status == AVRF_INVALID_DATA) S.resident_invalid = true; }

  void begin(Slot &S) {
    const double t0 = thread_cpu_us();
    avrf_ctx *c = S.c;
    c->stream = W.ingest;
    c->L = &c->own;
    int st = AVRF_OK;
    if (!S.from_host && S.resident_invalid) { W.cpu_us[0] += thread_cpu_us() - t0; done(S, AVRF_INVALID_DATA); return; }
    if (S.from_host) {
      S.resident_invalid = false;
      if (S.wire) st = guarded([&] { return ctx_stage_wire(c, P->kind, S.n, S.pks, S.ios, S.io_counts, S.ads, S.ad_lens, S.proofs, S.validate, /*wait=*/false); });
      else st = guarded([&] { return ctx_stage(c, P->kind, S.n, nullptr, S.pks, S.ios, S.io_counts, S.ads, S.ad_lens, S.proofs, /*wait=*/false); });
      S.has_batch = st == AVRF_OK;
    }
    if (st == AVRF_OK) st = guarded([&] { return batch_begin(c, P->kind); });
    if (st == AVRF_OK && hipEventRecord(S.ev, c->stream) != hipSuccess) { (void)hipGetLastError(); st = AVRF_ERR_NO_DEVICE; }
    W.cpu_us[0] += thread_cpu_us() - t0;
    if (st != AVRF_OK) { done(S, st); return; }
    S.seq = ++seq; S.state = S_BEGUN;
  }

  void hash(std::vector<int> &ready, size_t count) {
    const double t0 = thread_cpu_us();
    WeightJob jobs[16]; WeightJob *pj[16]; Slot *ss[16]; int k = 0;
    for (size_t i = 0; i < count; i++) {
      Slot &S = P->slots[ready[i]];
      if (batch_host_weights(S.c) || S.c->n == 0) {                    // sponge transcripts squeeze their own stream; empty batches have no transcript
        int st = guarded([&] { return batch_seed(S.c, P->kind, S.digest); });
        if (st != AVRF_OK) { done(S, st); continue; }
        S.state = S_HASHED;
        continue;
      }
      jobs[k].prefix = nullptr; jobs[k].prefix_len = 0; jobs[k].c16 = nullptr; jobs[k].resp = nullptr; jobs[k].n = 0; jobs[k].rsz = 0;
      jobs[k].msg = S.c->h_msg.as<uint8_t>(); jobs[k].msg_len = S.c->h_msg_len; pj[k] = &jobs[k]; ss[k] = &S; k++;
    }
    if (k == 1 || (k > 1 && !sha512_mb_available())) { for (int i = 0; i < k; i++) scalar_digest(jobs[i].msg, jobs[i].msg_len, jobs[i].digest); }
    else if (k > 1) { if (sha512_mb16_available()) sha512_weights_x16(pj, k); else { sha512_weights_x8(pj, k < 8 ? k : 8); if (k > 8) sha512_weights_x8(pj + 8, k - 8); } }
    for (int i = 0; i < k; i++) { memcpy(ss[i]->digest, jobs[i].digest, 64); ss[i]->state = S_HASHED; }
    ready.erase(ready.begin(), ready.begin() + count);
    if (k) { W.hash_groups++; W.hash_msgs += k; }
    W.cpu_us[2] += thread_cpu_us() - t0;
  }

  Lane *pick_lane() {                                                  // the least loaded lane with room, or nullptr
    Lane *best = nullptr;
    for (Lane *L : W.lanes) if (L->queued < P->depth && (!best || L->queued < best->queued)) best = L;
    return best;
  }

  void launch(Slot &S) {
    const double t0 = thread_cpu_us();
    S.lane = pick_lane(); S.lane->queued++;
    S.c->L = S.lane; S.c->stream = S.lane->stream;
    int st = guarded([&] { return batch_launch(S.c, P->kind, S.digest); });
    if (st == AVRF_OK && hipEventRecord(S.ev, S.lane->stream) != hipSuccess) { (void)hipGetLastError(); st = AVRF_ERR_NO_DEVICE; }
    W.cpu_us[3] += thread_cpu_us() - t0;
    if (st != AVRF_OK) { if (S.c->run_phase == 2) { (void)hipStreamSynchronize(S.lane->stream); S.lane->ws.pending_armed = false; S.pend.armed = false; } done(S, st); return; }
    S.seq = ++seq; S.state = S_MSM;
  }

  void finish(Slot &S) {
    const double t0 = thread_cpu_us();
    const int st = guarded([&] { return batch_end(S.c, P->kind); });
    W.cpu_us[4] += thread_cpu_us() - t0;
    done(S, st);
  }

  void loop() {
    (void)hipSetDevice(P->device);
    std::vector<int> ready;
    const double cpu0 = thread_cpu_us();
    for (;;) {
      bool progress = false;
      {
        std::lock_guard<std::mutex> lk(P->m);
        if (P->stop) break;
      }
      size_t n_begun = 0, n_msm = 0;
      Slot *oldest = nullptr;
      for (int si : W.slots) {                                         // finished MSM chains: fold, verdict, lane back
        Slot &S = P->slots[si];
        if (S.state != S_MSM) continue;
        if (hipEventQuery(S.ev) == hipSuccess) { finish(S); progress = true; }
        else { (void)hipGetLastError(); n_msm++; if (!oldest || S.seq < oldest->seq) oldest = &S; }
      }
      for (int si : W.slots) {                                         // hashed batches onto free lanes
        Slot &S = P->slots[si];
        if (S.state == S_HASHED && pick_lane()) { launch(S); progress = true; if (S.state == S_MSM) n_msm++; }
      }
      for (int si : W.slots) {                                         // new work: staging copies + prepare kernel
        Slot &S = P->slots[si];
        if (S.state == S_SUBMITTED) { begin(S); progress = true; }
      }
      for (int si : W.slots) {                                         // transcripts that have arrived
        Slot &S = P->slots[si];
        if (S.state != S_BEGUN) continue;
        if (hipEventQuery(S.ev) == hipSuccess) {
          const double t0 = thread_cpu_us();
          const int st = guarded([&] { return batch_collect(S.c, P->kind); });
          W.cpu_us[1] += thread_cpu_us() - t0;
          if (st != AVRF_OK) { latch(S, st); done(S, st); } else { S.state = S_READY; ready.push_back(si); }
          progress = true;
        } else { (void)hipGetLastError(); n_begun++; if (!oldest || S.seq < oldest->seq) oldest = &S; }
      }
      // hash a full group at once; a partial one only when nothing else is on its way and the lanes are running dry
      if (!ready.empty()) {
        const size_t g = (size_t)P->group;
        if (ready.size() >= g || (n_begun == 0 && 2 * n_msm <= W.capacity)) { hash(ready, ready.size() < g ? ready.size() : g); progress = true; }
      }
      if (progress) continue;
      if (oldest) { W.waits++; (void)hipEventSynchronize(oldest->ev); continue; }        // sleeps in the driver (blocking-sync events)
      std::unique_lock<std::mutex> lk(P->m);
      P->cv_work.wait(lk, [&] {
        if (P->stop) return true;
        for (int si : W.slots) if (P->slots[si].state == S_SUBMITTED) return true;
        return false;
      });
    }
    W.cpu_us[5] = thread_cpu_us() - cpu0;
  }
};

}  // namespace

extern "C" {

int avrf_pool_create(int suite, int device, int kind, int n_slots, int n_lanes, int lane_depth, int n_threads, int hash_group, avrf_pool **out) {
  if (!out || suite < 0 || suite >= AVRF_N_SUITES || (kind != 1 && kind != 2) || n_slots < 1 || n_slots > 4096 || n_lanes < 1 || n_lanes > 64 ||
      lane_depth < 0 || lane_depth > 64 || n_threads < 1 || n_threads > 64 || hash_group < 0 || hash_group > 16) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (n_threads > n_slots) n_threads = n_slots;
  if (n_lanes > n_slots) n_lanes = n_slots;
  if (n_lanes < n_threads) n_lanes = n_threads;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return AVRF_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  avrf_pool *P = new avrf_pool();
  P->suite = suite; P->device = device; P->kind = kind;
  P->group = hash_group ? hash_group : 1;
  P->depth = lane_depth ? lane_depth : 3;
  const bool ext = msm_te_pending_supported(suite);                    // (else: the chain's results live in the lane's workspace, one chain per lane)
  if (!ext) P->depth = 1;
  if (P->group > 1 && !sha512_mb_available()) P->group = 1;
  P->slots = std::vector<Slot>(n_slots); P->lanes = std::vector<Lane>(n_lanes); P->workers = std::vector<Worker>(n_threads);
  bool ok = true;
  for (auto &L : P->lanes) ok = ok && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess;
  for (auto &S : P->slots) {
    ok = ok && ctx_create(suite, device, /*lane_owner=*/false, &S.c) == AVRF_OK;
    ok = ok && hipEventCreateWithFlags(&S.ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
    if (ok && ext) S.c->pend = &S.pend;
  }
  for (int w = 0; w < n_threads && ok; w++) {
    Worker &W = P->workers[w];
    for (int i = w; i < n_slots; i += n_threads) W.slots.push_back(i);
    for (int i = w; i < n_lanes; i += n_threads) W.lanes.push_back(&P->lanes[i]);
    W.capacity = W.lanes.size() * (size_t)P->depth;
    ok = ok && hipStreamCreateWithFlags(&W.ingest, hipStreamNonBlocking) == hipSuccess;
  }
  if (!ok) { (void)hipGetLastError(); avrf_pool_destroy(P); return AVRF_ERR_NO_DEVICE; }
  // the workers hash transcripts and stage page-locked buffers for THIS device: on a multi-socket host they run on the CPUs of
  // the device's NUMA node (host_numa.h; the part of it inside the caller's affinity mask -- a caller that bound itself keeps its choice)
  cpu_set_t node_set; bool bind = false;
  if (numa_binding_enabled()) {
    char bdf[32] = {0}; std::vector<int> cpus;
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess) { P->numa_node = numa_cpus_of_pci(bdf, sysfs_root(), cpus); bind = P->numa_node >= 0 && numa_mask(cpus, node_set); }
    else (void)hipGetLastError();
  }
  P->numa_bound = bind;
  for (int w = 0; w < n_threads; w++) {
    P->workers[w].th = std::thread([P, w] { Run(P, P->workers[w]).loop(); });
    if (bind) (void)pthread_setaffinity_np(P->workers[w].th.native_handle(), sizeof node_set, &node_set);
  }
  *out = P;
  return AVRF_OK;
}

void avrf_pool_destroy(avrf_pool *P) {
  if (!P) return;
  { std::lock_guard<std::mutex> lk(P->m); P->stop = true; }
  P->cv_work.notify_all();
  for (auto &W : P->workers) if (W.th.joinable()) W.th.join();
  (void)hipSetDevice(P->device);
  (void)hipDeviceSynchronize();
  for (auto &W : P->workers) if (W.ingest) (void)hipStreamDestroy(W.ingest);
  for (auto &S : P->slots) { if (S.ev) (void)hipEventDestroy(S.ev); S.pend.release(); if (S.c) { S.c->pend = nullptr; S.c->L = &S.c->own; S.c->stream = nullptr; S.c->run_phase = 0; avrf_ctx_destroy(S.c); } }
  for (auto &L : P->lanes) L.release();
  delete P;
}

int avrf_pool_set_validation(avrf_pool *P, int level) {
  if (!P || level < 0 || level > 2) return AVRF_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(P->m);
  for (auto &S : P->slots) S.c->validate = level;
  return AVRF_OK;
}

static int pool_submit(avrf_pool *P, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts, const uint8_t *ads,
                       const uint32_t *ad_lens, const uint8_t *proofs, uint64_t *ticket, bool wire, int validate);
int avrf_pool_submit(avrf_pool *P, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts, const uint8_t *ads,
                     const uint32_t *ad_lens, const uint8_t *proofs, uint64_t *ticket) {
  return pool_submit(P, n, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs, ticket, false, 0);
}
// the same job from the reference's wire bytes (`serialize_compressed` points, proofs as CanonicalSerialize writes them): the
// points are decompressed -- validate != 0: and checked for non-identity and prime-order-subgroup membership -- on the device
// while the batch is staged; a point that fails makes the batch's verdict AVRF_INVALID_DATA
int avrf_pool_submit_wire(avrf_pool *P, size_t n, const uint8_t *pks, const uint8_t *ios, const uint32_t *io_counts, const uint8_t *ads,
                          const uint32_t *ad_lens, const uint8_t *proofs, int validate, uint64_t *ticket) {
  return pool_submit(P, n, pks, ios, io_counts, ads, ad_lens, proofs, ticket, true, validate);
}
static int pool_submit(avrf_pool *P, size_t n, const uint8_t *pks_xy, const uint8_t *ios_xy, const uint32_t *io_counts, const uint8_t *ads,
                       const uint32_t *ad_lens, const uint8_t *proofs, uint64_t *ticket, bool wire, int validate) {
  if (!P || !ticket || (n && (!proofs || !io_counts || !ad_lens)) || (n && P->kind == 1 && !pks_xy)) return AVRF_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(P->m);
  if (P->cycling) return AVRF_ERR_BAD_ARG;
  for (;;) {
    // the worker with the fewest slots in flight takes the batch
    int best = -1; size_t best_load = ~(size_t)0; bool any_running = false;
    for (size_t w = 0; w < P->workers.size(); w++) {
      size_t load = 0; int free_slot = -1;
      for (int si : P->workers[w].slots) { const int st = P->slots[si].state; if (st == S_FREE) { if (free_slot < 0) free_slot = si; } else { load++; if (st != S_DONE) any_running = true; } }
      if (free_slot >= 0 && load < best_load) { best_load = load; best = free_slot; }
    }
    if (best >= 0) {
      Slot &S = P->slots[best];
      S.n = n; S.pks = pks_xy; S.ios = ios_xy; S.io_counts = io_counts; S.ads = ads; S.ad_lens = ad_lens; S.proofs = proofs;
      S.from_host = true; S.wire = wire; S.validate = validate; S.resident_invalid = false; S.ticket = P->next_ticket++; S.status = 0;
      S.state = S_SUBMITTED;
      *ticket = S.ticket;
      P->cv_work.notify_all();
      return AVRF_OK;
    }
    if (!any_running) return AVRF_ERR_BAD_ARG;                         // every slot holds an uncollected verdict: avrf_pool_wait first
    P->cv_done.wait(lk);
  }
}

int avrf_pool_wait(avrf_pool *P, uint64_t ticket, int *status) {
  if (!P || !status) return AVRF_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(P->m);
  Slot *S = nullptr;
  for (auto &s : P->slots) if (s.ticket == ticket && s.state != S_FREE) { S = &s; break; }
  if (!S) return AVRF_ERR_BAD_ARG;
  P->cv_done.wait(lk, [&] { return S->state == S_DONE; });
  *status = S->status;
  S->state = S_FREE;
  P->cv_done.notify_all();
  return AVRF_OK;
}

int avrf_pool_resubmit(avrf_pool *P, uint64_t ticket, int from_host, uint64_t *new_ticket) {
  if (!P || !new_ticket) return AVRF_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(P->m);
  if (P->cycling) return AVRF_ERR_BAD_ARG;
  for (auto &S : P->slots) if (S.ticket == ticket && S.state == S_FREE && S.has_batch) {
    S.from_host = from_host != 0; S.ticket = P->next_ticket++; S.status = 0; S.state = S_SUBMITTED;
    *new_ticket = S.ticket;
    P->cv_work.notify_all();
    return AVRF_OK;
  }
  return AVRF_ERR_BAD_ARG;
}

int avrf_pool_cycle(avrf_pool *P, int from_host, uint64_t steps_block, double min_seconds, uint64_t max_steps, int expect_status,
                    uint64_t *steps_done, uint64_t *mismatches, double *seconds) {
  if (!P || !steps_block || !steps_done) return AVRF_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(P->m);
  if (P->cycling) return AVRF_ERR_BAD_ARG;
  size_t usable = 0;
  for (auto &S : P->slots) { if (S.state != S_FREE) return AVRF_ERR_BAD_ARG; if (S.has_batch) usable++; }
  if (!usable) return AVRF_ERR_BAD_ARG;
  P->cycling = true; P->cyc_closed = false; P->cyc_from_host = from_host != 0;
  P->cyc_issued = 0; P->cyc_done = 0; P->cyc_mismatch = 0; P->cyc_block = steps_block; P->cyc_target = steps_block;
  P->cyc_cap = max_steps ? (max_steps / steps_block ? max_steps / steps_block : 1) * steps_block : steps_block * 100000;
  P->cyc_expect = expect_status; P->cyc_min_us = min_seconds * 1e6;
  P->cyc_t0 = P->cyc_t_last = now_us();
  // deal the first steps round-robin over the workers so that every host thread starts with work
  size_t maxs = 0; for (auto &W : P->workers) maxs = W.slots.size() > maxs ? W.slots.size() : maxs;
  for (size_t k = 0; k < maxs; k++)
    for (auto &W : P->workers) if (k < W.slots.size()) {
      Slot &S = P->slots[W.slots[k]];
      if (S.has_batch && P->take_locked()) { S.from_host = P->cyc_from_host; S.ticket = P->next_ticket++; S.state = S_SUBMITTED; }
    }
  P->cv_work.notify_all();
  P->cv_done.wait(lk, [&] { return P->cyc_done == P->cyc_issued && (P->cyc_closed || P->cyc_mismatch); });
  // a mismatch stops the reissue of THAT slot only; wait for everything in flight
  P->cyc_closed = true;
  P->cv_done.wait(lk, [&] { return P->cyc_done == P->cyc_issued; });
  P->cycling = false;
  *steps_done = P->cyc_done;
  if (mismatches) *mismatches = P->cyc_mismatch;
  if (seconds) *seconds = (P->cyc_t_last - P->cyc_t0) * 1e-6;
  return AVRF_OK;
}

int avrf_pool_stats(avrf_pool *P, int reset, double *out, size_t n_out) {
  // out[0..5]: host-thread CPU microseconds, all workers: begin, collect, hash, launch, end, total (total is final at destroy only);
  // out[6] hash groups, out[7] hashed transcripts, out[8] blocking waits, out[9] k_accumulate ms total, out[10] k_accumulate launches
  if (!P || !out || n_out < 11) return AVRF_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(P->m);
  for (size_t i = 0; i < n_out; i++) out[i] = 0;
  for (auto &W : P->workers) {
    for (int i = 0; i < 6; i++) out[i] += W.cpu_us[i];
    out[6] += (double)W.hash_groups; out[7] += (double)W.hash_msgs; out[8] += (double)W.waits;
    if (reset) { for (double &v : W.cpu_us) v = 0; W.hash_groups = W.hash_msgs = W.waits = 0; }
  }
  for (auto &L : P->lanes) {
    out[9] += L.ws.accum_ms_total; out[10] += (double)L.ws.accum_launches;
    if (reset) { L.ws.accum_ms_total = 0; L.ws.accum_launches = 0; }
  }
  return AVRF_OK;
}

// Page-locked host memory for the buffers a caller hands to avrf_pool_submit / the *_stage entry points: copies from it are
// DMA transfers (no staging through the runtime's bounce buffers, no host time), see include/avrf.h "Ownership".
// (The pages are allocated -- and pinned -- by the calling thread: under the default local policy they come from the node that
// thread runs on, so for the duration of the call the thread is moved next to the CURRENT device (hipGetDevice), as the pool's
// workers are; a caller already bound to one node, e.g. a rank of bench.py, is left where it is.)
int avrf_host_alloc(size_t bytes, void **out) {
  if (!out) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (!bytes) return AVRF_OK;
  cpu_set_t saved, node_set; bool moved = false;
  if (numa_binding_enabled() && sched_getaffinity(0, sizeof saved, &saved) == 0) {
    int dev = 0; char bdf[32] = {0}; std::vector<int> cpus;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, dev) == hipSuccess &&
        numa_cpus_of_pci(bdf, sysfs_root(), cpus) >= 0 && numa_mask(cpus, node_set))
      moved = sched_setaffinity(0, sizeof node_set, &node_set) == 0;
    else (void)hipGetLastError();
  }
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (moved) (void)sched_setaffinity(0, sizeof saved, &saved);
  if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return AVRF_ERR_NO_DEVICE; }
  return AVRF_OK;
}
// test hook and diagnostic: NUMA node (-1: unknown) and CPUs of a PCI device under a sysfs root ("" or NULL = "/"); returns the
// number of CPUs written to cpus_out (at most cap)
int avrf_numa_cpus_of_pci(const char *pci_bdf, const char *sysfs_root_dir, int32_t *cpus_out, size_t cap, int32_t *node_out) {
  std::vector<int> cpus;
  const int node = numa_cpus_of_pci(pci_bdf, sysfs_root_dir, cpus);
  if (node_out) *node_out = node;
  size_t k = 0;
  for (; k < cpus.size() && k < cap && cpus_out; k++) cpus_out[k] = cpus[k];
  return (int)k;
}
// NUMA node the pool's workers were bound to (-1: not bound / unknown)
int avrf_pool_numa_node(avrf_pool *P) { return P && P->numa_bound ? P->numa_node : -1; }
void avrf_host_free(void *p) { if (p) (void)hipHostFree(p); }
int avrf_host_register(void *p, size_t bytes) {
  if (!p || !bytes) return AVRF_ERR_BAD_ARG;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return AVRF_ERR_NO_DEVICE; }
  return AVRF_OK;
}
int avrf_host_unregister(void *p) {
  if (!p) return AVRF_ERR_BAD_ARG;
  if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return AVRF_ERR_NO_DEVICE; }
  return AVRF_OK;
}

}  // extern "C"
