// host_pool.h -- one bounded, persistent pool of host worker threads for the per-proof bookkeeping between device rounds
// (ring prover / verifiers: transcripts, witness accumulators, decompression; the reference runs the same per-item
// preparation under rayon, src/ring.rs:1081-1086).
//
// Every context of the process shares the pool: min(64, CPUs in the affinity mask) threads, created once (what this replaced
// spawned up to 32 std::threads per parallel_for call, several times per chunk, per context).  The size deliberately follows
// the hardware threads the process may run on, NOT the cgroup CPU quota: the work between device rounds comes in short
// bursts, and a CFS quota meters CPU time per period, not parallelism -- a burst spread over 32 hardware threads ends sooner
// than the same burst on 16 and costs the same quota (measured on a 256-thread box with a 16-CPU quota, 4 contexts x 1024
// proofs, gpurun_out/r3b: ring batch verification 135 / 183 / 201 k/s and independent verification 74 / 87 / 103 k/s with
// 16 / 32 / 64 threads; the prover does not move: 11.2-11.3 k proofs/s).  AVRF_HOST_THREADS overrides the size.
// The calling thread works on its own job too, so a job always makes progress even when every worker is busy elsewhere.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

namespace avrf {

// hardware threads this process may run on
inline size_t host_cpu_affinity() {
  size_t n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { int c = CPU_COUNT(&set); if (c > 0 && (size_t)c < n) n = (size_t)c; }
  return n ? n : 1;
}
// CPU time this process may use, in CPUs: cgroup v2 quota / period (rounded up), bounded by the affinity mask
inline size_t host_cpu_quota() {
  size_t n = host_cpu_affinity();
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0}; long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && q[0] != 'm') {
      long long quota = atoll(q);
      if (quota > 0) { size_t c = (size_t)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
    }
    fclose(f);
  }
  return n ? n : 1;
}

class HostPool {
  struct Job {
    const std::function<void(size_t)> *fn; size_t n;
    std::atomic<size_t> next{0}, done{0};
  };
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::deque<std::shared_ptr<Job>> jobs_;
  std::vector<std::thread> workers_;

  // take items of `j` until none is left; true when this call finished the job's last item
  bool drain(Job &j) {
    bool last = false;
    for (size_t i; (i = j.next.fetch_add(1, std::memory_order_relaxed)) < j.n;) {
      (*j.fn)(i);
      if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.n) last = true;
    }
    return last;
  }
  void worker() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      while (!jobs_.empty() && jobs_.front()->next.load(std::memory_order_relaxed) >= jobs_.front()->n) jobs_.pop_front();
      if (jobs_.empty()) { cv_work_.wait(lk); continue; }
      std::shared_ptr<Job> j = jobs_.front();
      lk.unlock();
      const bool last = drain(*j);
      lk.lock();
      if (last) cv_done_.notify_all();
    }
  }
  explicit HostPool(size_t nthreads) {
    for (size_t t = 0; t < nthreads; t++) workers_.emplace_back([this] { worker(); });
    for (auto &w : workers_) w.detach();      // the pool lives as long as the process
  }

 public:
  size_t size() const { return workers_.size() + 1; }
  static HostPool &get() {
    static HostPool *p = [] {
      size_t nt = host_cpu_affinity(); if (nt > 64) nt = 64;
      if (const char *e = getenv("AVRF_HOST_THREADS")) { long v = atol(e); if (v >= 1 && v <= 256) nt = (size_t)v; }
      return new HostPool(nt > 1 ? nt - 1 : 0);   // + the calling thread
    }();
    return *p;
  }
  // fn(i) for every i < n, on the pool and the calling thread; returns when all are done.  `grain` = items worth one worker's
  // wake-up (a sleeping thread costs tens of microseconds to bring in: a chunk's 512 transcript steps of ~10 us each are work for
  // a few dozen threads, eight pairing checks of a millisecond each for eight)
  void run(size_t n, const std::function<void(size_t)> &fn, size_t grain = 1) {
    if (n == 0) return;
    if (n == 1 || workers_.empty()) { for (size_t i = 0; i < n; i++) fn(i); return; }
    auto j = std::make_shared<Job>(); j->fn = &fn; j->n = n;
    { std::lock_guard<std::mutex> lk(mu_); jobs_.push_back(j); }
    if (grain < 1) grain = 1;
    const size_t wake = (n - 1 + grain - 1) / grain;                 // the caller takes a share too
    if (wake >= workers_.size()) cv_work_.notify_all();
    else for (size_t i = 0; i < wake; i++) cv_work_.notify_one();
    drain(*j);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return j->done.load(std::memory_order_acquire) == j->n; });
  }
};

template <class Fn> static void parallel_for(size_t n, Fn fn, size_t grain = 1) {
  const std::function<void(size_t)> f = [&](size_t i) { fn(i); };
  HostPool::get().run(n, f, grain);
}

}  // namespace avrf
