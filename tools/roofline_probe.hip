// tools/roofline_probe.hip -> ark_vrf_amd/libavrf_probe.so: what bench.py measures IN ITS OWN RUN so that every `frac` of its line can
// be recomputed from fields of the same line.  Measurement infrastructure, not product: libavrf.so does not link it.
//   avrf_probe_mad_stream   the integer multiplier's issue rate on this chip: v_mad_u64_u32 issued by hand (sixteen independent 64-bit
//                           accumulators per lane, the stream tools/ubench.hip k_mad_asm runs), plus the shader clock the stream ran at,
//                           read INSIDE the kernel: delta s_memtime (shader-clock counter) / delta s_memrealtime (constant 100 MHz).
//   avrf_probe_valu_stream  the same for a plain 32-bit VALU instruction (v_xor_b32 chain issued by hand): the architectural 64 lanes / clk / CU.
//   avrf_probe_clock        one wave spinning for `spin_us` on its own stream while whatever else runs on the device: the shader clock
//                           UNDER that load (bench.py runs it beside k_accumulate launches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "../ark_vrf_amd/csrc/curves.h"      // the shipped mixed additions (bare loops: what the instruction stream itself sustains)

#define PCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -(int)e_ - 1000; } while (0)

template <int WHICH> __global__ void __launch_bounds__(256) k_stream(uint32_t *out, uint64_t *clk, int iters, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  uint64_t a8 = a0 * 17, a9 = a0 * 19, a10 = a0 * 21, a11 = a0 * 23, a12 = a0 * 25, a13 = a0 * 27, a14 = a0 * 29, a15 = a0 * 31;
  uint32_t x0 = seed * 3 + blockIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  uint32_t yv = (seed | 1) + (threadIdx.x & 1);
  uint32_t b0 = (uint32_t)a0, b1 = (uint32_t)a1, b2 = (uint32_t)a2, b3 = (uint32_t)a3, b4 = (uint32_t)a4, b5 = (uint32_t)a5, b6 = (uint32_t)a6, b7 = (uint32_t)a7;
  const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#define MAD_(acc, x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(yv) : "vcc")
#define XOR_(acc, x) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc) : "v"(x))
  for (int i = 0; i < iters; i++) {
    if (WHICH == 0) {
      MAD_(a0, x0); MAD_(a1, x1); MAD_(a2, x2); MAD_(a3, x3); MAD_(a4, x0); MAD_(a5, x1); MAD_(a6, x2); MAD_(a7, x3);
      MAD_(a8, x0); MAD_(a9, x1); MAD_(a10, x2); MAD_(a11, x3); MAD_(a12, x0); MAD_(a13, x1); MAD_(a14, x2); MAD_(a15, x3);
    } else {
      XOR_(b0, x0); XOR_(b1, x1); XOR_(b2, x2); XOR_(b3, x3); XOR_(b4, x0); XOR_(b5, x1); XOR_(b6, x2); XOR_(b7, x3);
      XOR_(b0, x1); XOR_(b1, x2); XOR_(b2, x3); XOR_(b3, x0); XOR_(b4, x1); XOR_(b5, x2); XOR_(b6, x3); XOR_(b7, x0);
    }
  }
#undef MAD_
#undef XOR_
  const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ a8 ^ a9 ^ a10 ^ a11 ^ a12 ^ a13 ^ a14 ^ a15;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
  if ((threadIdx.x & 63) == 0) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    clk[2 * (size_t)w] = t1 - t0; clk[2 * (size_t)w + 1] = r1 - r0;
  }
}
__global__ void __launch_bounds__(64) k_clock(uint64_t *clk, uint64_t spin_ticks, uint32_t count) {
  for (uint32_t s = 0; s < count; s++) {
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    uint64_t r1 = r0;
    while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { clk[2 * (size_t)s] = t1 - t0; clk[2 * (size_t)s + 1] = r1 - r0; }
  }
}

// The bare mixed-addition loops of the shipped arithmetic, no memory traffic: WHICH 0 = teu_madd<Bandersnatch> (k_accumulate<TeCurve>'s
// policy), 1 / 2 = the G1 accumulation policy on BLS12-381 / BN254.  The G1 bases are pseudo-random field elements, not curve points: the
// addition's instruction path is the same (its exceptional cases -- P = +-Q, identity -- do not occur either way) and only time is measured.
template <int WHICH> __global__ void __launch_bounds__(256, 2) k_madd_loop(uint32_t *out, const uint32_t *bases, int iters, uint32_t seed) {
  using namespace avrf;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (WHICH == 0) {
    using S = SuiteBandersnatch; using Fq = typename S::Fq;
    te_pre g, b;
    g.x = fp_const<Fq>(S::G_X); g.y = fp_const<Fq>(S::G_Y); g.k = fp_const<Fq>(S::G_K);
    b.x = fp_const<Fq>(S::B_X); b.y = fp_const<Fq>(S::B_Y); b.k = fp_const<Fq>(S::B_K);
    te_acc_u<S> p = teu_identity<S>();
    for (int i = 0; i < iters; i++) {
      const bool neg = ((lane * 2654435761u + i * 40503u + seed) >> 13) & 1;
      const te_pre q = ((lane + i) & 2) ? g : b;
      p = teu_madd<S>(p, q, neg);
    }
    const te_ext r = teu_to_ext<S>(p);
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= r.x.v[i] ^ r.y.v[i] ^ r.t.v[i] ^ r.z.v[i];
    out[lane] = x;
  } else {
    using C = typename std::conditional<WHICH == 1, G1Bls12381, G1Bn254>::type;
    using CV = G1Curve<C>; using AC = typename CV::accum; constexpr int N = C::Fq::N;
    // (the bases come from a small table in memory, as in the accumulation kernels: two bases held in registers beside the 256-register
    // addition spilled and halved the rate)
    typename AC::acc_t p = AC::from_base(CV::load_base(bases), false);
    for (int i = 0; i < iters; i++) {
      const uint32_t h = lane * 2654435761u + i * 40503u + seed;
      p = AC::madd(p, CV::load_base(bases + (size_t)((h >> 3) & 63u) * 2 * N), (h >> 13) & 1);
    }
    alignas(16) uint32_t tmp[AC::PART_WORDS];
    AC::store_part(tmp, p);
    uint32_t x = 0;
    for (int i = 0; i < AC::PART_WORDS; i++) x ^= tmp[i];
    out[lane] = x;
  }
}
template <int WHICH> static int run_madd_loop(int device, int blocks_per_cu, int iters, int reps, double *out) {
  PCK(hipSetDevice(device));
  hipDeviceProp_t prop; PCK(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount, blocks = cus * blocks_per_cu;
  uint32_t *d_out, *d_bases; PCK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  std::vector<uint32_t> hb(64 * 24);                                      // 64 pseudo-random "points": field elements below 2^(32 N - 8)
  for (size_t i = 0; i < hb.size(); i++) hb[i] = (uint32_t)(0x9e3779b9u * (i + 1) + 0x85ebca6bu * (i >> 3)) & ((i % 12 == 11 || i % 8 == 7) ? 0x00ffffffu : 0xffffffffu);
  PCK(hipMalloc(&d_bases, hb.size() * 4)); PCK(hipMemcpy(d_bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; PCK(hipEventCreate(&e0)); PCK(hipEventCreate(&e1));
  double best_ms = 1e30;
  for (int r = 0; r < reps + 1; r++) {
    PCK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_madd_loop<WHICH>, dim3(blocks), dim3(256), 0, 0, d_out, (const uint32_t *)d_bases, iters, 777u + r);
    PCK(hipEventRecord(e1, 0)); PCK(hipEventSynchronize(e1));
    float ms; PCK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best_ms) best_ms = ms;
  }
  out[0] = (double)blocks * 256.0 * iters / (best_ms * 1e-3) * 1e-9;       // G mixed additions / s
  out[1] = best_ms; out[2] = blocks_per_cu; out[3] = iters;
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d_out); hipFree(d_bases);
  return 0;
}

static double median_mhz(std::vector<uint64_t> &h, double wall_khz) {
  std::vector<double> v;
  for (size_t i = 0; i + 1 < h.size(); i += 2) if (h[i + 1]) v.push_back((double)h[i] / (double)h[i + 1] * wall_khz * 1e-3);
  if (v.empty()) return 0.0;
  std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
  return v[v.size() / 2];
}

template <int WHICH> static int run_stream(int device, int waves_per_cu, int iters, int reps, double *out) {
  PCK(hipSetDevice(device));
  hipDeviceProp_t prop; PCK(hipGetDeviceProperties(&prop, device));
  int wall_khz = 0; PCK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device));
  const int cus = prop.multiProcessorCount, blocks = cus * waves_per_cu / 4, threads = 256;
  uint32_t *d_out; uint64_t *d_clk;
  const size_t waves = (size_t)blocks * threads / 64;
  PCK(hipMalloc(&d_out, (size_t)blocks * threads * 4)); PCK(hipMalloc(&d_clk, waves * 16));
  hipEvent_t e0, e1; PCK(hipEventCreate(&e0)); PCK(hipEventCreate(&e1));
  double best_ms = 1e30;
  std::vector<uint64_t> h(2 * waves), hbest;
  for (int r = 0; r < reps + 1; r++) {                      // first launch untimed (code load, clock ramp)
    PCK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_stream<WHICH>, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, iters, 12345u + r);
    PCK(hipEventRecord(e1, 0)); PCK(hipEventSynchronize(e1));
    float ms; PCK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best_ms) { best_ms = ms; PCK(hipMemcpy(h.data(), d_clk, waves * 16, hipMemcpyDeviceToHost)); hbest = h; }
  }
  const double ops = (double)blocks * threads * (double)iters * 16.0;
  out[0] = ops / (best_ms * 1e-3) * 1e-12;                  // T lane-operations / s
  out[1] = median_mhz(hbest, wall_khz);                     // shader clock inside the kernel, MHz (median over the waves)
  out[2] = best_ms;
  out[3] = cus;
  out[4] = prop.clockRate * 1e-3;                           // the clock the runtime reports, MHz
  out[5] = out[1] > 0 ? ops / (best_ms * 1e-3) / (out[1] * 1e6) / cus : 0.0;   // lane-operations / clk / CU at the measured clock
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d_out); hipFree(d_clk);
  return 0;
}

extern "C" {
// out[6] = {T lane-ops/s, measured shader MHz, best launch ms, CUs, runtime-reported MHz, lane-ops/clk/CU at the measured clock}
int avrf_probe_mad_stream(int device, int waves_per_cu, int iters, int reps, double *out) { return run_stream<0>(device, waves_per_cu, iters, reps, out); }
int avrf_probe_valu_stream(int device, int waves_per_cu, int iters, int reps, double *out) { return run_stream<1>(device, waves_per_cu, iters, reps, out); }
// out[4] = {G mixed additions/s, best launch ms, workgroups of 256 per CU, iterations}; which: 0 twisted Edwards (Bandersnatch), 1 G1 BLS12-381, 2 G1 BN254
int avrf_probe_madd_loop(int device, int which, int blocks_per_cu, int iters, int reps, double *out) {
  if (which == 0) return run_madd_loop<0>(device, blocks_per_cu, iters, reps, out);
  if (which == 1) return run_madd_loop<1>(device, blocks_per_cu, iters, reps, out);
  if (which == 2) return run_madd_loop<2>(device, blocks_per_cu, iters, reps, out);
  return -1;
}
// one wave on its own (non-blocking) stream: `count` samples of spin_us microseconds each, back to back; mhz[count] = the shader clock of
// every sample (the wave sleeps between its reads of the two counters: it takes no issue slots worth mentioning from what runs beside it)
int avrf_probe_clock_series(int device, double spin_us, uint32_t count, double *mhz) {
  PCK(hipSetDevice(device));
  int wall_khz = 0; PCK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device));
  hipStream_t st; PCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint64_t *d_clk; PCK(hipMalloc(&d_clk, 16 * (size_t)count));
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, d_clk, (uint64_t)(spin_us * 1e-3 * wall_khz), count);
  PCK(hipStreamSynchronize(st));
  std::vector<uint64_t> h(2 * (size_t)count);
  PCK(hipMemcpy(h.data(), d_clk, 16 * (size_t)count, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < count; i++) mhz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * wall_khz * 1e-3 : 0.0;
  hipFree(d_clk); hipStreamDestroy(st);
  return 0;
}
// one sample; out[2] = {shader MHz, microseconds actually spun}
int avrf_probe_clock(int device, double spin_us, double *out) {
  int rc = avrf_probe_clock_series(device, spin_us, 1, out);
  out[1] = spin_us;
  return rc;
}
}
