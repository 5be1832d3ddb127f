// msm.hip -- Pippenger bucket MSM for twisted-Edwards curves on gfx950 (CDNA4).
//
// Replaces arkworks `VariableBaseMSM::msm_unchecked` (third-party ark-ec 0.6) at the
// reference call sites src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411.
// Any correct MSM yields the same group element; parity is on the normalised result
// (SURVEY.md A.9), so the decomposition below is designed for the GPU, not copied:
//
//   k_digits     one lane per scalar: signed c-bit digits for every window, key = bucket|sign,
//                global histogram per (window, bucket)
//   k_scan       one workgroup: exclusive prefix of the histogram (entry offsets) and of the
//                per-bucket LANE counts  lanes_b = ceil(count_b / SEG)
//   k_scatter    counting-sort scatter of term indices into bucket order
//   k_accumulate load-balanced: every lane owns <= SEG consecutive entries of ONE bucket (big
//                buckets simply get more lanes), does its mixed additions (8M each) on gathered
//                96-byte precomputed points, then a wave-level SEGMENTED shuffle reduction
//                folds the lanes of a bucket; runs that cross a wave boundary leave a partial
//   k_fixup      per bucket: identity for empty buckets, sum of wave partials for split ones
//   k_bits       bucket reduction without a serial running sum:  sum_b b*B_b =
//                sum_k 2^k * (sum of buckets whose index has bit k set); one workgroup per
//                (window, bit) tree-reduces its half of the buckets through LDS
//   host         Horner over the nwin*c bit sums (<= 2*260 point ops)
//
// Layout in HBM: points AoS te_pre (x|y|k, 96 B, gathered whole by one lane with 6 dwordx4
// loads); keys/sorted SoA per window (coalesced); buckets AoS te_ext (128 B).
#include "msm.h"
#include "te.h"
#include <stdio.h>
#include <stdlib.h>

namespace avrf {

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "avrf: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); abort(); } } while (0)

// ---------------------------------------------------------------- conversions

template <class S>
__global__ void k_pre_from_affine(const uint8_t *__restrict__ xy, uint32_t n, te_pre *__restrict__ out,
                                  uint32_t *__restrict__ flag, int check_curve) {
  using Fq = typename S::Fq;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp x = fp_load_le(xy + 64 * (size_t)i), y = fp_load_le(xy + 64 * (size_t)i + 32);
  uint32_t f = 0;
  if (ge_p<Fq>(x) || ge_p<Fq>(y)) f |= 1;
  fp xm = fp_to_mont<Fq>(x), ym = fp_to_mont<Fq>(y);
  if (check_curve && !te_on_curve<S>(xm, ym)) f |= 2;
  store_pre(out + i, te_make_pre<S>(xm, ym));
  if (f) atomicOr(flag, f);
}

void launch_pre_from_affine(int suite, const uint8_t *d_xy, size_t n, te_pre_raw *d_pre, uint32_t *d_flag,
                            int check_curve, hipStream_t stream) {
  if (!n) return;
  dim3 g((unsigned)((n + 255) / 256)), b(256);
  if (suite == 0) hipLaunchKernelGGL(k_pre_from_affine<SuiteBandersnatch>, g, b, 0, stream, d_xy, (uint32_t)n, (te_pre *)d_pre, d_flag, check_curve);
  else hipLaunchKernelGGL(k_pre_from_affine<SuiteBabyJubJub>, g, b, 0, stream, d_xy, (uint32_t)n, (te_pre *)d_pre, d_flag, check_curve);
}

// ---------------------------------------------------------------- digits + histogram

// signed digit of window w with carry chain; digits in [-(2^(c-1)-1), 2^(c-1)]
__global__ void k_digits(const uint32_t *__restrict__ scalars, uint32_t n, int c, int nwin,
                         uint32_t *__restrict__ keys, uint32_t *__restrict__ counts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[9];
  const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * (size_t)i);
  uint4 a = p[0], b = p[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; s[8] = 0;
  const uint32_t nb = 1u << (c - 1), mask = (1u << c) - 1;
  uint32_t carry = 0;
  for (int w = 0; w < nwin; w++) {
    int bit = w * c;
    uint32_t v = 0;
    if (bit < 256) {
      int li = bit >> 5, sh = bit & 31;
      uint64_t two = (uint64_t)s[li] | ((uint64_t)s[li + 1] << 32);
      v = (uint32_t)(two >> sh) & mask;
    }
    v += carry;
    uint32_t key = 0;
    if (v > nb) { key = ((1u << c) - v) | 0x80000000u; carry = 1; }   // negative digit: bucket 2^c - v
    else { key = v; carry = 0; }
    keys[(size_t)w * n + i] = key;
    uint32_t bucket = key & 0x7fffffffu;
    if (bucket) atomicAdd(&counts[(size_t)w * nb + bucket - 1], 1u);
  }
}

// One workgroup of 1024 lanes: exclusive scans of counts[] (entry offsets) and of the per-bucket lane
// counts ceil(count / seg) (lane offsets); both arrays get total + 1 entries.
__global__ void k_scan(const uint32_t *__restrict__ counts, uint32_t total, uint32_t seg,
                       uint32_t *__restrict__ offsets, uint32_t *__restrict__ lane_off) {
  __shared__ uint32_t part[1024], partl[1024];
  uint32_t t = threadIdx.x;
  uint32_t per = (total + 1023) / 1024;
  uint32_t lo = t * per, hi = lo + per; if (hi > total) hi = total; if (lo > total) lo = total;
  uint32_t sum = 0, suml = 0;
  for (uint32_t i = lo; i < hi; i++) { uint32_t c = counts[i]; sum += c; suml += (c + seg - 1) / seg; }
  part[t] = sum; partl[t] = suml;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    uint32_t v = (t >= off) ? part[t - off] : 0, vl = (t >= off) ? partl[t - off] : 0;
    __syncthreads();
    part[t] += v; partl[t] += vl;
    __syncthreads();
  }
  uint32_t run = part[t] - sum, runl = partl[t] - suml;
  for (uint32_t i = lo; i < hi; i++) { uint32_t c = counts[i]; offsets[i] = run; lane_off[i] = runl; run += c; runl += (c + seg - 1) / seg; }
  if (t == 1023) { offsets[total] = part[1023]; lane_off[total] = partl[1023]; }
}

__global__ void k_scatter(const uint32_t *__restrict__ keys, uint32_t n, int c, int nwin,
                          const uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursors,
                          uint32_t *__restrict__ sorted) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  int w = blockIdx.y;
  if (i >= n) return;
  const uint32_t nb = 1u << (c - 1);
  uint32_t key = keys[(size_t)w * n + i];
  uint32_t bucket = key & 0x7fffffffu;
  if (!bucket) return;
  size_t slot = (size_t)w * nb + bucket - 1;
  uint32_t pos = offsets[slot] + atomicAdd(&cursors[slot], 1u);
  sorted[pos] = i | (key & 0x80000000u);
}

// ---------------------------------------------------------------- bucket accumulation

AVRF_DI te_ext shfl_down_ext(const te_ext &p, int delta) {
  te_ext r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.x.v[i] = __shfl_down(p.x.v[i], delta); r.y.v[i] = __shfl_down(p.y.v[i], delta);
    r.t.v[i] = __shfl_down(p.t.v[i], delta); r.z.v[i] = __shfl_down(p.z.v[i], delta);
  }
  return r;
}

// Lane t owns a segment of bucket `slot` (the slot with lane_off[slot] <= t < lane_off[slot+1]).
// part[2*wave + k]: partial of the run of wave `wave` that includes lane 0 (k = 0) or that starts
// later and runs past lane 63 (k = 1); complete runs are written straight to buckets[].
template <class S>
__global__ void __launch_bounds__(256)
k_accumulate(const te_pre *__restrict__ pre, const uint32_t *__restrict__ sorted,
             const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ lane_off,
             uint32_t nslots, te_ext *__restrict__ buckets, te_ext *__restrict__ part) {
  using Fq = typename S::Fq;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wave = t >> 6;
  const uint32_t total_lanes = lane_off[nslots];
  const bool live = t < total_lanes;
  uint32_t slot = 0xffffffffu, l0 = 0, nl = 0;
  te_ext acc = te_identity<S>();
  if (live) {
    uint32_t lo = 0, hi = nslots;                       // last slot with lane_off[slot] <= t
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (lane_off[mid] <= t) lo = mid; else hi = mid; }
    slot = lo; l0 = lane_off[slot]; nl = lane_off[slot + 1] - l0;
    const uint32_t e0 = offsets[slot], cnt = offsets[slot + 1] - e0;
    const uint32_t per = (cnt + nl - 1) / nl, r = t - l0;
    uint32_t b = e0 + r * per, e = b + per; if (e > e0 + cnt) e = e0 + cnt;
    for (uint32_t i = b; i < e; i++) {
      uint32_t idx = sorted[i];
      te_pre q = load_pre(pre + (idx & 0x7fffffffu));
      if (idx & 0x80000000u) { q.x = fp_neg<Fq>(q.x); q.k = fp_neg<Fq>(q.k); }
      acc = te_madd<S>(acc, q);
    }
  }
  // segmented reduction by doubling: after step `off` a run head holds the sum of min(run, 2*off) lanes
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t oslot = __shfl_down(slot, off);
    bool take = live && (lane + off < 64) && (oslot == slot);
    if (!__any(take)) break;                            // no run in this wave is longer than `off`
    te_ext o = shfl_down_ext(acc, off);
    if (take) acc = te_add<S>(acc, o);
  }
  uint32_t pslot = __shfl_up(slot, 1);
  bool head = live && (lane == 0 || pslot != slot);
  if (head) {
    bool complete = (l0 >= (wave << 6)) && (l0 + nl <= (wave << 6) + 64);
    if (complete) store_ext(buckets + slot, acc);
    else store_ext(part + 2 * (size_t)wave + (lane == 0 ? 0 : 1), acc);
  }
}

template <class S>
__global__ void __launch_bounds__(256)
k_fixup(const uint32_t *__restrict__ lane_off, uint32_t nslots, const te_ext *__restrict__ part, te_ext *__restrict__ buckets) {
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  uint32_t l0 = lane_off[slot], nl = lane_off[slot + 1] - l0;
  if (nl == 0) { store_ext(buckets + slot, te_identity<S>()); return; }
  uint32_t wa = l0 >> 6, wb = (l0 + nl - 1) >> 6;
  if (wa == wb) return;                                  // complete inside one wave: already written
  te_ext acc = load_ext(part + 2 * (size_t)wa + ((l0 & 63) == 0 ? 0 : 1));
  for (uint32_t w = wa + 1; w <= wb; w++) acc = te_add<S>(acc, load_ext(part + 2 * (size_t)w));
  store_ext(buckets + slot, acc);
}

// ---------------------------------------------------------------- bucket reduction by index bits

// grid = nwin * c workgroups; workgroup (w, k) sums the buckets of window w whose index b (1..nb)
// has bit k set.  sum_b b*B_b = sum_k 2^k * T_k.
template <class S>
__global__ void __launch_bounds__(256)
k_bits(const te_ext *__restrict__ buckets, int c, te_ext *__restrict__ out) {
  __shared__ te_ext sh[256];
  int w = blockIdx.x / c, k = blockIdx.x % c;
  const uint32_t nb = 1u << (c - 1);
  const te_ext *B = buckets + (size_t)w * nb;
  te_ext acc = te_identity<S>();
  if (k == c - 1) {
    if (threadIdx.x == 0) acc = load_ext(B + (nb - 1));       // only b = nb = 2^(c-1)
  } else {
    uint32_t cnt = nb >> 1;                                    // b in [1, nb-1] with bit k set
    for (uint32_t m = threadIdx.x; m < cnt; m += blockDim.x) {
      uint32_t b = ((m >> k) << (k + 1)) | (1u << k) | (m & ((1u << k) - 1));
      acc = te_add<S>(acc, load_ext(B + (b - 1)));
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) { acc = te_add<S>(acc, sh[threadIdx.x + s]); sh[threadIdx.x] = acc; }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_ext(out + blockIdx.x, acc);
}

// ---------------------------------------------------------------- host engine

MsmPlan msm_plan(size_t n, int scalar_bits) {
  MsmPlan p;
  int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = lg - 4; if (c < 4) c = 4; if (c > 14) c = 14;
  p.c = c; p.lpb = 16;                                   // lpb = SEG: entries per lane
  if (const char *e = getenv("AVRF_MSM_C")) { int v = atoi(e); if (v >= 2 && v <= 20) p.c = v; }
  if (const char *e = getenv("AVRF_MSM_SEG")) { int v = atoi(e); if (v >= 1 && v <= 1024) p.lpb = v; }
  p.nb = 1 << (p.c - 1);
  p.nwin = (scalar_bits + 1 + p.c - 1) / p.c;
  return p;
}

void MsmWorkspace::ensure(size_t n, const MsmPlan &p) {
  size_t nbk = (size_t)p.nwin * p.nb, nbits = (size_t)p.nwin * p.c;
  size_t need_n = (size_t)p.nwin * n;
  if (need_n > cap_n) {
    if (keys) HIP_CHECK(hipFree(keys));
    if (sorted) HIP_CHECK(hipFree(sorted));
    HIP_CHECK(hipMalloc(&keys, need_n * 4)); HIP_CHECK(hipMalloc(&sorted, need_n * 4));
    cap_n = need_n;
  }
  if (nbk > cap_buckets) {
    if (counts) HIP_CHECK(hipFree(counts));
    if (offsets) HIP_CHECK(hipFree(offsets));
    if (lane_off) HIP_CHECK(hipFree(lane_off));
    if (buckets) HIP_CHECK(hipFree(buckets));
    HIP_CHECK(hipMalloc(&counts, nbk * 4)); HIP_CHECK(hipMalloc(&offsets, (nbk + 1) * 4)); HIP_CHECK(hipMalloc(&lane_off, (nbk + 1) * 4));
    HIP_CHECK(hipMalloc(&buckets, nbk * sizeof(te_ext_raw)));
    cap_buckets = nbk;
  }
  // lanes <= entries/seg + buckets  =>  waves <= that / 64 + 1; two partial slots per wave
  size_t max_lanes = need_n / (size_t)(p.lpb > 0 ? p.lpb : 1) + nbk + 64;
  size_t need_part = 2 * (max_lanes / 64 + 2);
  if (need_part > cap_part) {
    if (part) HIP_CHECK(hipFree(part));
    HIP_CHECK(hipMalloc(&part, need_part * sizeof(te_ext_raw)));
    cap_part = need_part;
  }
  if (nbits > cap_bits) {
    if (bits) HIP_CHECK(hipFree(bits));
    if (bits_host) HIP_CHECK(hipHostFree(bits_host));
    HIP_CHECK(hipMalloc(&bits, nbits * sizeof(te_ext_raw)));
    HIP_CHECK(hipHostMalloc(&bits_host, nbits * sizeof(te_ext_raw)));
    cap_bits = nbits;
  }
}
void MsmWorkspace::release() {
  if (keys) (void)hipFree(keys); if (sorted) (void)hipFree(sorted); if (counts) (void)hipFree(counts);
  if (offsets) (void)hipFree(offsets); if (lane_off) (void)hipFree(lane_off); if (buckets) (void)hipFree(buckets);
  if (part) (void)hipFree(part); if (bits) (void)hipFree(bits);
  if (bits_host) (void)hipHostFree(bits_host);
  if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); ev0 = ev1 = nullptr;
  keys = sorted = counts = offsets = lane_off = nullptr; buckets = part = bits = bits_host = nullptr;
  cap_n = cap_buckets = cap_bits = cap_part = 0;
}

template <class S>
static int msm_impl(const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws,
                    hipStream_t stream, HostExt *out) {
  using HT = HostTe<S>;
  *out = HT::identity();
  if (n == 0) return 0;
  MsmPlan p = msm_plan(n, S::Fr::BITS);
  ws.ensure(n, p);
  const uint32_t nbk = (uint32_t)p.nwin * p.nb;
  HIP_CHECK(hipMemsetAsync(ws.counts, 0, (size_t)nbk * 4, stream));
  dim3 b256(256), gn((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(k_digits, gn, b256, 0, stream, d_scalars, (uint32_t)n, p.c, p.nwin, ws.keys, ws.counts);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, ws.counts, nbk, (uint32_t)p.lpb, ws.offsets, ws.lane_off);
  HIP_CHECK(hipMemsetAsync(ws.counts, 0, (size_t)nbk * 4, stream));
  hipLaunchKernelGGL(k_scatter, dim3(gn.x, p.nwin), b256, 0, stream, ws.keys, (uint32_t)n, p.c, p.nwin, ws.offsets, ws.counts, ws.sorted);
  const te_pre *pre = (const te_pre *)d_pre;
  te_ext *bk = (te_ext *)ws.buckets;
  // upper bound on the lane count (the exact number lives in lane_off[nbk] on the device)
  size_t max_lanes = ((size_t)p.nwin * n) / (size_t)p.lpb + nbk;
  dim3 ga((unsigned)((max_lanes + 255) / 256));
  if (!ws.ev0) { HIP_CHECK(hipEventCreate(&ws.ev0)); HIP_CHECK(hipEventCreate(&ws.ev1)); }
  HIP_CHECK(hipEventRecord(ws.ev0, stream));
  hipLaunchKernelGGL(k_accumulate<S>, ga, b256, 0, stream, pre, ws.sorted, ws.offsets, ws.lane_off, nbk, bk, (te_ext *)ws.part);
  HIP_CHECK(hipEventRecord(ws.ev1, stream));
  hipLaunchKernelGGL(k_fixup<S>, dim3((nbk + 255) / 256), b256, 0, stream, ws.lane_off, nbk, (const te_ext *)ws.part, bk);
  const int nbits = p.nwin * p.c;
  hipLaunchKernelGGL(k_bits<S>, dim3(nbits), b256, 0, stream, bk, p.c, (te_ext *)ws.bits);
  HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.bits, (size_t)nbits * sizeof(te_ext_raw), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, ws.ev0, ws.ev1));
  ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = p;
  HostExt acc = HT::identity();
  for (int i = nbits - 1; i >= 0; i--) {
    acc = HT::dbl(acc);
    acc = HT::add(acc, HT::from_raw32(ws.bits_host[i].w));
  }
  *out = acc;
  return 0;
}

int msm_te_device(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n,
                  MsmWorkspace &ws, hipStream_t stream, HostExt *out) {
  if (suite == 0) return msm_impl<SuiteBandersnatch>(d_pre, d_scalars, n, ws, stream, out);
  if (suite == 1) return msm_impl<SuiteBabyJubJub>(d_pre, d_scalars, n, ws, stream, out);
  return -1;
}

}  // namespace avrf
