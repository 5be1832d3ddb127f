// msm.hip -- Pippenger bucket MSM for twisted-Edwards curves on gfx950 (CDNA4).
//
// Replaces arkworks `VariableBaseMSM::msm_unchecked` (third-party ark-ec 0.6) at the
// reference call sites src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411.
// Any correct MSM yields the same group element; parity is on the normalised result
// (SURVEY.md A.9), so the decomposition below is designed for the GPU, not copied:
//
//   k_digits     one lane per scalar: signed c-bit digits for every window, key = bucket|sign,
//                global histogram per (window, bucket)
//   k_scan       exclusive prefix over the nwin*nb histogram (single workgroup)
//   k_scatter    counting-sort scatter of term indices into bucket order
//   k_accumulate LPB lanes per bucket: strided mixed additions (8M each) from gathered
//                96-byte precomputed points, then a wave-shuffle tree over the LPB lanes
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

// exclusive scan of `total` counters by one workgroup of 1024 lanes
__global__ void k_scan(const uint32_t *__restrict__ counts, uint32_t total, uint32_t *__restrict__ offsets) {
  __shared__ uint32_t part[1024];
  uint32_t t = threadIdx.x;
  uint32_t per = (total + 1023) / 1024;
  uint32_t lo = t * per, hi = lo + per; if (hi > total) hi = total; if (lo > total) lo = total;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += counts[i];
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;
  for (uint32_t i = lo; i < hi; i++) { offsets[i] = run; run += counts[i]; }
  if (t == 1023) offsets[total] = part[1023];
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

AVRF_DI te_ext shfl_xor_ext(const te_ext &p, int mask) {
  te_ext r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.x.v[i] = __shfl_xor(p.x.v[i], mask); r.y.v[i] = __shfl_xor(p.y.v[i], mask);
    r.t.v[i] = __shfl_xor(p.t.v[i], mask); r.z.v[i] = __shfl_xor(p.z.v[i], mask);
  }
  return r;
}

template <class S, int LPB>
__global__ void __launch_bounds__(256)
k_accumulate(const te_pre *__restrict__ pre, const uint32_t *__restrict__ sorted,
             const uint32_t *__restrict__ offsets, uint32_t nbuckets_total, te_ext *__restrict__ buckets) {
  using Fq = typename S::Fq;
  uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t slot = gl / LPB, sub = gl % LPB;
  bool live = slot < nbuckets_total;
  uint32_t lo = 0, hi = 0;
  if (live) { lo = offsets[slot]; hi = offsets[slot + 1]; }
  te_ext acc = te_identity<S>();
  for (uint32_t e = lo + sub; e < hi; e += LPB) {
    uint32_t idx = sorted[e];
    te_pre q = load_pre(pre + (idx & 0x7fffffffu));
    if (idx & 0x80000000u) { q.x = fp_neg<Fq>(q.x); q.k = fp_neg<Fq>(q.k); }
    acc = te_madd<S>(acc, q);
  }
#pragma unroll
  for (int m = LPB / 2; m >= 1; m >>= 1) {
    te_ext o = shfl_xor_ext(acc, m);
    acc = te_add<S>(acc, o);
  }
  if (live && sub == 0) store_ext(buckets + slot, acc);
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
  p.c = c; p.nb = 1 << (c - 1);
  p.nwin = (scalar_bits + 1 + c - 1) / c;
  size_t avg = n / (size_t)p.nb;
  p.lpb = avg >= 16 ? 4 : 1;
  if (const char *e = getenv("AVRF_MSM_C")) { int v = atoi(e); if (v >= 2 && v <= 20) { p.c = v; p.nb = 1 << (v - 1); p.nwin = (scalar_bits + 1 + v - 1) / v; } }
  if (const char *e = getenv("AVRF_MSM_LPB")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) p.lpb = v; }
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
    if (buckets) HIP_CHECK(hipFree(buckets));
    HIP_CHECK(hipMalloc(&counts, nbk * 4)); HIP_CHECK(hipMalloc(&offsets, (nbk + 1) * 4));
    HIP_CHECK(hipMalloc(&buckets, nbk * sizeof(te_ext_raw)));
    cap_buckets = nbk;
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
  if (offsets) (void)hipFree(offsets); if (buckets) (void)hipFree(buckets); if (bits) (void)hipFree(bits);
  if (bits_host) (void)hipHostFree(bits_host);
  keys = sorted = counts = offsets = nullptr; buckets = bits = bits_host = nullptr; cap_n = cap_buckets = cap_bits = 0;
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
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, ws.counts, nbk, ws.offsets);
  HIP_CHECK(hipMemsetAsync(ws.counts, 0, (size_t)nbk * 4, stream));
  hipLaunchKernelGGL(k_scatter, dim3(gn.x, p.nwin), b256, 0, stream, ws.keys, (uint32_t)n, p.c, p.nwin, ws.offsets, ws.counts, ws.sorted);
  const te_pre *pre = (const te_pre *)d_pre;
  te_ext *bk = (te_ext *)ws.buckets;
  unsigned lanes = nbk * p.lpb;
  dim3 ga((lanes + 255) / 256);
  switch (p.lpb) {
    case 1: hipLaunchKernelGGL((k_accumulate<S, 1>), ga, b256, 0, stream, pre, ws.sorted, ws.offsets, nbk, bk); break;
    case 2: hipLaunchKernelGGL((k_accumulate<S, 2>), ga, b256, 0, stream, pre, ws.sorted, ws.offsets, nbk, bk); break;
    case 4: hipLaunchKernelGGL((k_accumulate<S, 4>), ga, b256, 0, stream, pre, ws.sorted, ws.offsets, nbk, bk); break;
    case 8: hipLaunchKernelGGL((k_accumulate<S, 8>), ga, b256, 0, stream, pre, ws.sorted, ws.offsets, nbk, bk); break;
    default: hipLaunchKernelGGL((k_accumulate<S, 16>), ga, b256, 0, stream, pre, ws.sorted, ws.offsets, nbk, bk); break;
  }
  const int nbits = p.nwin * p.c;
  hipLaunchKernelGGL(k_bits<S>, dim3(nbits), b256, 0, stream, bk, p.c, (te_ext *)ws.bits);
  HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.bits, (size_t)nbits * sizeof(te_ext_raw), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipGetLastError());
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
