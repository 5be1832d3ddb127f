// msm.hip -- Pippenger bucket MSM on gfx950 (CDNA4), curve-generic (curves.h): twisted Edwards (Thin / Pedersen batch
// verification) and short-Weierstrass G1 (KZG commitments of the ring SNARK).
//
// Replaces arkworks `VariableBaseMSM::msm_unchecked` (third-party ark-ec 0.6) at the
// reference call sites src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411 and inside w3f-pcs KZG.
// Any correct MSM yields the same group element; parity is on the normalised result
// (SURVEY.md A.9), so the decomposition below is designed for the GPU, not copied:
//
//   k_digits     one lane per scalar: signed c-bit digits of every window -> 16-bit keys
//                (bucket | sign<<15), window-major, coalesced
//   k_hist       workgroup (tile, window): LDS histogram of its tile of keys -> H[w][tile][b]
//   k_scan_win   workgroup per window: per-bucket prefix over tiles (in place), entry offsets, and the LANE
//                PLACEMENT: bucket b gets ceil(count_b / SEG) lanes, buckets ordered by descending entries-per-lane
//                (counting sort in LDS) so that the 64 lanes of a wave carry equal loads
//   k_scatter    workgroup (tile, window): LDS cursors seeded from the scanned histogram; every
//                key gets its slot with an LDS atomic -- no global atomics anywhere in the sort
//   k_accumulate every lane owns <= SEG consecutive entries of ONE bucket (big buckets simply get more
//                lanes), does its mixed additions on gathered precomputed points, then a wave-level
//                SEGMENTED shuffle reduction folds the lanes of a bucket; runs that cross a wave boundary
//                leave a partial
//   k_fixup      per bucket: identity for empty buckets, sum of wave partials for split ones
//   k_wsum_blk / k_wsum   weighted bucket sum sum_b b*B_b of one window (or of one fixed-base bucket set) by
//                a workgroup / a group of lanes: 2 additions per bucket + a suffix scan
//   k_rowcol, k_bits, k_bits_direct, k_horner   the bit-decomposition reduction sum_b b*B_b = sum_k 2^k T_k, kept for
//                G1 MSMs without a window table (ring_proof::index, the verifier's MSMs)
//   host         Horner over the window sums (nwin * (c doublings + 1 addition))
//
// Fixed-base mode (G1 only): the bases are a window table T[w][i] = 2^(c w) P_i, so ALL windows of a scalar vector
// share one bucket set; `batch` vectors over the same table run as `batch` bucket sets in one launch chain.
//
// Layout in HBM: points AoS (te_pre x|y|k 96 B; G1 affine x|y 64 / 96 B), gathered whole by one lane with dwordx4
// loads; keys/sorted SoA per window (coalesced); buckets AoS (te_ext 128 B; XYZZ 128 / 192 B).
#include "msm.h"
#include "curves.h"
#include "host_g1.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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

// ---------------------------------------------------------------- digits

// signed digit of window w with carry chain; digits in [-(2^(c-1)-1), 2^(c-1)];
// key = bucket (1..2^(c-1), 0 = skip) | sign << 15
// blockIdx.y = index of the scalar vector in a batch of MSMs over the same bases ("virtual windows"
// v = batch * nwin + w everywhere downstream)
// (With fixed-base window tables the nwin digit rows of one vector are simply consumed as ONE window of
// n * nwin keys: same layout, different interpretation downstream.)
__global__ void k_digits(const uint32_t *__restrict__ scalars, uint32_t n, uint32_t stride, int c, int nwin, uint16_t *__restrict__ keys) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t bat = blockIdx.y;
  keys += (size_t)bat * nwin * n;
  uint32_t s[9];
  const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * ((size_t)bat * stride + i));
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
    uint32_t key;
    if (v > nb) { key = ((1u << c) - v) | 0x8000u; carry = 1; }   // negative digit: bucket 2^c - v
    else { key = v; carry = 0; }
    keys[(size_t)w * n + i] = (uint16_t)key;
  }
}

// ---------------------------------------------------------------- counting sort (LDS only)

// H[(w * ntiles + tile) * nb + (bucket-1)] = number of keys of `tile` in that bucket
__global__ void __launch_bounds__(256)
k_hist(const uint16_t *__restrict__ keys, uint32_t n, uint32_t tile_len, int c, uint32_t *__restrict__ H) {
  extern __shared__ uint32_t lds[];
  const uint32_t nb = 1u << (c - 1), tile = blockIdx.x, w = blockIdx.y, ntiles = gridDim.x;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = 0;
  __syncthreads();
  uint32_t lo = tile * tile_len, hi = lo + tile_len; if (hi > n) hi = n;
  const uint16_t *kw = keys + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint32_t b = kw[i] & 0x7fffu;
    if (b) atomicAdd(&lds[b - 1], 1u);
  }
  __syncthreads();
  uint32_t *out = H + ((size_t)w * ntiles + tile) * nb;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) out[b] = lds[b];
}

// One workgroup (1024 lanes) per window.  In place: H[w][tile][b] becomes the exclusive prefix over
// tiles; offs/cnts per slot (= w*nb + b-1): entry offset (global, w*n based) and entry count.
// Lane placement: bucket b gets nl_b = ceil(cnt_b / seg) lanes of per_b = ceil(cnt_b / nl_b) entries each.  Buckets are
// placed in order of DESCENDING per_b (counting sort over per values in LDS), so the 64 lanes of a wave carry
// near-equal loads and the heaviest waves start first; lane_off[slot] = first lane (local to the window),
// lane_slot[w * lcap + lane] = bucket of that lane, lane_tot[w] = lanes used by the window.
__global__ void __launch_bounds__(1024)
k_scan_win(uint32_t *__restrict__ H, uint32_t n, uint32_t ntiles, int c, uint32_t seg, uint32_t lcap,
           uint32_t *__restrict__ offs, uint32_t *__restrict__ cnts, uint32_t *__restrict__ lane_off,
           uint32_t *__restrict__ lane_tot, uint32_t *__restrict__ lane_slot) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t bin[1026];                           // lanes per `per` value (0..seg), then bin cursors
  const uint32_t nb = 1u << (c - 1), w = blockIdx.x, t = threadIdx.x;
  uint32_t *Hw = H + (size_t)w * ntiles * nb;
  const uint32_t bpt = (nb + 1023) / 1024;
  uint32_t b0 = t * bpt, b1 = b0 + bpt; if (b1 > nb) b1 = nb; if (b0 > nb) b0 = nb;
  for (uint32_t i = t; i < 1026; i += 1024) bin[i] = 0;
  __syncthreads();
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; b++) {
    uint32_t run = 0;
    for (uint32_t k = 0; k < ntiles; k++) { uint32_t v = Hw[(size_t)k * nb + b]; Hw[(size_t)k * nb + b] = run; run += v; }
    cnts[(size_t)w * nb + b] = run;
    sum += run;
    if (run) { uint32_t nl = (run + seg - 1) / seg, per = (run + nl - 1) / nl; atomicAdd(&bin[per], nl); }
  }
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (t == 0) {                                            // bin start = lanes of all larger `per` values
    uint32_t run = 0;
    for (int p = (int)seg; p >= 1; p--) { uint32_t v = bin[p]; bin[p] = run; run += v; }
    lane_tot[w] = run;
  }
  __syncthreads();
  uint32_t run = part[t] - sum + w * n;
  for (uint32_t b = b0; b < b1; b++) {
    uint32_t cnt = cnts[(size_t)w * nb + b];
    offs[(size_t)w * nb + b] = run;
    run += cnt;
    uint32_t lo = 0;
    if (cnt) {
      uint32_t nl = (cnt + seg - 1) / seg, per = (cnt + nl - 1) / nl;
      lo = atomicAdd(&bin[per], nl);
      for (uint32_t r = 0; r < nl; r++) lane_slot[(size_t)w * lcap + lo + r] = b;
    }
    lane_off[(size_t)w * nb + b] = lo;
  }
}

// remap_n != 0 (fixed-base tables): key position p = dw * remap_n + i refers to table entry dw * remap_stride + i
// (or dw * remap_stride + bidx[v][i] when the vector names its own bases)
__global__ void __launch_bounds__(256)
k_scatter(const uint16_t *__restrict__ keys, uint32_t n, uint32_t tile_len, int c, const uint32_t *__restrict__ H,
          const uint32_t *__restrict__ offs, uint32_t *__restrict__ sorted, uint32_t remap_n, uint32_t remap_stride,
          const uint32_t *__restrict__ bidx) {
  extern __shared__ uint32_t lds[];
  const uint32_t nb = 1u << (c - 1), tile = blockIdx.x, w = blockIdx.y, ntiles = gridDim.x;
  const uint32_t *Hin = H + ((size_t)w * ntiles + tile) * nb;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = offs[(size_t)w * nb + b] + Hin[b];
  __syncthreads();
  uint32_t lo = tile * tile_len, hi = lo + tile_len; if (hi > n) hi = n;
  const uint16_t *kw = keys + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint32_t key = kw[i], b = key & 0x7fffu;
    if (b) {
      uint32_t pos = atomicAdd(&lds[b - 1], 1u);
      uint32_t idx = i;
      if (remap_n) {                                      // bidx: per-vector base indices (sparse MSMs over a shared table)
        uint32_t j = i % remap_n;
        idx = (i / remap_n) * remap_stride + (bidx ? bidx[(size_t)w * remap_n + j] : j);
      }
      sorted[pos] = idx | ((key & 0x8000u) << 16);
    }
  }
}

// out-of-line point ops for the (cold) reduction kernels: keeps their code size and compile time down
template <class CV> __device__ __noinline__ void cv_add_nf(typename CV::acc_t *r, const typename CV::acc_t *a, const typename CV::acc_t *b) { *r = CV::add(*a, *b); }
template <class CV> __device__ __noinline__ void cv_dbl_nf(typename CV::acc_t *r, const typename CV::acc_t *a) { *r = CV::dbl(*a); }
template <class CV> AVRF_DI typename CV::acc_t cv_add(const typename CV::acc_t &a, const typename CV::acc_t &b) {
  if (CV::INLINE_REDUCE_OPS) return CV::add(a, b);
  typename CV::acc_t r; cv_add_nf<CV>(&r, &a, &b); return r;
}
template <class CV> AVRF_DI typename CV::acc_t cv_dbl(const typename CV::acc_t &a) {
  if (CV::INLINE_REDUCE_OPS) return CV::dbl(a);
  typename CV::acc_t r; cv_dbl_nf<CV>(&r, &a); return r;
}


// ---------------------------------------------------------------- bucket accumulation (curve-generic)

// Lane t = w * lcap + lt owns a segment of the bucket `slot` = lane_slot[t] of window w, with
// lane_off[slot] <= lt < lane_off[slot] + lanes(slot) (the lanes of a bucket are consecutive).  part[2*wave + k]: partial of the run of wave
// `wave` that includes lane 0 (k = 0) or that starts later and runs past lane 63 (k = 1); complete
// runs are written straight to buckets[].  lcap is a multiple of 64 (a wave never spans two windows).
template <class CV>
__global__ void __launch_bounds__(256, CV::MIN_WAVES)
k_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
             const uint32_t *__restrict__ offs, const uint32_t *__restrict__ cnts, const uint32_t *__restrict__ lane_off,
             const uint32_t *__restrict__ lane_tot, const uint32_t *__restrict__ lane_slot, uint32_t nwin, uint32_t nb, uint32_t lcap,
             uint32_t seg, uint32_t *__restrict__ buckets, uint32_t *__restrict__ part) {
  using acc_t = typename CV::acc_t; using base_t = typename CV::base_t;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wave = t >> 6;
  const uint32_t w = t / lcap, lt = t - w * lcap;
  const bool live = w < nwin && lt < lane_tot[w];
  uint32_t slot = 0xffffffffu, l0 = 0, nl = 0;
  acc_t acc = CV::identity();
  if (live) {
    slot = w * nb + lane_slot[t];
    const uint32_t e0 = offs[slot], cnt = cnts[slot];
    l0 = lane_off[slot]; nl = (cnt + seg - 1) / seg;
    const uint32_t per = (cnt + nl - 1) / nl, r = lt - l0;
    uint32_t b = e0 + r * per, e = b + per; if (e > e0 + cnt) e = e0 + cnt;
    if (CV::PREFETCH) {
      // software-pipelined gather: the next base is in flight while the current addition runs
      if (b < e) {
        uint32_t idx = sorted[b];
        base_t q = CV::load_base(bases + (size_t)(idx & 0x7fffffffu) * CV::BASE_WORDS);
        for (uint32_t i = b; i < e; i++) {
          const uint32_t cidx = idx; const base_t cur = q;
          if (i + 1 < e) { idx = sorted[i + 1]; q = CV::load_base(bases + (size_t)(idx & 0x7fffffffu) * CV::BASE_WORDS); }
          acc = CV::madd(acc, cur, (cidx & 0x80000000u) != 0);
        }
      }
    } else {                                            // 381-bit points: registers are the scarcer resource
      for (uint32_t i = b; i < e; i++) {
        const uint32_t idx = sorted[i];
        acc = CV::madd(acc, CV::load_base(bases + (size_t)(idx & 0x7fffffffu) * CV::BASE_WORDS), (idx & 0x80000000u) != 0);
      }
    }
  }
  if (CV::SPLIT_REDUCE) {
    // 381-bit points: the general addition would set this kernel's register budget, so multi-lane buckets leave one
    // partial per lane and k_fixup adds them (dense: every bucket has 2-3 lanes there)
    if (live) CV::store_acc(nl == 1 ? buckets + (size_t)slot * CV::ACC_WORDS : part + (size_t)t * CV::ACC_WORDS, acc);
    return;
  }
  // segmented reduction by doubling: after step `off` a run head holds the sum of min(run, 2*off) lanes
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t oslot = __shfl_down(slot, off);
    bool take = live && (lane + off < 64) && (oslot == slot);
    if (!__any(take)) break;                            // no run in this wave is longer than `off`
    acc_t o = CV::shfl_down(acc, off);
    if (take) acc = CV::add(acc, o);
  }
  uint32_t pslot = __shfl_up(slot, 1);
  bool head = live && (lane == 0 || pslot != slot);
  if (head) {
    uint32_t g0 = w * lcap + l0, wbase = wave << 6;
    bool complete = (g0 >= wbase) && (g0 + nl <= wbase + 64);
    if (complete) CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, acc);
    else CV::store_acc(part + (2 * (size_t)wave + (lane == 0 ? 0 : 1)) * CV::ACC_WORDS, acc);
  }
}

template <class CV>
__global__ void __launch_bounds__(256, CV::RED_WAVES)
k_fixup(const uint32_t *__restrict__ cnts, const uint32_t *__restrict__ lane_off, uint32_t nslots, uint32_t nb,
        uint32_t lcap, uint32_t seg, const uint32_t *__restrict__ part, uint32_t *__restrict__ buckets) {
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  uint32_t cnt = cnts[slot], nl = (cnt + seg - 1) / seg;
  if (nl == 0) { if (!CV::ZERO_IS_IDENTITY) CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, CV::identity()); return; }
  uint32_t g0 = (slot / nb) * lcap + lane_off[slot];
  if (CV::SPLIT_REDUCE) {                                 // part[] holds one partial per lane of a multi-lane bucket
    if (nl == 1) return;
    typename CV::acc_t acc = CV::load_acc(part + (size_t)g0 * CV::ACC_WORDS);
    for (uint32_t r = 1; r < nl; r++) acc = CV::add(acc, CV::load_acc(part + (size_t)(g0 + r) * CV::ACC_WORDS));
    CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, acc);
    return;
  }
  uint32_t wa = g0 >> 6, wb = (g0 + nl - 1) >> 6;
  if (wa == wb) return;                                  // complete inside one wave: already written
  typename CV::acc_t acc = CV::load_acc(part + (2 * (size_t)wa + ((g0 & 63) == 0 ? 0 : 1)) * CV::ACC_WORDS);
  for (uint32_t wv = wa + 1; wv <= wb; wv++) acc = CV::add(acc, CV::load_acc(part + 2 * (size_t)wv * CV::ACC_WORDS));
  CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, acc);
}

// ---------------------------------------------------------------- bucket reduction by index bits

template <class CV> AVRF_DI typename CV::acc_t wave_sum(typename CV::acc_t acc) {
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) acc = CV::add(acc, CV::shfl_down(acc, off));
  return acc;                                             // valid in lane 0
}

// Bucket index b in [1, nb-1] (nb = 2^(c-1)) split as b = hi * 2^h + lo; B_0 = identity.
// One wave per task: tasks [0, NR) are row sums R_hi (2^h contiguous buckets), tasks [NR, NR+NC)
// are column sums C_lo (stride 2^h).  rc[w * (NR+NC) + task].
template <class CV>
__global__ void __launch_bounds__(256)
k_rowcol(const uint32_t *__restrict__ buckets, int c, int h, uint32_t total_waves, uint32_t *__restrict__ rc) {
  const uint32_t nb = 1u << (c - 1), NC = 1u << h, NR = nb >> h;
  const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (gw >= total_waves) return;
  const uint32_t tasks = NR + NC, w = gw / tasks, task = gw - w * tasks;
  const uint32_t *B = buckets + (size_t)w * nb * CV::ACC_WORDS;              // B[b-1]
  typename CV::acc_t acc = CV::identity();
  if (task < NR) {
    for (uint32_t lo = lane; lo < NC; lo += 64) { uint32_t b = task * NC + lo; if (b >= 1) acc = CV::add(acc, CV::load_acc(B + (size_t)(b - 1) * CV::ACC_WORDS)); }
  } else {
    uint32_t lo = task - NR;
    for (uint32_t hi = lane; hi < NR; hi += 64) { uint32_t b = hi * NC + lo; if (b >= 1) acc = CV::add(acc, CV::load_acc(B + (size_t)(b - 1) * CV::ACC_WORDS)); }
  }
  acc = wave_sum<CV>(acc);
  if (lane == 0) CV::store_acc(rc + (size_t)gw * CV::ACC_WORDS, acc);
}

// One wave per (window, bit k), k in [0, c): out[w*c + k] = T_k.
template <class CV>
__global__ void __launch_bounds__(256)
k_bits(const uint32_t *__restrict__ buckets, const uint32_t *__restrict__ rc, int c, int h, uint32_t total_waves, uint32_t *__restrict__ out) {
  const uint32_t nb = 1u << (c - 1), NC = 1u << h, NR = nb >> h;
  const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (gw >= total_waves) return;
  const uint32_t w = gw / c, k = gw - w * c;
  const uint32_t *RC = rc + (size_t)w * (NR + NC) * CV::ACC_WORDS;
  typename CV::acc_t acc = CV::identity();
  if ((int)k == c - 1) { if (lane == 0) acc = CV::load_acc(buckets + ((size_t)w * nb + (nb - 1)) * CV::ACC_WORDS); }   // only b = nb
  else if ((int)k < h) { for (uint32_t lo = lane; lo < NC; lo += 64) if ((lo >> k) & 1) acc = CV::add(acc, CV::load_acc(RC + (size_t)(NR + lo) * CV::ACC_WORDS)); }
  else { uint32_t kk = k - h; for (uint32_t hi = lane; hi < NR; hi += 64) if ((hi >> kk) & 1) acc = CV::add(acc, CV::load_acc(RC + (size_t)hi * CV::ACC_WORDS)); }
  acc = wave_sum<CV>(acc);
  if (lane == 0) CV::store_acc(out + (size_t)gw * CV::ACC_WORDS, acc);
}

// Small bucket counts (nb <= 256, the batched KZG-sized MSMs): one LANE per (window, bit k) adds the nb/2
// buckets whose index has bit k set -- the wave-per-task row/column kernels above would spend a whole
// wave (and a 6-step shuffle tree) on a handful of buckets.
template <class CV>
__global__ void __launch_bounds__(128)
k_bits_direct(const uint32_t *__restrict__ buckets, int c, uint32_t total, uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t nb = 1u << (c - 1), w = t / c, k = t - w * c;
  const uint32_t *B = buckets + (size_t)w * nb * CV::ACC_WORDS;
  typename CV::acc_t acc = CV::identity();
  if ((int)k == c - 1) acc = CV::load_acc(B + (size_t)(nb - 1) * CV::ACC_WORDS);
  else for (uint32_t m = 0; m < (nb >> 1); m++) {
    uint32_t b = ((m >> k) << (k + 1)) | (1u << k) | (m & ((1u << k) - 1));
    acc = CV::add(acc, CV::load_acc(B + (size_t)(b - 1) * CV::ACC_WORDS));
  }
  CV::store_acc(out + (size_t)t * CV::ACC_WORDS, acc);
}

// Window/bit Horner on the device for BATCHED MSMs: one lane per MSM walks its nbits bit sums,
// result = sum_p 2^p T_p.  (A single MSM finishes faster on the host; with hundreds of MSMs per launch
// chain the host Horner would dominate.)
template <class CV>
__global__ void __launch_bounds__(64)
k_horner(const uint32_t *__restrict__ bits, uint32_t nbits, uint32_t batch, uint32_t *__restrict__ out) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const uint32_t *T = bits + (size_t)b * nbits * CV::ACC_WORDS;
  typename CV::acc_t acc = CV::identity();
  for (int i = (int)nbits - 1; i >= 0; i--) {
    acc = CV::dbl(acc);
    acc = CV::add(acc, CV::load_acc(T + (size_t)i * CV::ACC_WORDS));
  }
  CV::store_acc(out + (size_t)b * CV::ACC_WORDS, acc);
}

// Fixed-base (table) batched MSMs have ONE bucket set per MSM: out[set] = sum_b b * B_b.  A group of 2^lps_log lanes
// (inside one wave) shares a set; lane g owns the m = nb >> lps_log consecutive buckets above g*m (running sums: 2 adds
// per bucket); then sum_g W_g + m * sum_g g * S_g by a group suffix scan and a group reduction.
template <class CV>
__global__ void __launch_bounds__(256, CV::RED_WAVES)
k_wsum(const uint32_t *__restrict__ buckets, uint32_t nb, uint32_t nsets, uint32_t lps_log, uint32_t *__restrict__ out) {
  using acc_t = typename CV::acc_t;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lps = 1u << lps_log, set = t >> lps_log, g = t & (lps - 1), m = nb >> lps_log;
  const bool live = set < nsets;
  acc_t S = CV::identity(), W = CV::identity();
  if (live) {
    const uint32_t *B = buckets + (size_t)set * nb * CV::ACC_WORDS;          // B[b-1]
    const uint32_t lo = g * m + 1;
#pragma unroll 1
    for (uint32_t b = lo + m - 1; b >= lo; b--) {                              // S = sum B_b, W = sum (b - g*m) B_b
      S = CV::add(S, CV::load_acc(B + (size_t)(b - 1) * CV::ACC_WORDS));
      W = CV::add(W, S);
    }
  }
  acc_t A = S;                                                                // suffix sums of S over the group
#pragma unroll 1
  for (uint32_t off = 1; off < lps; off <<= 1) {
    acc_t o = CV::shfl_down(A, off);
    if (g + off < lps) A = cv_add<CV>(A, o);
  }
  acc_t Z = g ? A : CV::identity();                                           // sum_g g * S_g = sum_{g >= 1} A_g
#pragma unroll 1
  for (uint32_t k = m; k > 1; k >>= 1) Z = cv_dbl<CV>(Z);
  acc_t V = cv_add<CV>(W, Z);
#pragma unroll 1
  for (uint32_t off = lps >> 1; off >= 1; off >>= 1) {
    acc_t o = CV::shfl_down(V, off);
    if (g + off < lps) V = cv_add<CV>(V, o);
  }
  if (live && g == 0) CV::store_acc(out + (size_t)set * CV::ACC_WORDS, V);
}

// Same weighted sum with a whole workgroup (wps waves) per bucket set, for launches with few sets: lane t of the
// group owns m = nb / (64 wps) consecutive buckets; each wave reduces to (V_w, Y_w) = (sum_l W_l + m l S_l, sum_l S_l)
// as above, and lane 0 finishes sum_w V_w + 64 m sum_w w Y_w over the wps pairs left in LDS.
template <class CV>
__global__ void __launch_bounds__(256, CV::RED_WAVES)
k_wsum_blk(const uint32_t *__restrict__ buckets, uint32_t nb, uint32_t *__restrict__ out) {
  using acc_t = typename CV::acc_t;
  extern __shared__ uint32_t lds[];                                            // wps x {V, Y}
  const uint32_t set = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, wps = blockDim.x >> 6;
  const uint32_t m = nb / blockDim.x;
  const uint32_t *B = buckets + (size_t)set * nb * CV::ACC_WORDS;              // B[b-1]
  acc_t S = CV::identity(), W = CV::identity();
  {
    const uint32_t lo = t * m + 1;
#pragma unroll 1
    for (uint32_t b = lo + m - 1; b >= lo; b--) {
      S = CV::add(S, CV::load_acc(B + (size_t)(b - 1) * CV::ACC_WORDS));
      W = CV::add(W, S);
    }
  }
  acc_t A = S;
#pragma unroll 1
  for (uint32_t off = 1; off < 64; off <<= 1) {
    acc_t o = CV::shfl_down(A, off);
    if (lane + off < 64) A = cv_add<CV>(A, o);
  }
  acc_t Z = lane ? A : CV::identity();
#pragma unroll 1
  for (uint32_t k = m; k > 1; k >>= 1) Z = cv_dbl<CV>(Z);
  acc_t V = cv_add<CV>(W, Z);
#pragma unroll 1
  for (uint32_t off = 32; off >= 1; off >>= 1) {
    acc_t o = CV::shfl_down(V, off);
    if (lane + off < 64) V = cv_add<CV>(V, o);
  }
  if (lane == 0) { CV::store_acc(lds + (2 * wv) * CV::ACC_WORDS, V); CV::store_acc(lds + (2 * wv + 1) * CV::ACC_WORDS, A); }
  __syncthreads();
  if (t == 0) {
    acc_t r = V, suf = CV::identity(), q = CV::identity();                     // q = sum_{w >= 1} w Y_w = sum of suffix sums
#pragma unroll 1
    for (uint32_t w = wps - 1; w >= 1; w--) {
      suf = cv_add<CV>(suf, CV::load_acc(lds + (2 * w + 1) * CV::ACC_WORDS));
      q = cv_add<CV>(q, suf);
      r = cv_add<CV>(r, CV::load_acc(lds + (2 * w) * CV::ACC_WORDS));
    }
#pragma unroll 1
    for (uint32_t k = 64 * m; k > 1; k >>= 1) q = cv_dbl<CV>(q);
    CV::store_acc(out + (size_t)set * CV::ACC_WORDS, cv_add<CV>(r, q));
  }
}

// ---------------------------------------------------------------- host engine

MsmPlan msm_plan(size_t n, int scalar_bits) {
  MsmPlan p;
  int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = lg >= 16 ? lg - 5 : lg - 3;                   // small (KZG-sized, batched) MSMs: ~8 entries per bucket
  if (c < 4) c = 4; if (c > 15) c = 15;
  p.c = c; p.lpb = 16;                                   // lpb = SEG: entries per lane
  if (const char *e = getenv("AVRF_MSM_C")) { int v = atoi(e); if (v >= 3 && v <= 15) p.c = v; }
  if (const char *e = getenv("AVRF_MSM_SEG")) { int v = atoi(e); if (v >= 1 && v <= 1024) p.lpb = v; }
  p.nb = 1 << (p.c - 1);
  p.nwin = (scalar_bits + 1 + p.c - 1) / p.c;
  return p;
}

static uint32_t tile_len_for(size_t n) { return 8192; }
static uint32_t lcap_for(size_t n, const MsmPlan &p) { return (uint32_t)(((n / (size_t)p.lpb + p.nb + 1) + 63) / 64 * 64); }

void MsmWorkspace::ensure(size_t n, const MsmPlan &p, size_t acc_bytes, size_t batch, bool lane_partials) {
  const size_t vwin = (size_t)p.nwin * batch;           // virtual windows
  size_t nbk = vwin * p.nb, nbits = vwin * p.c;
  size_t need_n = vwin * n;
  size_t ntiles = (n + tile_len_for(n) - 1) / tile_len_for(n);
  if (need_n > cap_n) {
    if (keys) HIP_CHECK(hipFree(keys));
    if (sorted) HIP_CHECK(hipFree(sorted));
    HIP_CHECK(hipMalloc(&keys, need_n * 2 + 16)); HIP_CHECK(hipMalloc(&sorted, need_n * 4));
    cap_n = need_n;
  }
  if (nbk * ntiles > cap_hist) {
    if (hist) HIP_CHECK(hipFree(hist));
    HIP_CHECK(hipMalloc(&hist, nbk * ntiles * 4));
    cap_hist = nbk * ntiles;
  }
  if (nbk > cap_slots) {
    if (cnts) HIP_CHECK(hipFree(cnts));
    if (offsets) HIP_CHECK(hipFree(offsets));
    if (lane_off) HIP_CHECK(hipFree(lane_off));
    HIP_CHECK(hipMalloc(&cnts, nbk * 4)); HIP_CHECK(hipMalloc(&offsets, nbk * 4)); HIP_CHECK(hipMalloc(&lane_off, nbk * 4));
    cap_slots = nbk;
  }
  if (vwin > cap_vwin) {
    if (lane_tot) HIP_CHECK(hipFree(lane_tot));
    HIP_CHECK(hipMalloc(&lane_tot, (vwin + 64) * 4));
    cap_vwin = vwin;
  }
  if (nbk * acc_bytes > cap_buckets) {
    if (buckets) HIP_CHECK(hipFree(buckets));
    if (rc) HIP_CHECK(hipFree(rc));
    HIP_CHECK(hipMalloc(&buckets, nbk * acc_bytes));
    HIP_CHECK(hipMalloc(&rc, nbk * acc_bytes));               // >= nwin * (NR + NC)
    cap_buckets = nbk * acc_bytes;
  }
  if (vwin * lcap_for(n, p) > cap_lanes) {
    if (lane_slot) HIP_CHECK(hipFree(lane_slot));
    cap_lanes = vwin * lcap_for(n, p);
    HIP_CHECK(hipMalloc(&lane_slot, cap_lanes * 4));
  }
  size_t need_part = lane_partials ? vwin * lcap_for(n, p) * acc_bytes : 2 * (vwin * lcap_for(n, p) / 64 + 2) * acc_bytes;
  if (need_part > cap_part) {
    if (part) HIP_CHECK(hipFree(part));
    HIP_CHECK(hipMalloc(&part, need_part));
    cap_part = need_part;
  }
  if (nbits * acc_bytes > cap_bits) {
    if (bits) HIP_CHECK(hipFree(bits));
    if (bits_host) HIP_CHECK(hipHostFree(bits_host));
    HIP_CHECK(hipMalloc(&bits, nbits * acc_bytes));
    HIP_CHECK(hipHostMalloc(&bits_host, nbits * acc_bytes));
    cap_bits = nbits * acc_bytes;
  }
}
void MsmWorkspace::release() {
  void *dev[] = {keys, sorted, hist, cnts, offsets, lane_off, lane_tot, lane_slot, buckets, rc, part, bits};
  for (void *q : dev) if (q) (void)hipFree(q);
  if (bits_host) (void)hipHostFree(bits_host);
  if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); ev0 = ev1 = nullptr;
  keys = nullptr; sorted = hist = cnts = offsets = lane_off = lane_tot = lane_slot = nullptr; cap_lanes = 0;
  buckets = rc = part = bits = bits_host = nullptr;
  cap_n = cap_slots = cap_buckets = cap_bits = cap_part = cap_hist = cap_vwin = 0;
}

// Runs the whole device pipeline for one MSM and leaves the nwin*c bit sums T_p in ws.bits_host
// (accumulator layout of CV); returns the number of bit sums.
// `batch` scalar vectors of length n over the SAME n bases (d_scalars = batch x n x 8 words): every vector
// gets its own nwin windows; returns the number of bit sums per vector (vector b's sums start at b * that).
// Fixed-base mode (table_c != 0): d_bases is a window table T[w * table_stride + i] = 2^(table_c * w) * P_i, so all
// windows of a vector share ONE bucket set: downstream it is a 1-window MSM over n * nwin (table) bases.
template <class CV>
static int msm_device(const uint32_t *d_bases, const uint32_t *d_scalars, size_t n_in, int scalar_bits, MsmWorkspace &ws, hipStream_t stream,
                      size_t batch = 1, int table_c = 0, size_t table_stride = 0, size_t scalar_stride = 0, const uint32_t *d_base_idx = nullptr) {
  MsmPlan p = msm_plan(n_in, scalar_bits);
  if (!scalar_stride) scalar_stride = n_in;                            // vector b's scalars start at b * scalar_stride
  if (!getenv("AVRF_MSM_SEG")) {
    // entries per lane: 32 once the launch has plenty of lanes, else 16.  (Measured on MI355X: longer segments save
    // cross-lane reductions but lose more to exposed gather latency -- 64+ entries per lane ran 10-50 % slower.)
    const size_t digs = table_c ? (size_t)((scalar_bits + 1 + table_c - 1) / table_c) : (size_t)p.nwin;
    p.lpb = batch * n_in * digs / 32 >= 65536 ? 32 : 16;
  }
  size_t n = n_in;
  uint32_t remap_n = 0, remap_stride = 0;
  dim3 b256(256);
  if (table_c) {
    const int dig_nwin = (scalar_bits + 1 + table_c - 1) / table_c;
    p.c = table_c; p.nb = 1 << (table_c - 1); p.nwin = 1;
    n = n_in * (size_t)dig_nwin;                                        // one window of n_in * dig_nwin keys per vector
    ws.ensure(n, p, (size_t)CV::ACC_WORDS * 4, batch, CV::SPLIT_REDUCE);
    hipLaunchKernelGGL(k_digits, dim3((unsigned)((n_in + 255) / 256), (unsigned)batch), b256, 0, stream, d_scalars, (uint32_t)n_in, (uint32_t)scalar_stride, p.c, dig_nwin, ws.keys);
    remap_n = (uint32_t)n_in; remap_stride = (uint32_t)table_stride;
  }
  const size_t acc_bytes = (size_t)CV::ACC_WORDS * 4;
  if (!table_c) ws.ensure(n, p, acc_bytes, batch, CV::SPLIT_REDUCE);
  const uint32_t vwin = (uint32_t)(p.nwin * batch);
  const uint32_t nbk = vwin * p.nb, seg = (uint32_t)p.lpb;
  const uint32_t tile_len = tile_len_for(n), ntiles = (uint32_t)((n + tile_len - 1) / tile_len);
  const uint32_t lcap = lcap_for(n, p);
  const size_t lds_bytes = (size_t)p.nb * 4;
  dim3 gn((unsigned)((n + 255) / 256));
  if (!table_c) hipLaunchKernelGGL(k_digits, dim3(gn.x, (unsigned)batch), b256, 0, stream, d_scalars, (uint32_t)n, (uint32_t)scalar_stride, p.c, p.nwin, ws.keys);
  hipLaunchKernelGGL(k_hist, dim3(ntiles, vwin), b256, lds_bytes, stream, ws.keys, (uint32_t)n, tile_len, p.c, ws.hist);
  hipLaunchKernelGGL(k_scan_win, dim3(vwin), dim3(1024), 0, stream, ws.hist, (uint32_t)n, ntiles, p.c, seg, lcap,
                     ws.offsets, ws.cnts, ws.lane_off, ws.lane_tot, ws.lane_slot);
  hipLaunchKernelGGL(k_scatter, dim3(ntiles, vwin), b256, lds_bytes, stream, ws.keys, (uint32_t)n, tile_len, p.c, ws.hist, ws.offsets, ws.sorted,
                     remap_n, remap_stride, d_base_idx);
  dim3 ga((unsigned)(((size_t)vwin * lcap + 255) / 256));
  if (!ws.ev0) { HIP_CHECK(hipEventCreate(&ws.ev0)); HIP_CHECK(hipEventCreate(&ws.ev1)); }
  if (CV::ZERO_IS_IDENTITY) HIP_CHECK(hipMemsetAsync(ws.buckets, 0, (size_t)nbk * acc_bytes, stream));   // empty buckets
  HIP_CHECK(hipEventRecord(ws.ev0, stream));
  hipLaunchKernelGGL(k_accumulate<CV>, ga, b256, 0, stream, d_bases, ws.sorted, ws.offsets, ws.cnts, ws.lane_off, ws.lane_tot, ws.lane_slot,
                     vwin, (uint32_t)p.nb, lcap, seg, ws.buckets, ws.part);
  HIP_CHECK(hipEventRecord(ws.ev1, stream));
  hipLaunchKernelGGL(k_fixup<CV>, dim3((nbk + 255) / 256), b256, 0, stream, ws.cnts, ws.lane_off, nbk, (uint32_t)p.nb, lcap, seg,
                     (const uint32_t *)ws.part, ws.buckets);
  static const bool window_sums_on = !(getenv("AVRF_TE_WINDOW_SUMS") && atoi(getenv("AVRF_TE_WINDOW_SUMS")) == 0);
  if constexpr (CV::WINDOW_SUMS) if (batch == 1 && window_sums_on) {          // one weighted sum per window; the host does nwin Horner steps of c doublings
    uint32_t wps = 1;
    while (wps < 4 && (uint32_t)p.nb >= 64 * wps * 2) wps *= 2;
    if (const char *e = getenv("AVRF_TE_WSUM_WPS")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) { wps = (uint32_t)v; while (wps > 1 && (uint32_t)p.nb < 64 * wps) wps >>= 1; } }
    if ((uint32_t)p.nb >= 64) {
      hipLaunchKernelGGL(k_wsum_blk<CV>, dim3(vwin), dim3(64 * wps), (size_t)wps * 2 * acc_bytes, stream, (const uint32_t *)ws.buckets, (uint32_t)p.nb, ws.rc);
    } else {
      uint32_t lps_log = 0; while ((2u << lps_log) <= (uint32_t)p.nb) lps_log++;
      hipLaunchKernelGGL(k_wsum<CV>, dim3((unsigned)((((size_t)vwin << lps_log) + 255) / 256)), b256, 0, stream, (const uint32_t *)ws.buckets, (uint32_t)p.nb,
                         vwin, lps_log, ws.rc);
    }
    HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.rc, (size_t)vwin * acc_bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, ws.ev0, ws.ev1));
    ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = p;
    return -(int)vwin;                                       // negative: bits_host holds window sums, not bit sums
  }
  const int h = (p.c - 1) / 2;
  const uint32_t tasks = (1u << h) + ((uint32_t)p.nb >> h);
  const int nbits = (int)vwin * p.c;
  if constexpr (CV::FIXED_TABLE) if (table_c) {   // one bucket set per MSM: weighted sum in one kernel, no bit sums
    uint32_t wps = 1;                                      // waves per bucket set when the launch has few sets
    // (a wave per set does the least work per bucket -- 2.5 adds against 4.4 with four waves -- and measured faster as
    // soon as a few hundred sets are in flight; the workgroup form is for the handful of sets of a single proof)
    const uint32_t wps_max = batch <= 64 ? 4 : 1;
    while (wps < wps_max && (uint32_t)p.nb >= 64 * wps * 2) wps *= 2;
    if (wps > 1) {
      hipLaunchKernelGGL(k_wsum_blk<CV>, dim3((unsigned)batch), dim3(64 * wps), (size_t)wps * 2 * acc_bytes, stream, (const uint32_t *)ws.buckets,
                         (uint32_t)p.nb, ws.rc);
    } else {
      uint32_t lps_log = 6;                                // lanes per bucket set: enough waves to cover the chip, <= nb
      while (lps_log > 2 && (batch << (lps_log - 1)) >= 2048 * 64) lps_log--;
      while ((1u << lps_log) > (uint32_t)p.nb) lps_log--;
      hipLaunchKernelGGL(k_wsum<CV>, dim3((unsigned)(((batch << lps_log) + 255) / 256)), b256, 0, stream, (const uint32_t *)ws.buckets, (uint32_t)p.nb,
                         (uint32_t)batch, lps_log, ws.rc);
    }
    HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.rc, batch * acc_bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, ws.ev0, ws.ev1));
    ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = p;
    return p.c;
  }
  if (p.nb <= 256 && batch >= 8) {
    hipLaunchKernelGGL(k_bits_direct<CV>, dim3(((unsigned)nbits + 127) / 128), dim3(128), 0, stream, (const uint32_t *)ws.buckets, p.c, (uint32_t)nbits, ws.bits);
  } else {
    hipLaunchKernelGGL(k_rowcol<CV>, dim3(((size_t)vwin * tasks * 64 + 255) / 256), b256, 0, stream, (const uint32_t *)ws.buckets, p.c, h,
                       vwin * tasks, ws.rc);
    hipLaunchKernelGGL(k_bits<CV>, dim3(((size_t)nbits * 64 + 255) / 256), b256, 0, stream, (const uint32_t *)ws.buckets, (const uint32_t *)ws.rc,
                       p.c, h, (uint32_t)nbits, ws.bits);
  }
  if (batch >= 8) {                                        // device Horner: only `batch` points come back
    hipLaunchKernelGGL(k_horner<CV>, dim3(((unsigned)batch + 63) / 64), dim3(64), 0, stream, (const uint32_t *)ws.bits, (uint32_t)(p.nwin * p.c),
                       (uint32_t)batch, ws.rc);
    HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.rc, batch * acc_bytes, hipMemcpyDeviceToHost, stream));
  } else {
    HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.bits, (size_t)nbits * acc_bytes, hipMemcpyDeviceToHost, stream));
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, ws.ev0, ws.ev1));
  ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = p;
  return p.nwin * p.c;
}

template <class S>
static int msm_te_impl(const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, HostExt *out) {
  using HT = HostTe<S>;
  *out = HT::identity();
  if (n == 0) return 0;
  int nbits = msm_device<TeCurve<S>>((const uint32_t *)d_pre, d_scalars, n, S::Fr::BITS, ws, stream);
  HostExt acc = HT::identity();
  const uint32_t *bh = ws.bits_host;
  if (nbits < 0) {                                        // window sums W_w: sum_w 2^(c w) W_w
    const int c = ws.last_plan.c;
    for (int w = -nbits - 1; w >= 0; w--) {
      for (int k = 0; k < c; k++) acc = HT::dbl(acc);
      acc = HT::add(acc, HT::from_raw32(bh + (size_t)w * 32));
    }
  } else for (int i = nbits - 1; i >= 0; i--) {           // bit sums T_p: sum_p 2^p T_p
    acc = HT::dbl(acc);
    acc = HT::add(acc, HT::from_raw32(bh + (size_t)i * 32));
  }
  *out = acc;
  return 0;
}

int msm_te_device(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n,
                  MsmWorkspace &ws, hipStream_t stream, HostExt *out) {
  if (suite == 0) return msm_te_impl<SuiteBandersnatch>(d_pre, d_scalars, n, ws, stream, out);
  if (suite == 1) return msm_te_impl<SuiteBabyJubJub>(d_pre, d_scalars, n, ws, stream, out);
  return -1;
}

// ---------------------------------------------------------------- G1 (KZG) MSM

// canonical affine coordinates (x || y, FQ_BYTES little-endian each; (0,0) = infinity) -> Montgomery bases
template <class C>
__global__ void k_g1_bases(const uint8_t *__restrict__ xy, uint32_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ flag) {
  using Fq = typename C::Fq; constexpr int N = Fq::N;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(xy + (size_t)i * N * 8);
  fpn<N> x = fn_load<N>(src), y = fn_load<N>(src + N);
  if (fn_ge_p<Fq>(x) || fn_ge_p<Fq>(y)) atomicOr(flag, 1u);
  fn_store<N>(out + (size_t)i * 2 * N, fn_to_mont<Fq>(x)); fn_store<N>(out + (size_t)i * 2 * N + N, fn_to_mont<Fq>(y));
}

// Fixed-base window table over `n` affine bases: table[w * n + i] = 2^(c w) * P_i, w < nwin, affine Montgomery.
template <class C>
__global__ void __launch_bounds__(64)
k_g1_table(const uint32_t *__restrict__ bases, uint32_t n, int c, int nwin, uint32_t *__restrict__ table) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename CV::base_t b = CV::load_base(bases + (size_t)i * 2 * N);
  typename CV::acc_t acc = CV::from_affine(b);
  for (int w = 0; w < nwin; w++) {
    uint32_t *o = table + ((size_t)w * n + i) * 2 * N;
    if (CV::is_identity(acc)) { fn_store<N>(o, fn_zero<N>()); fn_store<N>(o + N, fn_zero<N>()); }
    else {
      fpn<N> zzi = fn_inv<Fq>(acc.zz), zzzi = fn_inv<Fq>(acc.zzz);
      fn_store<N>(o, fn_mul<Fq>(acc.x, zzi)); fn_store<N>(o + N, fn_mul<Fq>(acc.y, zzzi));
    }
    if (w + 1 < nwin) for (int k = 0; k < c; k++) acc = CV::dbl(acc);
  }
}

void build_g1_table(int curve, const uint32_t *d_bases, size_t n, int c, int nwin, uint32_t *d_table, hipStream_t stream) {
  if (!n) return;
  dim3 g((unsigned)((n + 63) / 64)), b(64);
  if (curve == 0) hipLaunchKernelGGL(k_g1_table<G1Bls12381>, g, b, 0, stream, d_bases, (uint32_t)n, c, nwin, d_table);
  else hipLaunchKernelGGL(k_g1_table<G1Bn254>, g, b, 0, stream, d_bases, (uint32_t)n, c, nwin, d_table);
}

template <class C>
static int msm_g1_impl(const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy,
                       size_t batch, int table_c = 0, size_t table_stride = 0, size_t scalar_stride = 0, const uint32_t *d_base_idx = nullptr) {
  using HG = HostG1<C>;
  constexpr size_t OUT = 8 * C::Fq::N;                    // bytes of one affine result
  int nbits = n ? msm_device<G1Curve<C>>(d_bases, d_scalars, n, C::Fr::BITS, ws, stream, batch, table_c, table_stride, scalar_stride, d_base_idx) : 0;
  std::vector<typename HG::Pt> res(batch);
  for (size_t b = 0; b < batch; b++) {
    typename HG::Pt acc = HG::identity();
    if (batch >= 8 || table_c) acc = HG::from_raw32(ws.bits_host + b * 4 * C::Fq::N);      // Horner already done on the device
    else if (n) {
      const uint32_t *bh = ws.bits_host + b * (size_t)nbits * 4 * C::Fq::N;
      for (int i = nbits - 1; i >= 0; i--) {
        acc = HG::dbl(acc);
        acc = HG::add(acc, HG::from_raw32(bh + (size_t)i * 4 * C::Fq::N));
      }
    }
    res[b] = acc;
  }
  HG::to_affine_bytes_batch(res.data(), batch, out_xy);
  (void)OUT;
  return 0;
}

void launch_g1_bases(int curve, const uint8_t *d_xy, size_t n, uint32_t *d_out, uint32_t *d_flag, hipStream_t stream) {
  if (!n) return;
  dim3 g((unsigned)((n + 255) / 256)), b(256);
  if (curve == 0) hipLaunchKernelGGL(k_g1_bases<G1Bls12381>, g, b, 0, stream, d_xy, (uint32_t)n, d_out, d_flag);
  else hipLaunchKernelGGL(k_g1_bases<G1Bn254>, g, b, 0, stream, d_xy, (uint32_t)n, d_out, d_flag);
}

int msm_g1_device(int curve, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy,
                  size_t batch, size_t scalar_stride) {
  if (curve == 0) return msm_g1_impl<G1Bls12381>(d_bases, d_scalars, n, ws, stream, out_xy, batch, 0, 0, scalar_stride);
  if (curve == 1) return msm_g1_impl<G1Bn254>(d_bases, d_scalars, n, ws, stream, out_xy, batch, 0, 0, scalar_stride);
  return -1;
}
int msm_g1_fixed_device(int curve, const uint32_t *d_table, int table_c, size_t table_stride, const uint32_t *d_scalars, size_t n,
                        size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy, size_t batch, const uint32_t *d_base_idx) {
  if (curve == 0) return msm_g1_impl<G1Bls12381>(d_table, d_scalars, n, ws, stream, out_xy, batch, table_c, table_stride, scalar_stride, d_base_idx);
  if (curve == 1) return msm_g1_impl<G1Bn254>(d_table, d_scalars, n, ws, stream, out_xy, batch, table_c, table_stride, scalar_stride, d_base_idx);
  return -1;
}

}  // namespace avrf
