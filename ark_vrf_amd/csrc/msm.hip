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
//   k_scan_tiles lane per (window, bucket): prefix over tiles (in place), entry counts;  k_scan_offs  workgroup per window: entry offsets, window total
//   k_plan       one workgroup: entries per lane `per` = ceil(all entries / lanes the chip holds at once) and the first
//                lane of every window
//   k_scatter    workgroup (tile, window): LDS cursors seeded from the scanned histogram; every
//                key gets its slot with an LDS atomic -- no global atomics anywhere in the sort
//   k_accumulate EQUAL SHARES: every lane owns `per` consecutive entries of the sorted array, whatever buckets they fall
//                in, so all lanes of the launch do the same number of mixed additions and the launch is exactly one
//                residency round of the chip (no tail, no load sorting, no cross-lane reduction in the hot kernel).
//                At a bucket boundary the lane writes its partial sum to slot (lane + bucket) -- strictly increasing
//                along the sorted order, so the partials of one bucket are contiguous -- and starts over.
//   k_bucket_sum per bucket: identity for empty buckets, else the sum of its partials (a wave cooperates on buckets
//                that many lanes fed: skewed scalars)
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
#include "host_pool.h"
#include "curves.h"
#include "suite_dispatch.h"
#include "te_quad.h"
#include "host_g1.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace avrf {

// a failed HIP call (out of memory, lost device) unwinds to the C-ABI entry point, which returns AVRF_ERR_NO_DEVICE
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw HipFailure{e_, __FILE__, __LINE__}; } while (0)

// ---------------------------------------------------------------- conversions

// mont_in: the coordinates are ALREADY Montgomery limbs (arkworks' in-memory Fp; the zero-copy flavour of SURVEY.md 8b)
template <class S>
__global__ void __launch_bounds__(256) k_pre_from_affine(const uint8_t *__restrict__ xy, uint32_t n, te_pre *__restrict__ out,
                                  uint32_t *__restrict__ flag, int check_curve, int mont_in) {
  using Fq = typename S::Fq;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp x = fp_load_le(xy + 64 * (size_t)i), y = fp_load_le(xy + 64 * (size_t)i + 32);
  uint32_t f = 0;
  if (ge_p<Fq>(x) || ge_p<Fq>(y)) f |= 1;
  fp xm = mont_in ? x : fp_to_mont<Fq>(x), ym = mont_in ? y : fp_to_mont<Fq>(y);
  if (check_curve && !te_on_curve<S>(xm, ym)) f |= 2;
  store_pre(out + i, te_make_pre<S>(xm, ym));
  if (f) atomicOr(flag, f);
}

void launch_pre_from_affine(int suite, const uint8_t *d_xy, size_t n, te_pre_raw *d_pre, uint32_t *d_flag,
                            int check_curve, hipStream_t stream, int mont_in) {
  if (!n) return;
  dim3 g((unsigned)((n + 255) / 256)), b(256);
  with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type;
    hipLaunchKernelGGL(k_pre_from_affine<S>, g, b, 0, stream, d_xy, (uint32_t)n, (te_pre *)d_pre, d_flag, check_curve, mont_in); });
}
// scalars held as Montgomery limbs of Fr (arkworks' in-memory ScalarField) -> plain integers, in place; flag |= 4 when >= r
template <class S>
__global__ void __launch_bounds__(256) k_scalars_from_mont(uint32_t *__restrict__ sc, uint32_t n, uint32_t *__restrict__ flag) {
  using Fr = typename S::Fr;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fp k = load_fp(sc + 8 * (size_t)i);
  if (ge_p<Fr>(k)) atomicOr(flag, 4u);
  store_fp(sc + 8 * (size_t)i, fp_from_mont<Fr>(k));
}
void launch_scalars_from_mont(int suite, uint32_t *d_scalars, size_t n, uint32_t *d_flag, hipStream_t stream) {
  if (!n) return;
  with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type;
    hipLaunchKernelGGL(k_scalars_from_mont<S>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_scalars, (uint32_t)n, d_flag); });
}

// ---------------------------------------------------------------- digits

// signed digit of window w with carry chain; digits in [-(2^(c-1)-1), 2^(c-1)];
// key = bucket (1..2^(c-1), 0 = skip) | sign << 15
// blockIdx.y = index of the scalar vector in a batch of MSMs over the same bases ("virtual windows"
// v = batch * nwin + w everywhere downstream)
// (With fixed-base window tables the nwin digit rows of one vector are simply consumed as ONE window of
// n * nwin keys: same layout, different interpretation downstream.)
__global__ void __launch_bounds__(256) k_digits(const uint32_t *__restrict__ scalars, uint32_t n, uint32_t stride, int c, int nwin, uint16_t *__restrict__ keys, int mont) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t bat = blockIdx.y;
  keys += (size_t)bat * nwin * n;
  uint32_t s[9];
  const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * ((size_t)bat * stride + i));
  uint4 a = p[0], b = p[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; s[8] = 0;
  if (mont) {                                                // the vector holds Montgomery limbs (the ring prover's coefficient vectors): 1 = BLS12-381 Fr, 2 = BN254 Fr
    fp v;
#pragma unroll
    for (int k = 0; k < 8; k++) v.v[k] = s[k];
    v = mont == 1 ? fp_from_mont<FqBandersnatch>(v) : fp_from_mont<FqBabyJubJub>(v);
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = v.v[k];
  }
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

// Two steps.  k_scan_tiles, one lane per (window, bucket), adjacent lanes on adjacent buckets: in place, H[w][tile][b] becomes the
// exclusive prefix over tiles (loads issued eight tiles at a time: the chain is additions, not memory round trips) and
// cnts[slot] (slot = w*nb + b-1) the entry count.  k_scan_offs, one workgroup (1024 lanes) per window: offs[slot] = entry
// offset (global, w*n based), win_tot[w] = entries of the window.  (One fused workgroup per window took 55 us of a 1.17 ms
// batch: 4 buckets x 33 tiles of dependent, strided read-modify-writes per lane on 20 workgroups.)
__global__ void __launch_bounds__(256)
k_scan_tiles(uint32_t *__restrict__ H, uint32_t ntiles, int c, uint32_t *__restrict__ cnts) {
  const uint32_t nb = 1u << (c - 1), w = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  uint32_t *Hb = H + (size_t)w * ntiles * nb + b;
  uint32_t run = 0, k = 0;
  for (; k + 8 <= ntiles; k += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = Hb[(size_t)(k + j) * nb];
#pragma unroll
    for (int j = 0; j < 8; j++) { Hb[(size_t)(k + j) * nb] = run; run += v[j]; }
  }
  for (; k < ntiles; k++) { const uint32_t v = Hb[(size_t)k * nb]; Hb[(size_t)k * nb] = run; run += v; }
  cnts[(size_t)w * nb + b] = run;
}
__global__ void __launch_bounds__(1024)
k_scan_offs(const uint32_t *__restrict__ cnts, uint32_t n, int c, uint32_t *__restrict__ offs, uint32_t *__restrict__ win_tot) {
  __shared__ uint32_t part[1024];
  const uint32_t nb = 1u << (c - 1), w = blockIdx.x, t = threadIdx.x;
  const uint32_t bpt = (nb + 1023) / 1024;
  uint32_t b0 = t * bpt, b1 = b0 + bpt; if (b1 > nb) b1 = nb; if (b0 > nb) b0 = nb;
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; b++) sum += cnts[(size_t)w * nb + b];
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (t == 1023) win_tot[w] = part[1023];
  uint32_t run = part[t] - sum + w * n;
  for (uint32_t b = b0; b < b1; b++) { offs[(size_t)w * nb + b] = run; run += cnts[(size_t)w * nb + b]; }
}

// plan[0] = per (entries per lane), plan[1] = lanes used, plan[2] = 0 (heavy-bucket counter of k_bucket_sum); lane_base[v] = first lane of window v (lane_base[vwin] = plan[1]).
// per = max(per_min, ceil(sum of win_tot / lanes_target)): with lanes_target = what the chip holds in one residency round
// of k_accumulate, the launch is one round of equally loaded lanes.
__global__ void __launch_bounds__(1024)
k_plan(const uint32_t *__restrict__ win_tot, uint32_t vwin, uint32_t lanes_target, uint32_t per_min, uint32_t *__restrict__ lane_base,
       uint32_t *__restrict__ plan) {
  __shared__ uint64_t red[1024];
  __shared__ uint32_t sc[1024];
  __shared__ uint32_t carry_s;
  const uint32_t t = threadIdx.x;
  uint64_t s = 0;
  for (uint32_t v = t; v < vwin; v += 1024) s += win_tot[v];
  red[t] = s;
  __syncthreads();
  for (uint32_t off = 512; off >= 1; off >>= 1) { if (t < off) red[t] += red[t + off]; __syncthreads(); }
  const uint64_t total = red[0];
  // every window rounds its last lane up, so the launch uses up to total / per + vwin lanes: shares are sized for
  // lanes_target - vwin, or those few extra lanes land in a workgroup that is NOT resident in the first round -- it starts when
  // the first workgroup retires and the kernel ends a whole share later (round 5: waves lived 280 us of a 355 us launch)
  if (lanes_target > 8 * vwin) lanes_target -= vwin;
  uint64_t per64 = (total + lanes_target - 1) / lanes_target;
  if (per64 < per_min) per64 = per_min;
  const uint32_t per = (uint32_t)per64;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < vwin; base += 1024) {
    const uint32_t v = base + t;
    const uint32_t mine = v < vwin ? (win_tot[v] + per - 1) / per : 0;
    sc[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      uint32_t x = (t >= off) ? sc[t - off] : 0;
      __syncthreads();
      sc[t] += x;
      __syncthreads();
    }
    if (v < vwin) lane_base[v] = carry_s + sc[t] - mine;
    __syncthreads();
    if (t == 1023) carry_s += sc[1023];
    __syncthreads();
  }
  if (t == 0) { lane_base[vwin] = carry_s; plan[0] = per; plan[1] = carry_s; plan[2] = 0; }
}

// remap_n != 0 (fixed-base tables): key position p = dw * remap_n + i refers to table entry dw * remap_stride + i
// (or dw * remap_stride + bidx[v][i] when the vector names its own bases)
__global__ void __launch_bounds__(256)
k_scatter(const uint16_t *__restrict__ keys, uint32_t n, uint32_t tile_len, int c, const uint32_t *__restrict__ H,
          const uint32_t *__restrict__ offs, uint32_t *__restrict__ sorted, uint32_t remap_n, uint32_t remap_stride,
          const uint32_t *__restrict__ bidx, uint32_t ntiles, uint32_t vwin) {
  extern __shared__ uint32_t lds[];
  // XCD-aware placement: workgroups go round-robin over the 8 XCDs by linear id, and every XCD has its own L2.  All tiles of a
  // window run on ONE XCD (window w on XCD w mod 8), so the 4-byte scattered stores into that window's region of sorted[]
  // meet in one L2 and leave it as whole lines (with tiles of a window spread over all XCDs every L2 evicted partial lines:
  // 142 MB written for 21 MB of indices, rounds 1-2).
  const uint32_t nb = 1u << (c - 1);
  const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3, tile = k % ntiles, w = xcd + 8u * (k / ntiles);
  if (w >= vwin) return;
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
  if constexpr (CV::INLINE_REDUCE_OPS) return CV::add(a, b);
  else { typename CV::acc_t r; cv_add_nf<CV>(&r, &a, &b); return r; }
}
template <class CV> AVRF_DI typename CV::acc_t cv_dbl(const typename CV::acc_t &a) {
  if constexpr (CV::INLINE_REDUCE_OPS) return CV::dbl(a);
  else { typename CV::acc_t r; cv_dbl_nf<CV>(&r, &a); return r; }
}


// ---------------------------------------------------------------- bucket accumulation (curve-generic)

// Lane t owns the entries [e0, e0 + per) of window v (lane_base[v] <= t < lane_base[v + 1]; r = t - lane_base[v];
// e0 = v * n + r * per).  Partial sums go to part[(t + v * nb + b) * ACC_WORDS] for every bucket b the range touches.
template <class CV>
__global__ void __launch_bounds__(256, CV::MIN_WAVES)
k_accumulate(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
             const uint32_t *__restrict__ offs, const uint32_t *__restrict__ win_tot, const uint32_t *__restrict__ lane_base,
             const uint32_t *__restrict__ plan, uint32_t vwin, uint32_t nb, uint32_t n, uint32_t *__restrict__ part) {
  using AC = typename CV::accum;                              // the register form of the accumulator (curves.h)
  using acc_t = typename AC::acc_t; using base_t = typename CV::base_t;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per = plan[0];
  if (t >= plan[1]) return;
  uint32_t lo = 0, hi = vwin;                                  // largest v with lane_base[v] <= t
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (lane_base[mid] <= t) lo = mid; else hi = mid; }
  const uint32_t v = lo;
  const uint32_t wbeg = v * n, wend = wbeg + win_tot[v];
  const uint32_t e0 = wbeg + (t - lane_base[v]) * per;
  uint32_t e1 = e0 + per; if (e1 > wend) e1 = wend;
  const uint32_t *ow = offs + (size_t)v * nb;
  lo = 0; hi = nb;                                             // largest b with offs[b] <= e0: the bucket of entry e0
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (ow[mid] <= e0) lo = mid; else hi = mid; }
  uint32_t b = lo;
  uint32_t nxt = (b + 1 < nb) ? ow[b + 1] : 0xffffffffu;      // first entry of the next bucket
  const size_t slot0 = (size_t)t + (size_t)v * nb;
  acc_t acc;
  {
    // Gather pipeline through LDS: the base of entry k + 1 travels HBM -> LDS by DMA (global_load_lds_dwordx4: lane l's 16-byte
    // chunk c lands at chunk c's row + 16 l of the wave's buffer) while the addition of entry k runs, and costs NO registers until
    // it is read back at the top of its own iteration (the register form of this prefetch held 24 VGPRs across the whole
    // addition: 198 for the twisted-Edwards kernel; the 381-bit kernel had no room for it and waited out every gather).
    // Two buffers per wave; the index of entry k + 2 is in flight as well.
    extern __shared__ uint32_t acc_lds[];
    constexpr int BW = CV::BASE_WORDS, CH = BW / 4;
    uint32_t *wbuf = acc_lds + (threadIdx.x >> 6) * (2 * BW * 64);
    const uint32_t lane = threadIdx.x & 63;
    auto issue = [&](uint32_t ix, uint32_t slot) {
      const uint32_t *src = bases + (size_t)(ix & 0x7fffffffu) * BW;
#pragma unroll
      for (int c = 0; c < CH; c++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * c),
                                         (__attribute__((address_space(3))) void *)(wbuf + slot * (BW * 64) + c * 256), 16, 0, 0);
    };
    auto fetch = [&](uint32_t slot) {
      uint32_t w[BW];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(wbuf + slot * (BW * 64) + c * 256 + lane * 4);
        w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
      }
      return CV::base_from_words(w);
    };
    const uint32_t cnt = e1 - e0;
    if constexpr (!CV::PREFETCH) acc = AC::identity();
    uint32_t idx = cnt ? sorted[e0] : 0u;                       // (a lane with an empty share reads nothing and gathers nothing)
    uint32_t idx1 = cnt > 1 ? sorted[e0 + 1] : 0u;
    if (cnt) issue(idx, 0);
    for (uint32_t k = 0; k < cnt; k++) {                        // k is the same in every lane of the wave (shares start together)
      const uint32_t i = e0 + k, cidx = idx;
      // the DMA of this buffer was issued one iteration ago: the compiler's own wait insertion does not carry an LDS-DMA across
      // the loop's back edge (it put the ds_reads BEFORE its vmcnt(0)), so the wait is explicit -- and a compiler barrier.
      // lgkmcnt(0): the ds_reads of the OTHER buffer (last iteration's fetch) have returned before the DMA issued below overwrites it.
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const base_t cur = fetch(k & 1);
      idx = idx1;
      if (k + 1 < cnt) issue(idx, (k + 1) & 1);
      if (k + 2 < cnt) idx1 = sorted[i + 2];
      if constexpr (CV::PREFETCH) {                             // first entry of every lane: no addition, just the base
        if (k == 0) { acc = AC::from_base(cur, (cidx & 0x80000000u) != 0); continue; }
      }                                                          // (the 381-bit policy's madd takes the identity as it comes: one code path, fewer registers)
      if (i >= nxt) {                                          // bucket boundary inside the lane's range
        AC::store_part(part + (slot0 + b) * AC::PART_WORDS, acc);
        acc = AC::identity();
        do { b++; nxt = (b + 1 < nb) ? ow[b + 1] : 0xffffffffu; } while (i >= nxt);
      }
      acc = AC::madd(acc, cur, (cidx & 0x80000000u) != 0);
    }
  }
  AC::store_part(part + (slot0 + b) * AC::PART_WORDS, acc);
}

template <class T> struct type_tag { using type = T; };
template <class CV> AVRF_DI typename CV::acc_t wave_sum(typename CV::acc_t acc);

// Bucket `slot` = v * nb + b holds entries [offs, offs + cnt): they were accumulated by lanes lf..ll of window v, whose partials
// sit in part[lf + slot .. ll + slot].  Buckets fed by more than HEAVY lanes (the short top window of a 253-bit scalar; skewed
// scalars) are only listed here (plan[2] = count, heavy[] = slots) and summed by k_heavy_sum, one workgroup each.
constexpr uint32_t HEAVY_PARTIALS = 12;
template <class CV>
__global__ void __launch_bounds__(256, CV::RED_WAVES)
k_bucket_sum(const uint32_t *__restrict__ offs, const uint32_t *__restrict__ cnts, const uint32_t *__restrict__ lane_base,
             uint32_t *__restrict__ plan, uint32_t nslots, uint32_t nb, uint32_t n, const uint32_t *__restrict__ part,
             uint32_t *__restrict__ buckets, uint32_t *__restrict__ heavy) {
  using acc_t = typename CV::acc_t;
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const uint32_t per = plan[0];
  const uint32_t cnt = cnts[slot];
  if (!cnt) { if (!CV::ZERO_IS_IDENTITY) CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, CV::identity()); return; }
  const uint32_t v = slot / nb, rel = offs[slot] - v * n;
  const uint32_t lf = rel / per, ll = (rel + cnt - 1) / per, np = ll - lf + 1;
  if (np > HEAVY_PARTIALS) { heavy[atomicAdd(&plan[2], 1u)] = slot; return; }
  const size_t p0 = (size_t)lane_base[v] + lf + slot;
  using AC = typename CV::accum;
  acc_t acc = AC::load_part(part + p0 * AC::PART_WORDS);
#pragma unroll 1
  for (uint32_t k = 1; k < np; k++) acc = cv_add<CV>(acc, AC::load_part(part + (p0 + k) * AC::PART_WORDS));
  CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, acc);
}

template <class CV>
__global__ void __launch_bounds__(256, CV::RED_WAVES)
k_heavy_sum(const uint32_t *__restrict__ offs, const uint32_t *__restrict__ cnts, const uint32_t *__restrict__ lane_base,
            const uint32_t *__restrict__ plan, uint32_t nb, uint32_t n, const uint32_t *__restrict__ part,
            uint32_t *__restrict__ buckets, const uint32_t *__restrict__ heavy) {
  using acc_t = typename CV::acc_t;
  extern __shared__ uint32_t lds[];                                            // 4 accumulators
  const uint32_t per = plan[0], nheavy = plan[2], t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t slot = heavy[h], cnt = cnts[slot];
    const uint32_t v = slot / nb, rel = offs[slot] - v * n;
    const uint32_t lf = rel / per, ll = (rel + cnt - 1) / per, np = ll - lf + 1;
    const size_t p0 = (size_t)lane_base[v] + lf + slot;
    if constexpr (CV::QUAD) { q_heavy_sum<typename CV::suite>(part, p0, np, buckets + (size_t)slot * CV::ACC_WORDS, lds); continue; }
    acc_t a = CV::identity();
#pragma unroll 1
    for (uint32_t k = t; k < np; k += 256) a = cv_add<CV>(a, CV::accum::load_part(part + (p0 + k) * CV::accum::PART_WORDS));
    a = wave_sum<CV>(a);                                                        // valid in lane 0
    if (lane == 0) CV::store_acc(lds + wv * CV::ACC_WORDS, a);
    __syncthreads();
    if (t == 0) {
#pragma unroll 1
      for (uint32_t w = 1; w < 4; w++) a = cv_add<CV>(a, CV::load_acc(lds + w * CV::ACC_WORDS));
      CV::store_acc(buckets + (size_t)slot * CV::ACC_WORDS, a);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- bucket reduction by index bits

template <class CV> AVRF_DI typename CV::acc_t wave_sum(typename CV::acc_t acc) {
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) acc = cv_add<CV>(acc, CV::shfl_down(acc, off));
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
  if (live && g == 0) CV::store_out(out + (size_t)set * CV::OUT_WORDS, V);
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
    CV::store_out(out + (size_t)set * CV::OUT_WORDS, cv_add<CV>(r, q));
  }
}

// ---------------------------------------------------------------- host engine

// tuning knobs from the environment, read once per process
struct MsmEnv {
  int c = 0, per_min = 8, wsum_wps = 0, occ = 0, tile = 0; bool window_sums = true, tiny = true;
  size_t wsum_blk_max = 64;          // fixed-base launches of at most this many bucket sets take the workgroup-per-set weighted sum
  MsmEnv() {
    if (const char *e = getenv("AVRF_MSM_C")) { int v = atoi(e); if (v >= 3 && v <= 15) c = v; }
    if (const char *e = getenv("AVRF_MSM_OCC")) { int v = atoi(e); if (v >= 1 && v <= 8) occ = v; }   // resident k_accumulate waves per SIMD to fill
    if (const char *e = getenv("AVRF_MSM_TILE")) { int v = atoi(e); if (v >= 1024 && v <= (1 << 22)) tile = v; }       // keys per sort workgroup of a batched launch
    if (const char *e = getenv("AVRF_MSM_PER_MIN")) { int v = atoi(e); if (v >= 1 && v <= 4096) per_min = v; }
    if (const char *e = getenv("AVRF_TE_WSUM_WPS")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) wsum_wps = v; }
    if (const char *e = getenv("AVRF_TE_WINDOW_SUMS")) window_sums = atoi(e) != 0;
    if (const char *e = getenv("AVRF_G1_WSUM_BLK_MAX")) { int v = atoi(e); if (v >= 0 && v <= 65536) wsum_blk_max = (size_t)v; }
    if (const char *e = getenv("AVRF_MSM_TINY")) tiny = atoi(e) != 0;                    // (A/B hook: 0 sends small MSMs through the general chain too)
  }
};
static const MsmEnv &msm_env() { static const MsmEnv e; return e; }
// sort tile: keys per workgroup of k_hist / k_scatter.  2048 / 4096 / 8192 measured the same headline, single-context MSM time and ring
// rate (4096 moves 65 MB instead of 90 in k_scatter but doubles the histogram table); for the batched launches of the ring prover
// (>= 64 bucket sets: parallelism comes from the vectors) AVRF_MSM_TILE = 32 768 / 131 072 / one tile per vector were no faster on
// BLS12-381 and 5 % slower on BN254 (round 4: k_scatter moves 11 x its algorithmic bytes there, but it runs beside the other
// contexts' multiplier-bound k_accumulate, so its memory time is not on the critical path).
static uint32_t tile_len_for(size_t n, size_t vwin) { return vwin >= 64 && msm_env().tile ? (uint32_t)msm_env().tile : 8192u; }

MsmPlan msm_plan(size_t n, int scalar_bits) {
  MsmPlan p;
  int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = lg >= 16 ? lg - 5 : lg - 3;                   // small (KZG-sized, batched) MSMs: ~8 entries per bucket
  if (c < 4) c = 4; if (c > 15) c = 15;
  p.c = msm_env().c ? msm_env().c : c; p.lpb = 0;
  p.nb = 1 << (p.c - 1);
  p.nwin = (scalar_bits + 1 + p.c - 1) / p.c;
  return p;
}


// lanes one residency round of k_accumulate<CV> holds on the current device (CUs x resident workgroups x 256), and the
// dynamic LDS bytes that hold the kernel to that many workgroups per CU.  The kernel uses no LDS; asking for a slice of the
// CU's 160 KB is how a launch is kept BELOW what its registers would allow (AVRF_MSM_OCC, or the policy's MAX_WAVES): with
// several contexts in flight the waves of a register-light k_accumulate otherwise take every slot of a SIMD and the other
// contexts' latency-bound kernels (hashing, sort, reductions) no longer overlap with it.
struct AccShape { size_t lanes; unsigned lds; };
template <class CV> static AccShape accumulate_shape() {
  static AccShape cache[64] = {};
  static std::mutex mu;                                                       // contexts call this from their own host threads
  int dev = 0; HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[dev].lanes) {
    int cus = 0, blocks = 0, lds_cu = 0, lds_blk = 0;
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_accumulate<CV>, 256, 0));
    if (blocks < 1) blocks = 1;
    // LDS of one CU and the most one workgroup may ask for (160 KB both on gfx950; the slice below is computed from what the
    // device reports, and the cap is skipped where a slice cannot be expressed)
    if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds_cu <= 0) { (void)hipGetLastError(); lds_cu = 0; }
    if (hipDeviceGetAttribute(&lds_blk, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds_blk <= 0) { (void)hipGetLastError(); lds_blk = 64 * 1024; }
    {  // the runtime may report the 64 KB of older parts for gfx950, whose CUs have 160 KB (MI355X_MICROARCH.md)
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && !strncmp(prop.gcnArchName, "gfx950", 6)) {
        if (lds_cu < 160 * 1024) lds_cu = 160 * 1024;
        if (lds_blk < 160 * 1024) lds_blk = 160 * 1024;
      } else (void)hipGetLastError();
    }
    int cap = CV::MAX_WAVES;                                                  // a block of 256 lanes = one wave on each of a CU's 4 SIMDs
    if (msm_env().occ) cap = msm_env().occ;
    unsigned lds = 0;
    if (cap >= 1 && cap < blocks && lds_cu > 0) {
      const unsigned slice = (unsigned)(lds_cu / (cap + 1) + 1024);           // cap + 1 slices do not fit, cap slices do
      if ((size_t)slice * cap <= (size_t)lds_cu && slice <= (unsigned)lds_blk) {
        blocks = cap; lds = slice;
        if (lds > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)k_accumulate<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
    }
    {   // the gather pipeline's two buffers per wave (k_accumulate): at least that much dynamic LDS; resident workgroups follow
      const unsigned need = 4u * 2u * CV::BASE_WORDS * 64u * 4u;
      if (lds < need) lds = need;
      if (lds > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)k_accumulate<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int blk2 = 0;
      HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blk2, k_accumulate<CV>, 256, lds));
      if (blk2 >= 1 && blk2 < blocks) blocks = blk2;
    }
    cache[dev] = AccShape{(size_t)cus * blocks * 256, lds};
  }
  return cache[dev];
}
template <class CV> static size_t accumulate_lanes() { return accumulate_shape<CV>().lanes; }

// AVRF_MSM_POISON=1 (debugging): fresh workspace memory is filled with 0xA5 bytes, so that a kernel that reads what no kernel wrote fails
// every time instead of depending on what the allocator hands back
static bool msm_poison() { static const bool on = getenv("AVRF_MSM_POISON") != nullptr; return on; }
static void poison(void *p, size_t bytes, int which) {
  if (!msm_poison() || !p) return;
  if (const char *e = getenv("AVRF_MSM_POISON_SKIP")) { if (atoi(e) == which) return; }
  HIP_CHECK(hipMemset(p, 0xA5, bytes)); if (!getenv("AVRF_MSM_POISON_NOSYNC")) HIP_CHECK(hipDeviceSynchronize());
}
template <class T> static void grow(T *&p, size_t &cap, size_t need, size_t elem, int which = 0) {
  if (need <= cap) return;
  if (p) HIP_CHECK(hipFree(p));
  p = nullptr; cap = 0;
  HIP_CHECK(hipMalloc(&p, need * elem + 64));
  poison(p, need * elem + 64, which);
  cap = need;
}

void MsmWorkspace::ensure(size_t n, const MsmPlan &p, size_t acc_bytes, size_t batch, size_t lanes_max, size_t part_bytes) {
  const size_t vwin = (size_t)p.nwin * batch;           // virtual windows
  const size_t nbk = vwin * p.nb, nbits = vwin * p.c;
  const size_t need_n = vwin * n;
  const size_t ntiles = (n + tile_len_for(n, vwin) - 1) / tile_len_for(n, vwin);
  if (need_n > cap_n) {
    if (keys) HIP_CHECK(hipFree(keys));
    if (sorted) HIP_CHECK(hipFree(sorted));
    keys = nullptr; sorted = nullptr; cap_n = 0;
    HIP_CHECK(hipMalloc(&keys, need_n * 2 + 16)); HIP_CHECK(hipMalloc(&sorted, need_n * 4 + 16));
    poison(keys, need_n * 2 + 16, 1); poison(sorted, need_n * 4 + 16, 2);
    cap_n = need_n;
  }
  grow(hist, cap_hist, nbk * ntiles, 4, 3);
  if (nbk > cap_slots) {
    if (cnts) HIP_CHECK(hipFree(cnts));
    if (offsets) HIP_CHECK(hipFree(offsets));
    if (heavy) HIP_CHECK(hipFree(heavy));
    cnts = offsets = heavy = nullptr; cap_slots = 0;
    HIP_CHECK(hipMalloc(&cnts, nbk * 4)); HIP_CHECK(hipMalloc(&offsets, nbk * 4)); HIP_CHECK(hipMalloc(&heavy, nbk * 4));
    poison(cnts, nbk * 4, 4); poison(offsets, nbk * 4, 5); poison(heavy, nbk * 4, 6);
    cap_slots = nbk;
  }
  if (vwin > cap_vwin) {
    if (win_tot) HIP_CHECK(hipFree(win_tot));
    if (lane_base) HIP_CHECK(hipFree(lane_base));
    win_tot = lane_base = nullptr; cap_vwin = 0;
    HIP_CHECK(hipMalloc(&win_tot, (vwin + 64) * 4)); HIP_CHECK(hipMalloc(&lane_base, (vwin + 64) * 4));
    poison(win_tot, (vwin + 64) * 4, 7); poison(lane_base, (vwin + 64) * 4, 8);
    cap_vwin = vwin;
  }
  if (!plan_dev) { HIP_CHECK(hipMalloc(&plan_dev, 64)); HIP_CHECK(hipHostMalloc(&plan_host, 64)); plan_host[0] = plan_host[1] = 0; }
  if (nbk * acc_bytes > cap_buckets) {
    if (buckets) HIP_CHECK(hipFree(buckets));
    if (rc) HIP_CHECK(hipFree(rc));
    buckets = rc = nullptr; cap_buckets = 0;
    HIP_CHECK(hipMalloc(&buckets, nbk * acc_bytes));
    HIP_CHECK(hipMalloc(&rc, nbk * acc_bytes));               // >= nwin * (NR + NC)
    poison(buckets, nbk * acc_bytes, 9); poison(rc, nbk * acc_bytes, 10);
    cap_buckets = nbk * acc_bytes;
  }
  grow(part, cap_part, (lanes_max + nbk + 64) * (part_bytes ? part_bytes : acc_bytes), 1, 11);
  if (nbits * acc_bytes > cap_bits) {
    if (bits) HIP_CHECK(hipFree(bits));
    if (bits_host) HIP_CHECK(hipHostFree(bits_host));
    bits = bits_host = nullptr; cap_bits = 0;
    HIP_CHECK(hipMalloc(&bits, nbits * acc_bytes));
    poison(bits, nbits * acc_bytes, 12);
    HIP_CHECK(hipHostMalloc(&bits_host, nbits * acc_bytes));
    cap_bits = nbits * acc_bytes;
  }
}
void MsmWorkspace::release() {
  void *dev[] = {keys, sorted, hist, cnts, offsets, heavy, win_tot, lane_base, plan_dev, buckets, rc, part, bits};
  for (void *q : dev) if (q) (void)hipFree(q);
  if (bits_host) (void)hipHostFree(bits_host);
  if (plan_host) (void)hipHostFree(plan_host);
  if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); ev0 = ev1 = nullptr;
  keys = nullptr; sorted = hist = cnts = offsets = heavy = win_tot = lane_base = plan_dev = plan_host = nullptr;
  buckets = rc = part = bits = bits_host = nullptr;
  cap_n = cap_slots = cap_buckets = cap_bits = cap_part = cap_hist = cap_vwin = 0;
}

void MsmPending::ensure(size_t bytes) {
  if (!plan_host) HIP_CHECK(hipHostMalloc(&plan_host, 64));
  if (!ev0) { HIP_CHECK(hipEventCreate(&ev0)); HIP_CHECK(hipEventCreate(&ev1)); }
  if (bytes <= cap_bytes) return;
  if (bits_host) HIP_CHECK(hipHostFree(bits_host));
  bits_host = nullptr; cap_bytes = 0;
  HIP_CHECK(hipHostMalloc(&bits_host, bytes));
  cap_bytes = bytes;
}
void MsmPending::release() {
  if (bits_host) (void)hipHostFree(bits_host);
  if (plan_host) (void)hipHostFree(plan_host);
  if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1);
  bits_host = plan_host = nullptr; ev0 = ev1 = nullptr; cap_bytes = 0; armed = false;
}

// waits for the launch chain msm_device enqueued on `stream` and books its plan / k_accumulate timing
static void msm_wait(MsmWorkspace &ws, hipStream_t stream) {
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, ws.ev0, ws.ev1));
  ws.pending_plan.lpb = (int)ws.plan_host[0];
  ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = ws.pending_plan;
}

// Small MSMs (a BatchVerifier of up to a few hundred items, ONE independent verification run as its own equation, the verifier's
// two G1 sums of a few ring proofs): the Pippenger chain below is twelve launches of latency there (0.28 ms for five terms) and
// almost no work.  One launch instead: wave (p, v) sums the bases whose scalar of vector v has bit p set -- lane = term, a
// butterfly of ceil(log2 n) additions -- which are exactly the bit sums T_p the general path hands to the host's Horner
// (sum_p 2^p T_p; the sequential doublings run ~10 x faster on a host core than on a lone wave).
// (Up to 2 048 terms with workgroups of up to four waves -- one per SIMD, the whole register file each: a thread takes the terms
// t, t + T, .., the waves' sums meet in LDS --
// the reference's own batch-size sweep, 1 .. 256 items per BatchVerifier, benches/SUMMARY.md:43-61, lives entirely in this range.
// Above it the bit sums cost too much work -- n * bits / 2 additions against n * windows -- and the Pippenger chain takes over.)
constexpr size_t MSM_TINY_TERMS = 2048, MSM_TINY_VECTORS = 32;
template <class CV>
__global__ void __launch_bounds__(256)
k_msm_tiny_bits(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ scalars, uint32_t n, uint32_t stride, uint32_t *__restrict__ out) {
  extern __shared__ uint32_t lds[];                                            // one accumulator per wave
  const uint32_t p = blockIdx.x, v = blockIdx.y, t = threadIdx.x, T = blockDim.x, lane = t & 63, wv = t >> 6, nw = T >> 6;
  typename CV::acc_t acc = CV::identity();
  bool first = true;
#pragma unroll 1
  for (uint32_t k = t; k < n; k += T) {
    if (!((scalars[8 * ((size_t)v * stride + k) + (p >> 5)] >> (p & 31)) & 1u)) continue;
    const typename CV::base_t q = CV::load_base(bases + (size_t)k * CV::BASE_WORDS);
    acc = first ? CV::from_base(q, false) : CV::madd(acc, q, false);
    first = false;
  }
  uint32_t span = 1; while (span < n && span < 64) span <<= 1;
#pragma unroll 1
  for (uint32_t off = span >> 1; off >= 1; off >>= 1) acc = cv_add<CV>(acc, CV::shfl_down(acc, (int)off));
  if (nw > 1) {
    if (lane == 0) CV::store_acc(lds + wv * CV::ACC_WORDS, acc);
    __syncthreads();
    if (wv == 0) {
      acc = lane < nw ? CV::load_acc(lds + lane * CV::ACC_WORDS) : CV::identity();
#pragma unroll 1
      for (uint32_t off = nw >> 1; off >= 1; off >>= 1) acc = cv_add<CV>(acc, CV::shfl_down(acc, (int)off));
    }
  }
  if (t == 0) CV::store_acc(out + ((size_t)v * gridDim.x + p) * CV::ACC_WORDS, acc);
}

// Runs the whole device pipeline for one MSM and leaves the nwin*c bit sums T_p in ws.bits_host
// (accumulator layout of CV); returns the number of bit sums.
// `batch` scalar vectors of length n over the SAME n bases (d_scalars = batch x n x 8 words): every vector
// gets its own nwin windows; returns the number of bit sums per vector (vector b's sums start at b * that).
// Fixed-base mode (table_c != 0): d_bases is a window table T[w * table_stride + i] = 2^(table_c * w) * P_i, so all
// windows of a vector share ONE bucket set: downstream it is a 1-window MSM over n * nwin (table) bases.
// Throws HipFailure when a HIP call fails.
template <class CV>
static int msm_device(const uint32_t *d_bases, const uint32_t *d_scalars, size_t n_in, int scalar_bits, MsmWorkspace &ws, hipStream_t stream,
                      size_t batch = 1, int table_c = 0, size_t table_stride = 0, size_t scalar_stride = 0, const uint32_t *d_base_idx = nullptr,
                      bool defer = false, int scalars_mont = 0, MsmPending *pend = nullptr) {
  MsmPlan p = msm_plan(n_in, scalar_bits);
  if (!scalar_stride) scalar_stride = n_in;                            // vector b's scalars start at b * scalar_stride
  // (the G1 callers take eight or more vectors back as finished points -- device Horner below -- so they keep that form)
  if (n_in <= MSM_TINY_TERMS && batch <= (CV::FIXED_TABLE ? (size_t)7 : MSM_TINY_VECTORS) && !table_c && !d_base_idx && !scalars_mont && msm_env().tiny) {
    p.c = 1; p.nwin = scalar_bits; p.nb = 1; p.lpb = 0;
    const size_t acc_b = (size_t)CV::ACC_WORDS * 4, nbits = (size_t)scalar_bits;
    ws.ensure(n_in, p, acc_b, batch, 256);
    if (!ws.ev0) { HIP_CHECK(hipEventCreate(&ws.ev0)); HIP_CHECK(hipEventCreate(&ws.ev1)); }
    if (pend) { pend->ensure(batch * nbits * acc_b); pend->plan_host[0] = 0; }
    hipEvent_t e0 = pend ? pend->ev0 : ws.ev0, e1 = pend ? pend->ev1 : ws.ev1;
    HIP_CHECK(hipEventRecord(e0, stream));
    unsigned threads = 64; while (threads < n_in && threads < 256) threads <<= 1;           // a power of two of waves: the LDS butterfly halves it
    hipLaunchKernelGGL(k_msm_tiny_bits<CV>, dim3((unsigned)nbits, (unsigned)batch), dim3(threads), (threads / 64) * acc_b, stream, d_bases, d_scalars, (uint32_t)n_in,
                       (uint32_t)scalar_stride, ws.bits);
    HIP_CHECK(hipEventRecord(e1, stream));
    HIP_CHECK(hipMemcpyAsync(pend ? pend->bits_host : ws.bits_host, ws.bits, batch * nbits * acc_b, hipMemcpyDeviceToHost, stream));
    if (pend) pend->plan = p; else ws.pending_plan = p;
    if (!defer) msm_wait(ws, stream);
    return (int)nbits;
  }
  const size_t lanes_target = accumulate_lanes<CV>();
  size_t n = n_in;
  uint32_t remap_n = 0, remap_stride = 0;
  dim3 b256(256);
  int dig_nwin = p.nwin;
  if (table_c) {
    dig_nwin = (scalar_bits + 1 + table_c - 1) / table_c;
    p.c = table_c; p.nb = 1 << (table_c - 1); p.nwin = 1;
    n = n_in * (size_t)dig_nwin;                                        // one window of n_in * dig_nwin keys per vector
    remap_n = (uint32_t)n_in; remap_stride = (uint32_t)table_stride;
  }
  // fixed-base mode: the reduction kernels run over CV::red (G1: general additions on the unsaturated limbs, buckets in raw limbs)
  using RV = typename CV::red;
  const bool red = CV::FIXED_TABLE && table_c != 0;
  const size_t acc_bytes = (size_t)(red ? (int)RV::ACC_WORDS : (int)CV::ACC_WORDS) * 4;
  const uint32_t vwin = (uint32_t)(p.nwin * batch);
  const size_t lanes_max = lanes_target + vwin + 256;                   // sum_v ceil(tot_v / per) <= total / per + vwin
  if ((size_t)vwin * n >= 0xffff0000ull) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};   // 32-bit entry offsets
  ws.ensure(n, p, acc_bytes, batch, lanes_max, (size_t)CV::accum::PART_WORDS * 4);
  const uint32_t nbk = vwin * p.nb;
  const uint32_t tile_len = tile_len_for(n, vwin), ntiles = (uint32_t)((n + tile_len - 1) / tile_len);
  const size_t lds_bytes = (size_t)p.nb * 4;
  hipLaunchKernelGGL(k_digits, dim3((unsigned)((n_in + 255) / 256), (unsigned)batch), b256, 0, stream, d_scalars, (uint32_t)n_in, (uint32_t)scalar_stride,
                     p.c, dig_nwin, ws.keys, scalars_mont);
  hipLaunchKernelGGL(k_hist, dim3(ntiles, vwin), b256, lds_bytes, stream, ws.keys, (uint32_t)n, tile_len, p.c, ws.hist);
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)((p.nb + 255) / 256), vwin), b256, 0, stream, ws.hist, ntiles, p.c, ws.cnts);
  hipLaunchKernelGGL(k_scan_offs, dim3(vwin), dim3(1024), 0, stream, (const uint32_t *)ws.cnts, (uint32_t)n, p.c, ws.offsets, ws.win_tot);
  hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, stream, (const uint32_t *)ws.win_tot, vwin, (uint32_t)lanes_target, (uint32_t)msm_env().per_min,
                     ws.lane_base, ws.plan_dev);
  hipLaunchKernelGGL(k_scatter, dim3(8u * ((vwin + 7u) / 8u) * ntiles), b256, lds_bytes, stream, ws.keys, (uint32_t)n, tile_len, p.c, ws.hist, ws.offsets, ws.sorted,
                     remap_n, remap_stride, d_base_idx, ntiles, vwin);
  if (!ws.ev0) { HIP_CHECK(hipEventCreate(&ws.ev0)); HIP_CHECK(hipEventCreate(&ws.ev1)); }
  if (pend) pend->ensure((size_t)vwin * 3 * acc_bytes);
  hipEvent_t ev0 = pend ? pend->ev0 : ws.ev0, ev1 = pend ? pend->ev1 : ws.ev1;
  if (CV::ZERO_IS_IDENTITY) HIP_CHECK(hipMemsetAsync(ws.buckets, 0, (size_t)nbk * acc_bytes, stream));   // empty buckets
  HIP_CHECK(hipEventRecord(ev0, stream));
  hipLaunchKernelGGL(k_accumulate<CV>, dim3((unsigned)((lanes_max + 255) / 256)), b256, accumulate_shape<CV>().lds, stream, d_bases, (const uint32_t *)ws.sorted,
                     (const uint32_t *)ws.offsets, (const uint32_t *)ws.win_tot, (const uint32_t *)ws.lane_base, (const uint32_t *)ws.plan_dev, vwin,
                     (uint32_t)p.nb, (uint32_t)n, ws.part);
  HIP_CHECK(hipEventRecord(ev1, stream));
  auto bucket_sums = [&](auto tag) {
    using BV = typename decltype(tag)::type;
    hipLaunchKernelGGL(k_bucket_sum<BV>, dim3((nbk + 255) / 256), b256, 0, stream, (const uint32_t *)ws.offsets, (const uint32_t *)ws.cnts,
                       (const uint32_t *)ws.lane_base, ws.plan_dev, nbk, (uint32_t)p.nb, (uint32_t)n, (const uint32_t *)ws.part, ws.buckets, ws.heavy);
    hipLaunchKernelGGL(k_heavy_sum<BV>, dim3(nbk < 1024 ? nbk : 1024), b256, 4 * acc_bytes, stream, (const uint32_t *)ws.offsets, (const uint32_t *)ws.cnts,
                       (const uint32_t *)ws.lane_base, (const uint32_t *)ws.plan_dev, (uint32_t)p.nb, (uint32_t)n, (const uint32_t *)ws.part, ws.buckets,
                       (const uint32_t *)ws.heavy);
  };
  if (red) bucket_sums(type_tag<RV>{}); else bucket_sums(type_tag<CV>{});
  HIP_CHECK(hipMemcpyAsync(pend ? pend->plan_host : ws.plan_host, ws.plan_dev, 8, hipMemcpyDeviceToHost, stream));
  // defer: the caller collects the results later (msm_wait): everything up to the copies back is enqueued, nothing is waited for
  if (pend) pend->plan = p; else ws.pending_plan = p;
  auto finish = [&]() { if (!defer) msm_wait(ws, stream); };
  if constexpr (CV::WINDOW_SUMS) if (batch == 1 && msm_env().window_sums) {
    // weighted bucket sum of every window with the four-lanes-per-point kernels (te_quad.h): three points per window come
    // back; the host folds them into its window Horner (scales 1, 2^4, 16 m), so the device does no doublings at all
    const uint32_t nb = (uint32_t)p.nb;
    const uint32_t wpw = nb >= 256 ? 16u : 1u, m = nb >= 256 ? nb / 256 : (nb >= 16 ? nb / 16 : 1u);
    const uint32_t nwaves = vwin * wpw;
    hipLaunchKernelGGL(k_wsum_q1<typename CV::suite>, dim3((nwaves + 3) / 4), b256, 0, stream, (const uint32_t *)ws.buckets, nb, m, wpw, nwaves, ws.rc);
    hipLaunchKernelGGL(k_wsum_q2<typename CV::suite>, dim3(vwin), dim3(64), 0, stream, (const uint32_t *)ws.rc, wpw, ws.bits);
    HIP_CHECK(hipMemcpyAsync(pend ? pend->bits_host : ws.bits_host, ws.bits, (size_t)vwin * 3 * acc_bytes, hipMemcpyDeviceToHost, stream));
    finish();
    int lg = 4; while ((1u << lg) < 16 * m) lg++;
    if (pend) pend->wsum_lg = lg; else ws.wsum_lg = lg;
    return -(int)vwin;                                       // negative: bits_host holds window triples, not bit sums
  }
  if (pend) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};   // external results exist for the window-sum form only
  const int h = (p.c - 1) / 2;
  const uint32_t tasks = (1u << h) + ((uint32_t)p.nb >> h);
  const int nbits = (int)vwin * p.c;
  if constexpr (CV::FIXED_TABLE) if (table_c) {   // one bucket set per MSM: weighted sum in one kernel, no bit sums
    uint32_t wps = 1;                                      // waves per bucket set when the launch has few sets
    // (a wave per set does the least work per bucket -- 2.5 adds against 4.4 with four waves -- and measured faster as
    // soon as a few hundred sets are in flight; the workgroup form is for the handful of sets of a single proof)
    // (round 3 re-check with 512-set launches, which fill only half the SIMDs with one wave per set: two waves per set 12.4 k proofs/s,
    // four 11.9 k against 12.6-12.7 k -- with several contexts in flight the idle SIMDs are not idle, total work decides)
    const uint32_t wps_max = batch <= msm_env().wsum_blk_max ? 4 : 1;
    while (wps < wps_max && (uint32_t)p.nb >= 64 * wps * 2) wps *= 2;
    if (wps > 1) {
      hipLaunchKernelGGL(k_wsum_blk<RV>, dim3((unsigned)batch), dim3(64 * wps), (size_t)wps * 2 * acc_bytes, stream, (const uint32_t *)ws.buckets,
                         (uint32_t)p.nb, ws.rc);
    } else {
      uint32_t lps_log = 6;                                // lanes per bucket set: enough waves to cover the chip, <= nb
      while (lps_log > 2 && (batch << (lps_log - 1)) >= 2048 * 64) lps_log--;
      while ((1u << lps_log) > (uint32_t)p.nb) lps_log--;
      hipLaunchKernelGGL(k_wsum<RV>, dim3((unsigned)(((batch << lps_log) + 255) / 256)), b256, 0, stream, (const uint32_t *)ws.buckets, (uint32_t)p.nb,
                         (uint32_t)batch, lps_log, ws.rc);
    }
    HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.rc, batch * (size_t)RV::OUT_WORDS * 4, hipMemcpyDeviceToHost, stream));
    finish();
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
  finish();
  return p.nwin * p.c;
}

template <class S>
static int msm_te_enqueue_impl(const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, MsmPending *pend) {
  if (pend) {
    pend->n = n; pend->ret = 0; pend->armed = true;
    if (n) pend->ret = msm_device<TeCurve<S>>((const uint32_t *)d_pre, d_scalars, n, S::Fr::BITS, ws, stream, 1, 0, 0, 0, nullptr, /*defer=*/true, 0, pend);
    return 0;
  }
  ws.pending_n = n; ws.pending_ret = 0; ws.pending_armed = true;
  if (n) ws.pending_ret = msm_device<TeCurve<S>>((const uint32_t *)d_pre, d_scalars, n, S::Fr::BITS, ws, stream, 1, 0, 0, 0, nullptr, /*defer=*/true);
  return 0;
}
template <class S>
static int msm_te_finish_impl(MsmWorkspace &ws, hipStream_t stream, HostExt *out, MsmPending *pend) {
  using HT = HostTe<S>;
  *out = HT::identity();
  int nbits, c, lg; const uint32_t *bh;
  if (pend) {
    if (!pend->armed) return -1;
    pend->armed = false;
    if (pend->n == 0) return 0;
    HIP_CHECK(hipEventElapsedTime(&ws.accum_ms_last, pend->ev0, pend->ev1));   // (the caller saw the chain complete)
    pend->plan.lpb = (int)pend->plan_host[0];
    ws.accum_ms_total += ws.accum_ms_last; ws.accum_launches++; ws.last_plan = pend->plan;
    nbits = pend->ret; c = pend->plan.c; lg = pend->wsum_lg; bh = pend->bits_host;
  } else {
    if (!ws.pending_armed) return -1;                     // nothing was enqueued (or another call consumed the chain): an error, never the identity
    ws.pending_armed = false;
    if (ws.pending_n == 0) return 0;
    ws.pending_n = 0;
    msm_wait(ws, stream);
    nbits = ws.pending_ret; c = ws.last_plan.c; lg = ws.wsum_lg; bh = ws.bits_host;
  }
  HostExt acc = HT::identity();
  if (nbits < 0) {                                        // window triples: W_w = P1 + 2^4 P2 + 2^lg P3; sum_w 2^(c w) W_w
    for (int w = -nbits - 1; w >= 0; w--) {
      for (int k = 0; k < c - lg; k++) acc = HT::dbl(acc);
      acc = HT::add(acc, HT::from_raw32(bh + ((size_t)w * 3 + 2) * 32));
      for (int k = 0; k < lg - 4; k++) acc = HT::dbl(acc);
      acc = HT::add(acc, HT::from_raw32(bh + ((size_t)w * 3 + 1) * 32));
      for (int k = 0; k < 4; k++) acc = HT::dbl(acc);
      acc = HT::add(acc, HT::from_raw32(bh + (size_t)w * 3 * 32));
    }
  } else for (int i = nbits - 1; i >= 0; i--) {           // bit sums T_p: sum_p 2^p T_p
    acc = HT::dbl(acc);
    acc = HT::add(acc, HT::from_raw32(bh + (size_t)i * 32));
  }
  *out = acc;
  return 0;
}

int msm_te_enqueue(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, MsmPending *pend) {
  if (suite < 0 || suite >= AVRF_N_SUITES) return -1;
  return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return msm_te_enqueue_impl<S>(d_pre, d_scalars, n, ws, stream, pend); });
}
int msm_te_finish(int suite, MsmWorkspace &ws, hipStream_t stream, HostExt *out, MsmPending *pend) {
  if (suite < 0 || suite >= AVRF_N_SUITES) return -1;
  return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return msm_te_finish_impl<S>(ws, stream, out, pend); });
}
template <class S>
static int msm_te_small_vectors_impl(const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, size_t nv, MsmWorkspace &ws, hipStream_t stream, HostExt *out) {
  using HT = HostTe<S>;
  if (!n || n > MSM_TINY_TERMS || !nv || nv > MSM_TINY_VECTORS || !msm_env().tiny) return -1;
  const int nbits = msm_device<TeCurve<S>>((const uint32_t *)d_pre, d_scalars, n, S::Fr::BITS, ws, stream, nv, 0, 0, n);
  ws.pending_armed = false;
  const uint32_t *bh = ws.bits_host;
  parallel_for(nv, [&](size_t v) {                          // the vectors' Horners side by side (sum_p 2^p T_p)
    HostExt acc = HT::identity();
    for (int i = nbits - 1; i >= 0; i--) { acc = HT::dbl(acc); acc = HT::add(acc, HT::from_raw32(bh + ((size_t)v * nbits + i) * 32)); }
    out[v] = acc;
  });
  return 0;
}
int msm_te_small_vectors(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, size_t n_vectors, MsmWorkspace &ws, hipStream_t stream,
                         HostExt *out) {
  if (suite < 0 || suite >= AVRF_N_SUITES) return -1;
  return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return msm_te_small_vectors_impl<S>(d_pre, d_scalars, n, n_vectors, ws, stream, out); });
}
bool msm_te_pending_supported(int suite) {
  if (suite < 0 || suite >= AVRF_N_SUITES || !msm_env().window_sums) return false;
  return with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return (bool)TeCurve<S>::WINDOW_SUMS; });
}
int msm_te_device(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n,
                  MsmWorkspace &ws, hipStream_t stream, HostExt *out) {
  if (int e = msm_te_enqueue(suite, d_pre, d_scalars, n, ws, stream)) return e;
  return msm_te_finish(suite, ws, stream, out);
}

// ---------------------------------------------------------------- G1 (KZG) MSM

// canonical affine coordinates (x || y, FQ_BYTES little-endian each; (0,0) = infinity) -> Montgomery bases
template <class C>
__global__ void __launch_bounds__(256) k_g1_bases(const uint8_t *__restrict__ xy, uint32_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ flag) {
  using Fq = typename C::Fq; constexpr int N = Fq::N;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(xy + (size_t)i * N * 8);
  fpn<N> x = fn_load<N>(src), y = fn_load<N>(src + N);
  if (fn_ge_p<Fq>(x) || fn_ge_p<Fq>(y)) atomicOr(flag, 1u);
  fn_store<N>(out + (size_t)i * 2 * N, fn_to_mont<Fq>(x)); fn_store<N>(out + (size_t)i * 2 * N + N, fn_to_mont<Fq>(y));
}

// CanonicalDeserialize of compressed G1 points (ark-serialize; the zcash big-endian form for BLS12-381, little-endian with the
// flags in the last byte for BN254 -- SURVEY.md A.1), one lane per point: x from the bytes, y = (x^3 + b)^((p + 1) / 4)
// (p = 3 mod 4 for both fields), sign by the "lexicographically largest" flag.  out_xy = x || y canonical little-endian
// ((0, 0) for the point at infinity), ok[i] = 0 undecodable / not on the curve, 1 a point, 2 infinity.  The ring verifiers
// need the y coordinates on the HOST too (their transcript absorbs uncompressed points), so the result goes back; a 381-bit
// square root is ~570 field multiplications, 60 us on a host core.
template <class C>
__global__ void __launch_bounds__(64) k_g1_decompress(const uint8_t *__restrict__ comp, uint32_t n, uint8_t *__restrict__ out_xy, uint8_t *__restrict__ ok) {
  using Fq = typename C::Fq; constexpr int N = Fq::N, FQB = 4 * N;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t *b = comp + (size_t)i * FQB;
  uint8_t le[FQB]; bool big, inf;
  if (FQB == 48) {
    inf = b[0] & 0x40; big = b[0] & 0x20;
    if (!(b[0] & 0x80)) { ok[i] = 0; return; }
    for (int k = 0; k < FQB; k++) le[k] = b[FQB - 1 - k];
    le[FQB - 1] &= 0x1f;
  } else {
    inf = b[FQB - 1] & 0x40; big = b[FQB - 1] & 0x80;
    for (int k = 0; k < FQB; k++) le[k] = b[k];
    le[FQB - 1] &= 0x3f;
  }
  fpn<N> x;
  for (int k = 0; k < N; k++) x.v[k] = (uint32_t)le[4 * k] | ((uint32_t)le[4 * k + 1] << 8) | ((uint32_t)le[4 * k + 2] << 16) | ((uint32_t)le[4 * k + 3] << 24);
  uint32_t *o = reinterpret_cast<uint32_t *>(out_xy + (size_t)i * 2 * FQB);
  if (inf) {                                                          // canonical encoding only: no sort flag, every other bit zero
    const bool good = !big && fn_is_zero(x);
    fn_store<N>(o, fn_zero<N>()); fn_store<N>(o + N, fn_zero<N>());
    ok[i] = good ? 2 : 0; return;
  }
  if (fn_ge_p<Fq>(x)) { ok[i] = 0; return; }
  const fpn<N> xm = fn_to_mont<Fq>(x);
  const fpn<N> rhs = fn_add<Fq>(fn_mul<Fq>(fn_sqr<Fq>(xm), xm), fn_const<Fq>(C::B));
  uint32_t e[N];                                                      // (p + 1) / 4 = ((p - 1) / 2 + 1) / 2
  { uint64_t c = 1; for (int k = 0; k < N; k++) { c += Fq::HALF[k]; e[k] = (uint32_t)c; c >>= 32; } }
  for (int k = 0; k < N; k++) e[k] = (e[k] >> 1) | (k + 1 < N ? e[k + 1] << 31 : 0u);
  fpn<N> y = fn_one<Fq>();
#pragma unroll 1
  for (int k = 32 * N - 1; k >= 0; k--) {
    y = fn_sqr<Fq>(y);
    if ((e[k >> 5] >> (k & 31)) & 1) y = fn_mul<Fq>(y, rhs);
  }
  if (!fn_eq(fn_sqr<Fq>(y), rhs)) { ok[i] = 0; return; }             // not on the curve
  fpn<N> yp = fn_from_mont<Fq>(y);
  bool is_big = false;                                                // yp > (p - 1) / 2 ?
  for (int k = N - 1; k >= 0; k--) if (yp.v[k] != Fq::HALF[k]) { is_big = yp.v[k] > Fq::HALF[k]; break; }
  if (is_big != big) yp = fn_from_mont<Fq>(fn_neg<Fq>(y));
  fn_store<N>(o, x); fn_store<N>(o + N, yp);
  ok[i] = 1;
}

// Fixed-base window table over `n` affine bases: table[w * n + i] = 2^(c w) * P_i, w < nwin, affine Montgomery.
// One lane per base walks its nwin rows twice: forwards it leaves X, Y in the table slot and ZZ, ZZZ and the running product of
// the ZZ ZZZ before the row in `tmp` (3 N words per entry); ONE inversion of the lane's total; backwards it peels 1 / (ZZ ZZZ)
// off row by row (Montgomery's trick) and normalises.  (Two inversions per entry -- 1 140 of the ~1 230 multiplications an
// entry cost -- made the two tables of a ring setup 84 ms of a 2 048-proof profile.)
template <class C>
__global__ void __launch_bounds__(64)
k_g1_table(const uint32_t *__restrict__ bases, uint32_t n, int c, int nwin, uint32_t *__restrict__ table, uint32_t *__restrict__ tmp) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename CV::base_t b = CV::load_base(bases + (size_t)i * 2 * N);
  typename CV::acc_t acc = CV::from_affine(b);
  fpn<N> run = fn_one<Fq>();
#pragma unroll 1
  for (int w = 0; w < nwin; w++) {
    const size_t e = (size_t)w * n + i;
    fn_store<N>(table + e * 2 * N, acc.x); fn_store<N>(table + e * 2 * N + N, acc.y);
    fn_store<N>(tmp + e * 3 * N, acc.zz); fn_store<N>(tmp + e * 3 * N + N, acc.zzz); fn_store<N>(tmp + e * 3 * N + 2 * N, run);
    if (!CV::is_identity(acc)) run = fn_mul<Fq>(run, fn_mul<Fq>(acc.zz, acc.zzz));
    if (w + 1 < nwin) for (int k = 0; k < c; k++) acc = CV::dbl(acc);
  }
  fpn<N> inv = fn_inv<Fq>(run);
#pragma unroll 1
  for (int w = nwin - 1; w >= 0; w--) {
    const size_t e = (size_t)w * n + i;
    uint32_t *o = table + e * 2 * N;
    const fpn<N> zz = fn_load<N>(tmp + e * 3 * N), zzz = fn_load<N>(tmp + e * 3 * N + N);
    if (fn_is_zero(zz)) { fn_store<N>(o, fn_zero<N>()); fn_store<N>(o + N, fn_zero<N>()); continue; }
    const fpn<N> dinv = fn_mul<Fq>(inv, fn_load<N>(tmp + e * 3 * N + 2 * N));     // 1 / (ZZ ZZZ) of this row
    inv = fn_mul<Fq>(inv, fn_mul<Fq>(zz, zzz));
    fn_store<N>(o, fn_mul<Fq>(fn_load<N>(o), fn_mul<Fq>(dinv, zzz)));             // X / ZZ
    fn_store<N>(o + N, fn_mul<Fq>(fn_load<N>(o + N), fn_mul<Fq>(dinv, zz)));      // Y / ZZZ
  }
}

void build_g1_table(int curve, const uint32_t *d_bases, size_t n, int c, int nwin, uint32_t *d_table, hipStream_t stream) {
  if (!n) return;
  const size_t fqn = curve == 0 ? G1Bls12381::Fq::N : G1Bn254::Fq::N;
  // Plain scratch, the stream waited for, then freed: setup-time work.  (Until round 6 the scratch came from the stream-ordered pool --
  // hipMallocAsync / hipFreeAsync -- so that the build stayed asynchronous.  Next to the synchronous hipMalloc / hipFree of the MSM
  // workspaces that was a race: after two setups on two streams the first batched commitment of avrf_ring_srs_generate returned the
  // point at infinity for its first vector, every time, and not with HIP_LAUNCH_BLOCKING=1 or a device synchronisation in between
  // (tests/test_gpu_ring.py::test_srs_generate caught it once an unrelated null-stream hipMemset, which had been hiding it, went away).)
  uint32_t *tmp = nullptr;
  HIP_CHECK(hipMalloc((void **)&tmp, (size_t)nwin * n * 3 * fqn * 4));
  dim3 g((unsigned)((n + 63) / 64)), b(64);
  if (curve == 0) hipLaunchKernelGGL(k_g1_table<G1Bls12381>, g, b, 0, stream, d_bases, (uint32_t)n, c, nwin, d_table, tmp);
  else hipLaunchKernelGGL(k_g1_table<G1Bn254>, g, b, 0, stream, d_bases, (uint32_t)n, c, nwin, d_table, tmp);
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipFree(tmp));
}

template <class C>
static int msm_g1_impl(const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy,
                       size_t batch, int table_c = 0, size_t table_stride = 0, size_t scalar_stride = 0, const uint32_t *d_base_idx = nullptr,
                       int scalars_mont = 0) {
  using HG = HostG1<C>;
  constexpr size_t OUT = 8 * C::Fq::N;                    // bytes of one affine result
  int nbits = n ? msm_device<G1Curve<C>>(d_bases, d_scalars, n, C::Fr::BITS, ws, stream, batch, table_c, table_stride, scalar_stride, d_base_idx, false, scalars_mont) : 0;
  std::vector<typename HG::Pt> res(batch);
  // (a few vectors without the device Horner: their bit-sum Horners -- nbits doublings and additions each -- run side by side)
  auto fold = [&](size_t b) {
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
  };
  if (batch >= 2 && batch < 8 && !table_c) parallel_for(batch, fold);
  else for (size_t b = 0; b < batch; b++) fold(b);
  HG::to_affine_bytes_batch(res.data(), batch, out_xy);
  (void)OUT;
  return 0;
}

// Prime-order-subgroup test of BLS12-381 G1 points (cofactor ~2^126, so curve membership is not enough; the reference gets
// it from ark-serialize's Validate::Yes when it deserialises a RingProof / RingCommitment, src/ring.rs:97-150):
//   P in G1  <=>  phi(P) = [-z^2] P,  phi(x, y) = (beta x, y),  z = 0xd201000000010000  (Scott 2021, "A note on group
// membership tests for G1, G2 and GT on BLS pairing-friendly curves"): two 64-bit double-and-add ladders instead of a
// 255-bit one.  One lane per point; (0, 0) = infinity passes.  BN254 G1 has cofactor 1: nothing to test.
__device__ static const uint32_t BLS12_381_BETA_MONT[12] = {0x798a64e8, 0x30f1361b, 0x7ece5a2a, 0xf3b8ddab, 0xc61577f7, 0x16a8ca3a,
                                                            0x74fd029b, 0xc26a2ff8, 0x60701c6e, 0x3636b766, 0x241b6160, 0x051ba4ab};
__global__ void __launch_bounds__(64)
k_g1_subgroup_bls(const uint32_t *__restrict__ bases, uint32_t n, uint32_t *__restrict__ flag, int32_t *__restrict__ rec_status, uint32_t ppr) {
  using CV = G1Curve<G1Bls12381>; using Fq = G1Bls12381::Fq; constexpr int N = Fq::N;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CV::base_t P = CV::load_base(bases + (size_t)i * 2 * N);
  if (fn_is_zero(P.x) && fn_is_zero(P.y)) return;
  const uint64_t z = 0xd201000000010000ull;
  CV::acc_t a = CV::from_affine(P);
#pragma unroll 1
  for (int b = 62; b >= 0; b--) { a = CV::dbl(a); if ((z >> b) & 1) a = CV::madd(a, P, false); }
  const CV::acc_t q1 = a;                                                   // [z] P
#pragma unroll 1
  for (int b = 62; b >= 0; b--) { a = CV::dbl(a); if ((z >> b) & 1) a = CV::add(a, q1); }      // [z^2] P
  fpn<N> beta; for (int k = 0; k < N; k++) beta.v[k] = BLS12_381_BETA_MONT[k];
  // [z^2] P == (beta x, -y)  in XYZZ:  X = beta x ZZ,  Y = -y ZZZ,  ZZ != 0
  bool ok = !fn_is_zero(a.zz);
  ok = ok && fn_eq(a.x, fn_mul<Fq>(fn_mul<Fq>(beta, P.x), a.zz));
  ok = ok && fn_eq(a.y, fn_neg<Fq>(fn_mul<Fq>(P.y, a.zzz)));
  if (!ok) { atomicOr(flag, 2u); if (rec_status) rec_status[i / ppr] = 2; }
}
void launch_g1_subgroup_check(int curve, const uint32_t *d_bases, size_t n, uint32_t *d_flag, hipStream_t stream, int32_t *d_rec_status, uint32_t ppr) {
  if (!n || curve != 0) return;
  hipLaunchKernelGGL(k_g1_subgroup_bls, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_bases, (uint32_t)n, d_flag, d_rec_status, ppr ? ppr : 1u);
}

// Small linear combinations, one 16-lane group per item: lane t < tpi computes scalar_t * P_t, lanes [0, split) are summed into
// point 0, lanes [split, tpi) into point 1; both leave as Montgomery affine x | y ((0, 0) = infinity) at
// out[(2 item + q) * 2N] -- the two G1 arguments of an independent KZG pairing check (RingVerifier::verify, src/ring.rs:242),
// whose bases differ per item so that no bucket method applies.
// GLV (BLS12-381): phi(x, y) = (beta x, y) multiplies by -z^2 on G1 (the relation k_g1_subgroup_bls tests), so with
// k = k1 + k2 z^2 (k1 = k mod z^2, k2 = k div z^2, both < 2^128; the caller splits, scalars arrive as k1 | k2 << 128)
// k P = k1 P + k2 (-phi(P)): one chain of 128 doublings over the joint 2-bit table {i P + j (beta x, -y)} instead of 255
// doublings with a wave-divergent addition behind each.  Without GLV: 255-bit double-and-add.
template <class C, bool GLV>
__global__ void __launch_bounds__(64)
k_g1_lincomb(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ scalars, uint32_t n_items, uint32_t tpi, uint32_t split,
             uint32_t *__restrict__ out) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  const uint32_t lane = threadIdx.x & 63, k = lane & 15;
  uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = item < n_items;
  if (!live) item = n_items - 1;
  typename CV::acc_t acc = CV::identity();
  if (k < tpi) {
    const size_t t = (size_t)item * tpi + k;
    const typename CV::base_t P = CV::load_base(bases + t * 2 * N);
    uint32_t sc[8];
    { const uint4 *sp = reinterpret_cast<const uint4 *>(scalars + t * 8); uint4 a = sp[0], b = sp[1];
      sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w; }
    if constexpr (GLV && N == 12) {
      typename CV::base_t Q;                                           // -phi(P)
      { fpn<N> beta;
#pragma unroll
        for (int i = 0; i < N; i++) beta.v[i] = BLS12_381_BETA_MONT[i];
        Q.x = fn_mul<Fq>(beta, P.x); Q.y = fn_neg<Fq>(P.y);
        if (fn_is_zero(P.x) && fn_is_zero(P.y)) Q = P; }               // infinity stays infinity
      typename CV::acc_t tab[16];                                      // tab[4 j + i] = i P + j Q (per-lane scratch)
      tab[0] = CV::identity(); tab[4] = CV::from_affine(Q);
      tab[8] = CV::madd(tab[4], Q, false); tab[12] = CV::madd(tab[8], Q, false);
#pragma unroll 1
      for (int j = 0; j < 4; j++)
#pragma unroll 1
        for (int i = 1; i < 4; i++) tab[4 * j + i] = CV::madd(tab[4 * j + i - 1], P, false);
#pragma unroll 1
      for (int w = 63; w >= 0; w--) {
        acc = CV::dbl(CV::dbl(acc));
        const uint32_t d1 = (sc[w >> 4] >> (2 * (w & 15))) & 3u, d2 = (sc[4 + (w >> 4)] >> (2 * (w & 15))) & 3u;
        if (d1 | d2) acc = CV::add(acc, tab[4 * d2 + d1]);
      }
    } else {
#pragma unroll 1
      for (int bit = 255; bit >= 0; bit--) {
        acc = CV::dbl(acc);
        if ((sc[bit >> 5] >> (bit & 31)) & 1) acc = CV::madd(acc, P, false);
      }
    }
  }
  // lanes 0 and `split` gather their segment (the other lanes add along and are ignored)
#pragma unroll 1
  for (uint32_t t = 1; t < 16; t++) {
    uint32_t src = k < split ? t : split + t;
    const bool inside = k < split ? src < split : src < tpi;
    if (src > 15) src = 15;
    typename CV::acc_t o;
#pragma unroll
    for (int i = 0; i < N; i++) {
      o.x.v[i] = __shfl(acc.x.v[i], src, 16); o.y.v[i] = __shfl(acc.y.v[i], src, 16);
      o.zz.v[i] = __shfl(acc.zz.v[i], src, 16); o.zzz.v[i] = __shfl(acc.zzz.v[i], src, 16);
    }
    if ((k == 0 || k == split) && inside) acc = CV::add(acc, o);
  }
  if (live && (k == 0 || k == split)) {
    uint32_t *o = out + ((size_t)item * 2 + (k == 0 ? 0 : 1)) * 2 * N;
    if (CV::is_identity(acc)) { fn_store<N>(o, fn_zero<N>()); fn_store<N>(o + N, fn_zero<N>()); }
    else {
      const fpn<N> inv = fn_inv<Fq>(fn_mul<Fq>(acc.zz, acc.zzz));
      fn_store<N>(o, fn_mul<Fq>(acc.x, fn_mul<Fq>(inv, acc.zzz)));       // X / ZZ
      fn_store<N>(o + N, fn_mul<Fq>(acc.y, fn_mul<Fq>(inv, acc.zz)));    // Y / ZZZ
    }
  }
}
void launch_g1_lincomb(int curve, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n_items, uint32_t tpi, uint32_t split, uint32_t *d_out,
                       hipStream_t stream, bool glv_split_scalars) {
  if (!n_items) return;
  const dim3 grid((unsigned)((n_items * 16 + 63) / 64)), block(64);
  if (curve == 0 && glv_split_scalars) hipLaunchKernelGGL((k_g1_lincomb<G1Bls12381, true>), grid, block, 0, stream, d_bases, d_scalars, (uint32_t)n_items, tpi, split, d_out);
  else if (curve == 0) hipLaunchKernelGGL((k_g1_lincomb<G1Bls12381, false>), grid, block, 0, stream, d_bases, d_scalars, (uint32_t)n_items, tpi, split, d_out);
  else hipLaunchKernelGGL((k_g1_lincomb<G1Bn254, false>), grid, block, 0, stream, d_bases, d_scalars, (uint32_t)n_items, tpi, split, d_out);
}

void g1_glv_split_bls(uint64_t k[4]) {
  const uint64_t z = 0xd201000000010000ull;
  uint64_t q1[4], q2[4];                                            // q1 = k div z, q2 = q1 div z; floor(floor(k / z) / z) = floor(k / z^2)
  unsigned __int128 rem = 0;
  for (int i = 3; i >= 0; i--) { unsigned __int128 cur = (rem << 64) | k[i]; q1[i] = (uint64_t)(cur / z); rem = cur % z; }
  rem = 0;
  for (int i = 3; i >= 0; i--) { unsigned __int128 cur = (rem << 64) | q1[i]; q2[i] = (uint64_t)(cur / z); rem = cur % z; }
  // k1 = k - q2 z^2 (fits 128 bits): low 128 bits of the difference
  const unsigned __int128 z2 = (unsigned __int128)z * z, q = ((unsigned __int128)q2[1] << 64) | q2[0];
  const unsigned __int128 klo = ((unsigned __int128)k[1] << 64) | k[0];
  const unsigned __int128 k1 = klo - q * z2;                         // mod 2^128; exact because k1 < z^2 < 2^128
  k[0] = (uint64_t)k1; k[1] = (uint64_t)(k1 >> 64); k[2] = q2[0]; k[3] = q2[1];
}

void launch_g1_decompress(int curve, const uint8_t *d_comp, size_t n, uint8_t *d_out_xy, uint8_t *d_ok, hipStream_t stream) {
  if (!n) return;
  dim3 g((unsigned)((n + 63) / 64)), b(64);
  if (curve == 0) hipLaunchKernelGGL(k_g1_decompress<G1Bls12381>, g, b, 0, stream, d_comp, (uint32_t)n, d_out_xy, d_ok);
  else hipLaunchKernelGGL(k_g1_decompress<G1Bn254>, g, b, 0, stream, d_comp, (uint32_t)n, d_out_xy, d_ok);
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
// ---------------------------------------------------------------- fixed-base MSM over a table of ALL multiples (no buckets, no sort)
//
// A KZG commitment is an MSM over ONE fixed base set (the SRS), and an MI355X has 288 GB of HBM.  G1DirectTable holds, for every
// window row w and base i, every multiple m 2^(c w) P_i a signed c-bit digit can ask for (m = 1 .. 2^(c-1)): a commitment is then the
// plain SUM of one table point per (coefficient, row) -- n * rows mixed additions into per-lane accumulators and a tree over the lanes'
// partial sums.  No digit sort, no bucket sums, no weighted bucket reduction (the latter two were a quarter of the ring prover's device
// time), and the window can be as wide as memory allows: c = 15 for the 6 145 powers of a ring-1024 setup = 17 rows instead of the 22 of
// the 12-bit bucket form (1.71 G points, 164 GB).  The gathers are random 96-byte reads over the whole table: 12.7 G/s measured on a
// 200 GB table (tools/gather_probe.hip, profiles/r6_gather_probe.txt), four times what 24 k proofs/s ask for, 8 us each -- covered by
// the LDS-DMA prefetch of k_accumulate, whose addition takes longer than that.
//   layout: point (w, m, i) = (m + 1) 2^(c w) P_i at t.off[w] + m * n + i (affine Montgomery, (0, 0) = infinity): a build step writes
//   whole rows of bases at a time.  Rows: R = ceil(bits / c) digit rows; the top one holds only the multiples its digits can reach, and a
//   carry row (one multiple) follows when the signed recoding can carry out of it.
template <class C> static void direct_shape(size_t n, int c, G1DirectTable *t) {
  constexpr int BITS = C::Fr::BITS;
  const int R = (BITS + c - 1) / c;
  const uint32_t nb = 1u << (c - 1);
  // largest digit of the top row: (r - 1) >> (c (R - 1)), plus the carry coming in
  uint32_t top = 0;
  { const int bit = c * (R - 1); uint64_t w[9] = {}; for (int k = 0; k < 8; k++) w[k] = C::Fr::P[k];
    uint64_t borrow = 1; for (int k = 0; k < 8; k++) { const uint64_t d = w[k] - borrow; borrow = w[k] < borrow; w[k] = d & 0xffffffffu; }     // r - 1
    const int li = bit >> 5, sh = bit & 31; const uint64_t two = w[li] | (w[li + 1] << 32); top = (uint32_t)((two >> sh) & ((1ull << c) - 1)) + 1u; }
  t->c = c; t->n = n; t->rows = R;
  uint64_t off = 0;
  for (int w = 0; w < R; w++) { t->mult[w] = (w + 1 < R) ? nb : (top <= nb ? top : nb); t->off[w] = off; off += (uint64_t)t->mult[w] * n; }
  if (top > nb) { t->mult[R] = 1; t->off[R] = off; off += n; t->rows = R + 1; }
  t->points = off; t->bytes = off * 2 * C::Fq::N * 4;
}
size_t g1_direct_table_shape(int curve, size_t n, int c, G1DirectTable *t) {
  G1DirectTable tmp; if (!t) t = &tmp;
  t->curve = curve;
  if (curve == 0) direct_shape<G1Bls12381>(n, c, t); else direct_shape<G1Bn254>(n, c, t);
  return t->bytes;
}

// One lane per (row, base): the multiples m0 + 1 .. m0 + K of its base, as k_g1_table normalises a lane's rows -- X, Y into the table
// slot, ZZ, ZZZ and the running product into tmp, ONE inversion per lane and chunk, Montgomery's trick backwards.
template <class C>
__global__ void __launch_bounds__(64)
k_g1_multiples(const uint32_t *__restrict__ rowbase, uint32_t n, uint32_t rows, G1DirectTable t, uint32_t m0, uint32_t K, uint32_t *__restrict__ table,
               uint32_t *__restrict__ tmp, uint32_t lanes) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= lanes) return;
  const uint32_t w = lane / n, i = lane - w * n;
  if (m0 >= t.mult[w]) return;
  const uint32_t kend = m0 + K < t.mult[w] ? m0 + K : t.mult[w];
  const typename CV::base_t B = CV::load_base(rowbase + ((size_t)w * n + i) * 2 * N);
  typename CV::acc_t acc = CV::from_affine(B);
  if (m0) acc = CV::madd(CV::from_affine(CV::load_base(table + (t.off[w] + (uint64_t)(m0 - 1) * n + i) * 2 * N)), B, false);
  fpn<N> run = fn_one<Fq>();
#pragma unroll 1
  for (uint32_t m = m0; m < kend; m++) {
    uint32_t *o = table + (t.off[w] + (uint64_t)m * n + i) * 2 * N;
    uint32_t *q = tmp + ((size_t)(m - m0) * lanes + lane) * 3 * N;
    fn_store<N>(o, acc.x); fn_store<N>(o + N, acc.y);
    fn_store<N>(q, acc.zz); fn_store<N>(q + N, acc.zzz); fn_store<N>(q + 2 * N, run);
    if (!CV::is_identity(acc)) run = fn_mul<Fq>(run, fn_mul<Fq>(acc.zz, acc.zzz));
    if (m + 1 < kend) acc = m == 0 ? CV::dbl(acc) : CV::madd(acc, B, false);
  }
  fpn<N> inv = fn_inv<Fq>(run);
#pragma unroll 1
  for (uint32_t m = kend; m-- > m0;) {
    uint32_t *o = table + (t.off[w] + (uint64_t)m * n + i) * 2 * N;
    const uint32_t *q = tmp + ((size_t)(m - m0) * lanes + lane) * 3 * N;
    const fpn<N> zz = fn_load<N>(q), zzz = fn_load<N>(q + N);
    if (fn_is_zero(zz)) { fn_store<N>(o, fn_zero<N>()); fn_store<N>(o + N, fn_zero<N>()); continue; }
    const fpn<N> dinv = fn_mul<Fq>(inv, fn_load<N>(q + 2 * N));              // 1 / (ZZ ZZZ) of this multiple
    inv = fn_mul<Fq>(inv, fn_mul<Fq>(zz, zzz));
    fn_store<N>(o, fn_mul<Fq>(fn_load<N>(o), fn_mul<Fq>(dinv, zzz)));        // X / ZZ
    fn_store<N>(o + N, fn_mul<Fq>(fn_load<N>(o + N), fn_mul<Fq>(dinv, zz)));  // Y / ZZZ
  }
}
template <class C> static void build_direct_impl(const uint32_t *d_bases, G1DirectTable *t, hipStream_t stream) {
  constexpr int N = C::Fq::N;
  const size_t n = t->n; const uint32_t rows = (uint32_t)t->rows;
  if (t->points >= 0x7fffffffull) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};     // 31-bit entry indices
  HIP_CHECK(hipMalloc(&t->d, t->bytes));
  uint32_t *rowbase = nullptr, *tmp = nullptr;
  HIP_CHECK(hipMalloc(&rowbase, (size_t)rows * n * 2 * N * 4));
  build_g1_table(t->curve, d_bases, n, t->c, (int)rows, rowbase, stream);
  const uint32_t lanes = (uint32_t)(rows * n), K = 128;
  HIP_CHECK(hipMalloc(&tmp, (size_t)K * lanes * 3 * N * 4));
  uint32_t mmax = 0; for (uint32_t w = 0; w < rows; w++) mmax = t->mult[w] > mmax ? t->mult[w] : mmax;
  for (uint32_t m0 = 0; m0 < mmax; m0 += K)
    hipLaunchKernelGGL(k_g1_multiples<C>, dim3((lanes + 63) / 64), dim3(64), 0, stream, (const uint32_t *)rowbase, (uint32_t)n, rows, *t, m0, K, t->d, tmp, lanes);
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipFree(tmp)); HIP_CHECK(hipFree(rowbase));
}
void build_g1_direct_table(int curve, const uint32_t *d_bases, size_t n, int c, G1DirectTable *t, hipStream_t stream) {
  g1_direct_table_shape(curve, n, c, t);
  if (curve == 0) build_direct_impl<G1Bls12381>(d_bases, t, stream); else build_direct_impl<G1Bn254>(d_bases, t, stream);
}
void free_g1_direct_table(G1DirectTable *t) { if (t->d) (void)hipFree(t->d); t->d = nullptr; }

constexpr uint32_t DIRECT_ZERO = 0xffffffffu;
// entry (vector, coefficient i, row w) -> table index | sign << 31 (DIRECT_ZERO for a zero digit); the digits are k_digits' signed recoding
// (base_idx != nullptr: the sparse form -- entry i of vector b multiplies table base base_idx[b * n + i])
__global__ void __launch_bounds__(256) k_direct_index(const uint32_t *__restrict__ scalars, uint32_t n, uint32_t stride, G1DirectTable t, uint32_t *__restrict__ idx, int mont,
                                                      const uint32_t *__restrict__ base_idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t bat = blockIdx.y, rows = (uint32_t)t.rows, tn = (uint32_t)t.n;
  const int c = t.c;
  uint32_t s[9];
  const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * ((size_t)bat * stride + i));
  const uint4 a = p[0], b = p[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w; s[8] = 0;
  if (mont) {
    fp v;
#pragma unroll
    for (int k = 0; k < 8; k++) v.v[k] = s[k];
    v = mont == 1 ? fp_from_mont<FqBandersnatch>(v) : fp_from_mont<FqBabyJubJub>(v);
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = v.v[k];
  }
  uint32_t *o = idx + ((size_t)bat * n + i) * rows;
  const uint32_t ib = base_idx ? base_idx[(size_t)bat * n + i] : i;
  const uint32_t nb = 1u << (c - 1), mask = (1u << c) - 1;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < rows; w++) {
    const int bit = (int)w * c;
    uint32_t v = 0;
    if (bit < 256) {
      const int li = bit >> 5, sh = bit & 31;
      const uint64_t two = (uint64_t)s[li] | ((uint64_t)(li + 1 < 9 ? s[li + 1] : 0u) << 32);
      v = (uint32_t)(two >> sh) & mask;
    }
    v += carry;
    uint32_t m, sign = 0;
    if (v > nb) { m = (1u << c) - v; sign = 0x80000000u; carry = 1; } else { m = v; carry = 0; }
    // (a scalar below r never asks a row for more than it holds: direct_shape sized the top row and the carry row from r - 1)
    o[w] = (m == 0 || m > t.mult[w] || ib >= tn) ? DIRECT_ZERO : (uint32_t)(t.off[w] + (uint64_t)(m - 1) * tn + ib) | sign;
  }
}

// Lane t sums the table points of the entries [e0, e1) of ONE vector (t / lpv); its partial sum goes to part[t].  The gather
// pipeline is k_accumulate's: the point of entry k + 1 travels HBM -> LDS by DMA while the addition of entry k runs.
template <class CV>
__global__ void __launch_bounds__(256, CV::MIN_WAVES)
k_accumulate_direct(const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t E, uint32_t lpv, uint32_t per, uint32_t batch,
                    uint32_t *__restrict__ part) {
  using AC = typename CV::accum; using acc_t = typename AC::acc_t; using base_t = typename CV::base_t;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t vec = t / lpv, r = t - vec * lpv;
  extern __shared__ uint32_t acc_lds[];
  constexpr int BW = CV::BASE_WORDS, CH = BW / 4;
  uint32_t *wbuf = acc_lds + (threadIdx.x >> 6) * (2 * BW * 64);
  const uint32_t lane = threadIdx.x & 63;
  const bool live = vec < batch;
  const size_t v0 = (size_t)(live ? vec : 0) * E;
  uint32_t cnt = 0; size_t e0 = v0;
  if (live && (uint64_t)r * per < E) { e0 = v0 + (size_t)r * per; cnt = (uint64_t)r * per + per <= E ? per : (uint32_t)(E - (uint64_t)r * per); }
  auto issue = [&](uint32_t ix, uint32_t slot) {
    const uint32_t *src = table + (size_t)(ix == DIRECT_ZERO ? 0u : (ix & 0x7fffffffu)) * BW;
#pragma unroll
    for (int c = 0; c < CH; c++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * c),
                                       (__attribute__((address_space(3))) void *)(wbuf + slot * (BW * 64) + c * 256), 16, 0, 0);
  };
  auto fetch = [&](uint32_t slot) {
    uint32_t w[BW];
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(wbuf + slot * (BW * 64) + c * 256 + lane * 4);
      w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
    }
    return CV::base_from_words(w);
  };
  acc_t acc = AC::identity();
  uint32_t ix = cnt ? idx[e0] : DIRECT_ZERO;
  uint32_t ix1 = cnt > 1 ? idx[e0 + 1] : DIRECT_ZERO;
  if (cnt) issue(ix, 0);
  for (uint32_t k = 0; k < cnt; k++) {
    const uint32_t cix = ix;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // the DMA of this buffer, and the reads of the other one before it is overwritten
    const base_t cur = fetch(k & 1);
    ix = ix1;
    if (k + 1 < cnt) issue(ix, (k + 1) & 1);
    if (k + 2 < cnt) ix1 = idx[e0 + k + 2];
    if (cix != DIRECT_ZERO) acc = AC::madd(acc, cur, (cix & 0x80000000u) != 0);
  }
  if (t < batch * lpv) AC::store_part(part + (size_t)t * AC::PART_WORDS, acc);
}
// one WAVE per vector: lane t sums the partials t, t + 64, .. of its vector one after the other, then the wave's butterfly -- 64 (lpv / 64 + 5)
// lane-additions per vector instead of the 7 lpv of a lane per partial (the kernel is latency-bound either way; what it does not issue, the
// other contexts' accumulations do) -> one point (canonical XYZZ for the host, as the bucket form's k_wsum leaves it)
template <class RV>
__global__ void __launch_bounds__(64, RV::RED_WAVES)
k_direct_reduce(const uint32_t *__restrict__ part, uint32_t lpv, uint32_t *__restrict__ out) {
  using acc_t = typename RV::acc_t;
  const uint32_t vec = blockIdx.x, t = threadIdx.x;
  acc_t a = RV::identity();
#pragma unroll 1
  for (uint32_t k = t; k < lpv; k += 64) a = cv_add<RV>(a, RV::accum::load_part(part + ((size_t)vec * lpv + k) * RV::accum::PART_WORDS));
  a = wave_sum<RV>(a);
  if (t == 0) RV::store_out(out + (size_t)vec * RV::OUT_WORDS, a);
}
template <class C>
static int msm_g1_direct_impl(const G1DirectTable &t, const uint32_t *d_scalars, size_t n, size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream,
                              uint8_t *out_xy, size_t batch, int scalars_mont, const uint32_t *d_base_idx) {
  using CV = G1Curve<C>; using RV = typename CV::red; using HG = HostG1<C>;
  if (!n || !batch) return -1;
  const size_t rows = (size_t)t.rows, E = n * rows;
  if (E * batch >= 0xffff0000ull || (!d_base_idx && n > t.n)) throw HipFailure{hipErrorInvalidValue, __FILE__, __LINE__};
  const AccShape shape = accumulate_shape<CV>();
  size_t lpv = shape.lanes / batch; if (lpv > 1024) lpv = 1024; if (lpv < 1) lpv = 1;
  const size_t per = (E + lpv - 1) / lpv, lanes = batch * lpv;
  // workspace: the entry indices take `sorted`, the partial sums `part`, the results `rc` / `bits_host`
  MsmPlan p; p.c = t.c; p.nwin = 1; p.nb = 1; p.lpb = (int)per;
  ws.ensure(E, p, (size_t)RV::OUT_WORDS * 4 > (size_t)RV::ACC_WORDS * 4 ? (size_t)RV::OUT_WORDS * 4 : (size_t)RV::ACC_WORDS * 4, batch, lanes + 256,
            (size_t)CV::accum::PART_WORDS * 4);
  static std::once_flag once[2];
  std::call_once(once[t.curve ? 1 : 0], [&]() {
    if (shape.lds > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)k_accumulate_direct<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shape.lds));
  });
  hipLaunchKernelGGL(k_direct_index, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, stream, d_scalars, (uint32_t)n, (uint32_t)(scalar_stride ? scalar_stride : n),
                     t, ws.sorted, scalars_mont, d_base_idx);
  if (!ws.ev0) { HIP_CHECK(hipEventCreate(&ws.ev0)); HIP_CHECK(hipEventCreate(&ws.ev1)); }
  HIP_CHECK(hipEventRecord(ws.ev0, stream));
  hipLaunchKernelGGL(k_accumulate_direct<CV>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), shape.lds, stream, (const uint32_t *)t.d, (const uint32_t *)ws.sorted,
                     (uint32_t)E, (uint32_t)lpv, (uint32_t)per, (uint32_t)batch, ws.part);
  HIP_CHECK(hipEventRecord(ws.ev1, stream));
  hipLaunchKernelGGL(k_direct_reduce<RV>, dim3((unsigned)batch), dim3(64), 0, stream, (const uint32_t *)ws.part, (uint32_t)lpv, ws.rc);
  HIP_CHECK(hipMemcpyAsync(ws.bits_host, ws.rc, batch * (size_t)RV::OUT_WORDS * 4, hipMemcpyDeviceToHost, stream));
  ws.plan_host[0] = (uint32_t)per;
  ws.pending_plan = p;
  msm_wait(ws, stream);
  std::vector<typename HG::Pt> res(batch);
  for (size_t b = 0; b < batch; b++) res[b] = HG::from_raw32(ws.bits_host + b * 4 * C::Fq::N);
  HG::to_affine_bytes_batch(res.data(), batch, out_xy);
  return 0;
}
int msm_g1_direct_device(const G1DirectTable &t, const uint32_t *d_scalars, size_t n, size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream,
                         uint8_t *out_xy, size_t batch, int scalars_mont, const uint32_t *d_base_idx) {
  if (t.curve == 0) return msm_g1_direct_impl<G1Bls12381>(t, d_scalars, n, scalar_stride, ws, stream, out_xy, batch, scalars_mont, d_base_idx);
  return msm_g1_direct_impl<G1Bn254>(t, d_scalars, n, scalar_stride, ws, stream, out_xy, batch, scalars_mont, d_base_idx);
}

int msm_g1_fixed_device(int curve, const uint32_t *d_table, int table_c, size_t table_stride, const uint32_t *d_scalars, size_t n,
                        size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy, size_t batch, const uint32_t *d_base_idx,
                        int scalars_mont) {
  if (curve == 0) return msm_g1_impl<G1Bls12381>(d_table, d_scalars, n, ws, stream, out_xy, batch, table_c, table_stride, scalar_stride, d_base_idx, scalars_mont);
  if (curve == 1) return msm_g1_impl<G1Bn254>(d_table, d_scalars, n, ws, stream, out_xy, batch, table_c, table_stride, scalar_stride, d_base_idx, scalars_mont);
  return -1;
}

}  // namespace avrf
