// ring.hip -- Ring-VRF membership SNARK: setup (SRS), ring indexing (`ring_proof::index`,
// src/ring.rs:404,416) and the ring prover (`RingProver::prove`, src/ring.rs:220), following the
// byte-exact specification in SURVEY.md Appendix A.5 / A.7 (w3f-ring-proof 0.0.10, un-vendored).
//
// Split of the work: proofs are proved in lockstep chunks.  On the device: the witness columns
// (expanded from their sparse description), all NTTs, the constraint aggregation on the 4N domain, quotient,
// evaluations, linearisation and opening quotients (k_ring_* / k_ntt_* below) and every KZG commitment as a batched
// fixed-base MSM (msm.hip) over window tables of the SRS -- the four witness columns in the Lagrange basis, where
// they are sparse -- and (round 4) the <= 254 twisted-Edwards additions of every proof's witness accumulator with their
// batch normalisation (k_ring_witness_acc).  On host threads between the device rounds: only the Fiat-Shamir transcript
// (SHAKE128).  Verification: transcript replay and scalars on the
// host pool, two G1 MSMs on the device, one 2-pairing check on the host (host_pairing.h).
#include "../../include/avrf.h"
#include "te.h"
#include "te_quad.h"
#include "host_g1.h"
#include "host_pairing.h"
#include "host_te.h"
#include "msm.h"
#include "pairing.h"
#include "suite_dispatch.h"
#include "host_shake128.h"
#include "host_sha512.h"
#include "host_sha256.h"
#include "host_pool.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/random.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace avrf {

// a failed HIP call unwinds to the extern "C" entry point (guarded() below), which returns AVRF_ERR_NO_DEVICE
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw avrf::HipFailure{e_, __FILE__, __LINE__}; } while (0)

// ------------------------------------------------------------------------------------------------
// device NTT over Fr of the pairing curve (radix-2, stages fused through LDS; batch of equal sizes);
// tw[k] = w^k (Montgomery), k < n/2

template <class F>
__global__ void k_ntt_bitrev(uint32_t *__restrict__ data, uint32_t n, int logn, uint32_t batch) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t b = t / n, i = t - b * n;
  if (b >= batch) return;
  uint32_t j = __brev(i) >> (32 - logn);
  if (i < j) {
    uint32_t *p = data + ((size_t)b * n + i) * 8, *q = data + ((size_t)b * n + j) * 8;
    fp x = load_fp(p), y = load_fp(q);
    store_fp(p, y); store_fp(q, x);
  }
}
template <class F>
__global__ void k_ntt_scale(uint32_t *__restrict__ data, uint32_t total, fp k) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  uint32_t *p = data + (size_t)t * 8;
  store_fp(p, fp_mul<F>(load_fp(p), k));
}

// Up to 8 consecutive butterfly stages [s0, s1) on a tile held in LDS: 2^(s1-s0) rows with row stride 2^s0 elements
// times 2^cols_log adjacent columns (<= 256 elements, whole 128-byte lines when s0 > 0).  A size-2^13 transform is
// 2 passes over HBM instead of 13.  brev_load: the tile is gathered from the bit-reversed positions of `src` (another
// buffer, `src_n` <= n valid elements per vector, the rest read as zero -- zero-extension of a shorter coefficient
// vector costs nothing); otherwise src == dst is read in place.  do_scale: multiply by `scale` on the way out.
template <class F>
__global__ void __launch_bounds__(128)
k_ntt_fused(const uint32_t *__restrict__ src, uint32_t src_n, uint32_t *__restrict__ dst, uint32_t n, int logn, int s0, int s1, uint32_t cols_log,
            const uint32_t *__restrict__ tw, int brev_load, fp scale, int do_scale) {
  __shared__ uint32_t sh[256 * 8];
  const uint32_t cols = 1u << cols_log, tile = (1u << (s1 - s0)) << cols_log;
  const uint32_t tiles_per_vec = n / tile, b = blockIdx.x / tiles_per_vec, t = blockIdx.x - b * tiles_per_vec;
  const uint32_t low_groups = (1u << s0) >> cols_log;
  const uint32_t hi = t / low_groups, lo = t - hi * low_groups;
  const uint32_t base = (hi << s1) + (lo << cols_log);
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t i = base + ((e >> cols_log) << s0) + (e & (cols - 1));
    fp v;
    if (brev_load) { const uint32_t j = __brev(i) >> (32 - logn); v = j < src_n ? load_fp(src + ((size_t)b * src_n + j) * 8) : fp_zero(); }
    else v = load_fp(src + ((size_t)b * n + i) * 8);
    store_fp(sh + e * 8, v);
  }
  __syncthreads();
  // (the twiddles are read from global memory inside the butterflies: preloading the tile - cols <= 255 twiddles of a pass into a
  // second 8 KB of LDS measured slower -- 559 us against 400 us per launch in the 2 048-proof profile, proofs/s unchanged)
  for (int s = s0; s < s1; s++) {
    const int hl = s - s0;                                             // half = 2^hl rows
    for (uint32_t j = threadIdx.x; j < tile / 2; j += blockDim.x) {
      const uint32_t c = j & (cols - 1), jr = j >> cols_log;
      const uint32_t grp = jr >> hl, pos = jr & ((1u << hl) - 1);
      const uint32_t r0 = (grp << (hl + 1)) + pos, r1 = r0 + (1u << hl);
      const uint32_t pg = (pos << s0) + (lo << cols_log) + c;          // position inside the stage's global group
      const fp w = load_fp(tw + ((size_t)pg << (logn - 1 - s)) * 8);
      uint32_t *p = sh + ((r0 << cols_log) + c) * 8, *q = sh + ((r1 << cols_log) + c) * 8;
      const fp u = load_fp(p), v = fp_mul<F>(load_fp(q), w);
      store_fp(p, fp_add<F>(u, v)); store_fp(q, fp_sub<F>(u, v));
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
    const uint32_t i = base + ((e >> cols_log) << s0) + (e & (cols - 1));
    fp v = load_fp(sh + e * 8);
    if (do_scale) v = fp_mul<F>(v, scale);
    store_fp(dst + ((size_t)b * n + i) * 8, v);
  }
}

// (i)NTT of `batch` vectors of size n (tw = powers of the root or of its inverse; scale = 1/n for the inverse).
// src == dst: in place (a bit-reversal pass first).  Otherwise dst = NTT(zero-extended src), src vectors of src_n <= n.
template <class F>
static void ntt_launch2(const uint32_t *d_src, uint32_t src_n, uint32_t *d_dst, uint32_t n, const uint32_t *d_tw, uint32_t batch, const fp *scale, hipStream_t st) {
  int logn = 0; while ((1u << logn) < n) logn++;
  const bool inplace = d_src == d_dst;
  if (inplace) hipLaunchKernelGGL(k_ntt_bitrev<F>, dim3(((size_t)n * batch + 255) / 256), dim3(256), 0, st, d_dst, n, logn, batch);
  fp one; memset(&one, 0, sizeof one);
  for (int s0 = 0; s0 < logn;) {
    int s1 = s0 + 8 < logn ? s0 + 8 : logn;
    if (s1 < logn && logn - s1 < 3) s1 = logn - 3 > s0 ? logn - 3 : s1;   // keep the last pass at >= 3 stages
    uint32_t cols_log = 8 - (uint32_t)(s1 - s0); if ((int)cols_log > s0) cols_log = (uint32_t)s0;
    const uint32_t tile = (1u << (s1 - s0)) << cols_log;
    const bool last = s1 == logn;
    hipLaunchKernelGGL(k_ntt_fused<F>, dim3((unsigned)((size_t)(n / tile) * batch)), dim3(tile / 2 < 128 ? (tile / 2 ? tile / 2 : 1) : 128), 0, st,
                       (s0 == 0 && !inplace) ? d_src : (const uint32_t *)d_dst, src_n, d_dst, n, logn, s0, s1, cols_log, d_tw,
                       (s0 == 0 && !inplace) ? 1 : 0, (last && scale) ? *scale : one, (last && scale) ? 1 : 0);
    s0 = s1;
  }
}
template <class F>
static void ntt_launch(uint32_t *d_data, uint32_t n, const uint32_t *d_tw, uint32_t batch, const fp *scale, hipStream_t st) {
  ntt_launch2<F>(d_data, n, d_data, n, d_tw, batch, scale, st);
}

// ------------------------------------------------------------------------------------------------
// PIOP constraint aggregation on the 4N domain: one lane per domain point.
// e4 = evaluations of bits | ip_acc | acc_x | acc_y (4 x M), fixed4 = points.x | points.y | selector
// (3 x M), l4 = L_first | L_last (2 x M), tw4[k] = w4^k (k < M/2; w4^(M/2) = -1).
struct RingConsts { fp alpha[7]; fp w_last, seedx, seedy, resx, resy; };

template <class S>
__global__ void __launch_bounds__(256)
k_ring_constraints(const uint32_t *__restrict__ e4, const uint32_t *__restrict__ fixed4, const uint32_t *__restrict__ l4,
                   const uint32_t *__restrict__ tw4, const RingConsts *__restrict__ consts, uint32_t M, uint32_t *__restrict__ out) {
  using F = typename S::Fq;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const uint32_t proof = blockIdx.y;                                  // one proof per grid row
  e4 += (size_t)proof * 4 * M * 8; out += (size_t)proof * M * 8;
  const RingConsts &c = consts[proof];
  const uint32_t is = (i + 4) & (M - 1);                              // column(wX) on the 4N domain
  auto at = [&](const uint32_t *base, uint32_t col, uint32_t idx) { return load_fp(base + ((size_t)col * M + idx) * 8); };
  const fp one = fp_one<F>();
  const fp b = at(e4, 0, i), ip = at(e4, 1, i), x1 = at(e4, 2, i), y1 = at(e4, 3, i);
  const fp ips = at(e4, 1, is), x3 = at(e4, 2, is), y3 = at(e4, 3, is);
  const fp x2 = at(fixed4, 0, i), y2 = at(fixed4, 1, i), sel = at(fixed4, 2, i);
  const fp lf = at(l4, 0, i), ll = at(l4, 1, i);
  fp xi = load_fp(tw4 + (size_t)(i & (M / 2 - 1)) * 8);
  if (i >= M / 2) xi = fp_neg<F>(xi);
  const fp nl = fp_sub<F>(xi, c.w_last), omb = fp_sub<F>(one, b);
  const fp x1y1 = fp_mul<F>(x1, y1), x2y2 = fp_mul<F>(x2, y2);
  fp c0 = fp_mul<F>(fp_sub<F>(fp_sub<F>(ips, ip), fp_mul<F>(sel, b)), nl);
  fp t1 = fp_add<F>(fp_mul<F>(y1, y2), mul_a<S>(fp_mul<F>(x1, x2)));
  fp c1 = fp_mul<F>(fp_add<F>(fp_mul<F>(b, fp_sub<F>(fp_sub<F>(fp_mul<F>(x3, t1), x1y1), x2y2)), fp_mul<F>(omb, fp_sub<F>(x3, x1))), nl);
  fp t2 = fp_sub<F>(fp_mul<F>(x1, y2), fp_mul<F>(x2, y1));
  fp c2 = fp_mul<F>(fp_add<F>(fp_mul<F>(b, fp_add<F>(fp_sub<F>(fp_mul<F>(y3, t2), x1y1), x2y2)), fp_mul<F>(omb, fp_sub<F>(y3, y1))), nl);
  fp c3 = fp_mul<F>(b, omb);
  fp c4 = fp_add<F>(fp_mul<F>(lf, fp_sub<F>(x1, c.seedx)), fp_mul<F>(ll, fp_sub<F>(x1, c.resx)));
  fp c5 = fp_add<F>(fp_mul<F>(lf, fp_sub<F>(y1, c.seedy)), fp_mul<F>(ll, fp_sub<F>(y1, c.resy)));
  fp c6 = fp_add<F>(fp_mul<F>(lf, ip), fp_mul<F>(ll, fp_sub<F>(ip, one)));
  fp s = fp_mul<F>(c.alpha[0], c0);
  s = fp_add<F>(s, fp_mul<F>(c.alpha[1], c1)); s = fp_add<F>(s, fp_mul<F>(c.alpha[2], c2)); s = fp_add<F>(s, fp_mul<F>(c.alpha[3], c3));
  s = fp_add<F>(s, fp_mul<F>(c.alpha[4], c4)); s = fp_add<F>(s, fp_mul<F>(c.alpha[5], c5)); s = fp_add<F>(s, fp_mul<F>(c.alpha[6], c6));
  store_fp(out + (size_t)i * 8, s);
}


// ------------------------------------------------------------------------------------------------
// per-proof polynomial rounds of the prover on the device; grid.y / blockIdx = proof.
// All vectors are Montgomery Fr, 8 words per element.

struct Fp3 { fp v[3]; };

template <class F> AVRF_DI fp fp_pow_u32(fp a, uint32_t e) {
  fp r = fp_one<F>();
  while (e) { if (e & 1) r = fp_mul<F>(r, a); e >>= 1; if (e) a = fp_sqr<F>(a); }
  return r;
}
// sum over the 256 lanes of a workgroup (sh: 256 x 8 words); result valid in every lane
template <class F> AVRF_DI fp block_sum256(fp v, uint32_t *sh) {
  const uint32_t t = threadIdx.x;
  __syncthreads();
  store_fp(sh + t * 8, v);
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (t < s) store_fp(sh + t * 8, fp_add<F>(load_fp(sh + t * 8), load_fp(sh + (t + s) * 8)));
    __syncthreads();
  }
  return load_fp(sh);
}
// value at x of the length-len polynomial c: lane t takes coefficients [t*L, (t+1)*L)
template <class F> AVRF_DI fp block_eval256(const uint32_t *c, uint32_t len, const fp &x, uint32_t *sh) {
  const uint32_t t = threadIdx.x, L = (len + 255) / 256;
  uint32_t lo = t * L, hi = lo + L; if (hi > len) hi = len;
  fp h = fp_zero();
  for (uint32_t i = hi; i > lo; i--) h = fp_add<F>(fp_mul<F>(h, x), load_fp(c + (size_t)(i - 1) * 8));
  if (lo < hi) h = fp_mul<F>(h, fp_pow_u32<F>(x, lo));
  return block_sum256<F>(h, sh);
}

// quotient: q = agg * Z / (X^N - 1) with Z = X^3 + z2 X^2 + z1 X + z0 (the three zk rows), agg of length M = 4N:
// t[k] = sum_m z[m] agg[k-m];  q[i] = sum_{j >= 1} t[i + jN]   (exact division: the remainder is zero)
template <class F>
__global__ void __launch_bounds__(256)
k_ring_quotient(const uint32_t *__restrict__ agg, uint32_t N, uint32_t qlen, Fp3 z, uint32_t *__restrict__ q) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, M = 4 * N;
  if (i >= qlen) return;
  const uint32_t *a = agg + (size_t)blockIdx.y * M * 8;
  fp acc = fp_zero();
  for (uint32_t k = i + N; k <= M + 2; k += N) {
    if (k < M) acc = fp_add<F>(acc, fp_mul<F>(z.v[0], load_fp(a + (size_t)k * 8)));
    if (k >= 1 && k - 1 < M) acc = fp_add<F>(acc, fp_mul<F>(z.v[1], load_fp(a + (size_t)(k - 1) * 8)));
    if (k >= 2 && k - 2 < M) acc = fp_add<F>(acc, fp_mul<F>(z.v[2], load_fp(a + (size_t)(k - 2) * 8)));
    if (k >= 3 && k - 3 < M) acc = fp_add<F>(acc, load_fp(a + (size_t)(k - 3) * 8));
  }
  store_fp(q + ((size_t)blockIdx.y * qlen + i) * 8, acc);
}

// ev[proof][c] = poly_c(zeta_proof), c: px | py | sel (fixed, shared) | bits | ip | ax | ay (coef: proof x 4 x N)
template <class F>
__global__ void __launch_bounds__(256)
k_ring_evals(const uint32_t *__restrict__ coef, const uint32_t *__restrict__ fixed, const uint32_t *__restrict__ zeta, uint32_t N,
             uint32_t *__restrict__ ev) {
  __shared__ uint32_t sh[256 * 8];
  const uint32_t c = blockIdx.x, p = blockIdx.y;
  const uint32_t *poly = c < 3 ? fixed + (size_t)c * N * 8 : coef + ((size_t)p * 4 + (c - 3)) * N * 8;
  fp v = block_eval256<F>(poly, N, load_fp(zeta + (size_t)p * 8), sh);
  if (threadIdx.x == 0) store_fp(ev + ((size_t)p * 7 + c) * 8, v);
}

// linearisation polynomial lin = f0 ip + f1 ax + f2 ay (f from the evaluations, A.7 step 6) and lin(zeta w)
template <class S>
__global__ void __launch_bounds__(256)
k_ring_lin(const uint32_t *__restrict__ coef, const RingConsts *__restrict__ consts, const uint32_t *__restrict__ zeta,
           const uint32_t *__restrict__ ev, fp w, uint32_t N, uint32_t *__restrict__ lin, uint32_t *__restrict__ lin_zw) {
  using F = typename S::Fq;
  __shared__ uint32_t sh[256 * 8];
  const uint32_t p = blockIdx.x, t = threadIdx.x;
  const RingConsts &c = consts[p];
  const uint32_t *e = ev + (size_t)p * 7 * 8;
  const fp z = load_fp(zeta + (size_t)p * 8), one = fp_one<F>();
  const fp x2 = load_fp(e), y2 = load_fp(e + 8), b = load_fp(e + 24), x1 = load_fp(e + 40), y1 = load_fp(e + 48);
  const fp nlz = fp_sub<F>(z, c.w_last), omb = fp_sub<F>(one, b);
  const fp k1 = fp_add<F>(fp_mul<F>(b, fp_add<F>(fp_mul<F>(y1, y2), mul_a<S>(fp_mul<F>(x1, x2)))), omb);
  const fp k2 = fp_add<F>(fp_mul<F>(b, fp_sub<F>(fp_mul<F>(x1, y2), fp_mul<F>(x2, y1))), omb);
  const fp f0 = fp_mul<F>(nlz, c.alpha[0]), f1 = fp_mul<F>(nlz, fp_mul<F>(c.alpha[1], k1)), f2 = fp_mul<F>(nlz, fp_mul<F>(c.alpha[2], k2));
  const uint32_t *ip = coef + ((size_t)p * 4 + 1) * N * 8, *ax = ip + (size_t)N * 8, *ay = ax + (size_t)N * 8;
  uint32_t *out = lin + (size_t)p * N * 8;
  for (uint32_t i = t; i < N; i += 256)
    store_fp(out + (size_t)i * 8, fp_add<F>(fp_add<F>(fp_mul<F>(f0, load_fp(ip + (size_t)i * 8)), fp_mul<F>(f1, load_fp(ax + (size_t)i * 8))),
                                            fp_mul<F>(f2, load_fp(ay + (size_t)i * 8))));
  __syncthreads();
  fp v = block_eval256<F>(out, N, fp_mul<F>(z, w), sh);
  if (t == 0) store_fp(lin_zw + (size_t)p * 8, v);
}

// aggregated opening polynomial at zeta: sum_c nu_c poly_c + nu_7 q   (length qlen; the 7 columns have length N)
template <class F>
__global__ void __launch_bounds__(256)
k_ring_aggz(const uint32_t *__restrict__ coef, const uint32_t *__restrict__ fixed, const uint32_t *__restrict__ q, const uint32_t *__restrict__ nu,
            uint32_t N, uint32_t qlen, uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= qlen) return;
  const uint32_t *nup = nu + (size_t)p * 8 * 8;
  fp acc = fp_mul<F>(load_fp(nup + 7 * 8), load_fp(q + ((size_t)p * qlen + i) * 8));
  if (i < N) {
    for (int c = 0; c < 3; c++) acc = fp_add<F>(acc, fp_mul<F>(load_fp(nup + c * 8), load_fp(fixed + ((size_t)c * N + i) * 8)));
    for (int c = 0; c < 4; c++) acc = fp_add<F>(acc, fp_mul<F>(load_fp(nup + (3 + c) * 8), load_fp(coef + (((size_t)p * 4 + c) * N + i) * 8)));
  }
  store_fp(out + ((size_t)p * qlen + i) * 8, acc);
}

// out = (c(X) - c(z)) / (X - z), z = zs[proof] * zmul: out[i-1] = A_i = c[i] + z A_{i+1}.  Lane t owns indices
// [t*L, (t+1)*L); the carry into a chunk comes from a suffix scan of the chunk values with multiplier z^L.
template <class F>
__global__ void __launch_bounds__(256)
k_ring_divlin(const uint32_t *__restrict__ in, uint32_t in_stride, uint32_t len, const uint32_t *__restrict__ zs, fp zmul,
              uint32_t *__restrict__ out, uint32_t out_stride) {
  __shared__ uint32_t sh[256 * 8];
  const uint32_t p = blockIdx.x, t = threadIdx.x, L = (len + 255) / 256;
  const uint32_t *c = in + (size_t)p * in_stride * 8;
  uint32_t *o = out + (size_t)p * out_stride * 8;
  const fp z = fp_mul<F>(load_fp(zs + (size_t)p * 8), zmul);
  uint32_t lo = t * L, hi = lo + L; if (hi > len) hi = len; if (lo > len) lo = len;
  fp h = fp_zero();
  for (uint32_t i = hi; i > lo; i--) h = fp_add<F>(fp_mul<F>(h, z), load_fp(c + (size_t)(i - 1) * 8));
  // carry[t] = sum_{t' > t} h_{t'} z^(L (t' - t - 1)):  start from h_{t+1}, then Hillis-Steele with z^(L d)
  store_fp(sh + t * 8, h);
  __syncthreads();
  fp A = t + 1 < 256 ? load_fp(sh + (t + 1) * 8) : fp_zero();
  fp mul = fp_pow_u32<F>(z, L);
  for (uint32_t d = 1; d < 256; d <<= 1) {
    __syncthreads();
    store_fp(sh + t * 8, A);
    __syncthreads();
    if (t + d < 256) A = fp_add<F>(A, fp_mul<F>(mul, load_fp(sh + (t + d) * 8)));
    mul = fp_sqr<F>(mul);
  }
  for (uint32_t i = hi; i > lo; i--) {
    A = fp_add<F>(load_fp(c + (size_t)(i - 1) * 8), fp_mul<F>(A, z));
    if (i >= 2) store_fp(o + (size_t)(i - 2) * 8, A);
  }
}

// Witness columns from their sparse description (A.7 step 1): pos = sorted rows with bit 1 (<= 256 per proof,
// padded), vals = the cnt+1 successive accumulator points (x | y), kidx = the signer's row, zk = the values of the
// last 3 rows of each column (proof x 4 x 3; nullptr: zeros).  cols[proof] = bits | ip | ax | ay (4 x N evaluations).
template <class F>
__global__ void __launch_bounds__(256)
k_ring_witness_cols(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ kidx,
                    const uint32_t *__restrict__ vals, const uint32_t *__restrict__ zk, uint32_t N, uint32_t cap, uint32_t *__restrict__ cols) {
  __shared__ uint32_t sp[256];
  const uint32_t p = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  sp[threadIdx.x] = pos[(size_t)p * 256 + threadIdx.x];
  __syncthreads();
  if (i >= N) return;
  const uint32_t m = cnt[p], ki = kidx[p];
  uint32_t lo = 0, hi = m;                               // r = number of pos < i
  while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (sp[mid] < i) lo = mid + 1; else hi = mid; }
  const bool bit = lo < m && sp[lo] == i;
  const fp one = fp_one<F>(), zero = fp_zero();
  uint32_t *c = cols + (size_t)p * 4 * N * 8;
  if (i >= cap) {                                          // the 3 zero-knowledge rows: random when hiding, else zero
    for (int col = 0; col < 4; col++)
      store_fp(c + ((size_t)col * N + i) * 8, zk ? load_fp(zk + (((size_t)p * 4 + col) * 3 + (i - cap)) * 8) : zero);
    return;
  }
  store_fp(c + (size_t)i * 8, bit ? one : zero);
  store_fp(c + ((size_t)N + i) * 8, i > ki ? one : zero);
  const uint32_t *v = vals + ((size_t)p * 257 + lo) * 16;
  store_fp(c + ((size_t)2 * N + i) * 8, load_fp(v));
  store_fp(c + ((size_t)3 * N + i) * 8, load_fp(v + 8));
}
// RingProver round 0 on the device (A.7 step 1; w3f-ring-proof `PiopProver::build`): the witness in sparse form.  One lane per
// proof.  Rows with bit 1: the signer's key, then the set bits of the blinding (row keyset + i).  The accumulator column only
// changes there: acc_0 = seed, acc_{j+1} = acc_j + point[pos_j] on the suite's twisted-Edwards curve (mixed additions, the ring's
// points tabulated as te_pre at index time), all m + 1 <= 255 partial sums normalised with ONE inversion (Montgomery's trick over
// the lane's own chain; the extended points and prefix products live in `scratch`, 257 x 160 bytes per proof).  Written: the row
// list and count, the affine partial sums (k_ring_witness_cols expands them into the columns), the four sparse vectors of the
// Lagrange-basis commitments (bits | index polynomial | acc_x | acc_y: base index + PLAIN scalar), and per proof the result
// and the instance (= result - seed) for the transcript.  Until round 3 this ran on host threads (<= 254 additions + 2
// inversions per proof): 4 ms of two cores per 128-proof chunk, the largest host item of a rank of an 8-GPU node.
template <class S>
__global__ void __launch_bounds__(64)
k_ring_witness_acc(const te_pre *__restrict__ points, const uint32_t *__restrict__ kidx, const uint8_t *__restrict__ blind, uint32_t n, uint32_t keyset,
                   uint32_t L, uint32_t N, uint32_t cap, fp seedx, fp seedy, const uint32_t *__restrict__ zk, uint32_t *__restrict__ scratch,
                   uint32_t *__restrict__ pos, uint32_t *__restrict__ cnt, uint32_t *__restrict__ vals, uint32_t *__restrict__ sc, uint32_t *__restrict__ bi,
                   uint32_t *__restrict__ out) {
  // FOUR lanes per proof (te_quad.h: lane jc of the quad holds coordinate jc of the accumulator): a mixed addition is two rounds
  // of multiplications instead of eight sequential ones, the backward pass two rounds per row instead of four multiplications.
  using F = typename S::Fq;
  constexpr uint32_t MP = 264;
  const uint32_t lane = threadIdx.x & 63, jc = lane & 3;
  uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const bool live = p < n;
  if (!live) p = n - 1;                                                        // (whole quads stay in step for the cross-lane moves; they write nothing)
  uint32_t *ppos = pos + (size_t)p * 256;
  const uint32_t ki = kidx[p];
  // the row list (every lane of the quad derives it; lane 0 writes it)
  uint32_t m = 1;
  for (uint32_t i = 0; i < L; i++) if ((blind[32 * (size_t)p + (i >> 3)] >> (i & 7)) & 1) m++;
  if (live && jc == 0) {
    uint32_t w = 0; ppos[w++] = ki;
    for (uint32_t i = 0; i < L; i++) if ((blind[32 * (size_t)p + (i >> 3)] >> (i & 7)) & 1) ppos[w++] = keyset + i;
    for (uint32_t j = w; j < 256; j++) ppos[j] = 0xffffffffu;
    cnt[p] = m;
  }
  // pass 1: the chain.  Scratch entry i of the proof: x | y | z of acc_i | prefix product z_0 .. z_{i-1}; lane jc stores word block jc
  // (lane 2 carries T, which is not kept: it stores the prefix product instead)
  uint32_t *sp = scratch + (size_t)p * 257 * 32;
  fp acc = jc == 0 ? seedx : jc == 1 ? seedy : jc == 2 ? fp_mul<F>(seedx, seedy) : fp_one<F>();
  fp run = fp_one<F>();                                                        // (meaningful on lane 3, mirrored to lane 2 for the store)
  uint32_t next_row = ki, bit = 0;
#pragma unroll 1
  for (uint32_t j = 0; j <= m; j++) {
    const fp z = qperm<3, 3, 3, 3>(acc);
    if (live) store_fp(sp + (size_t)j * 32 + 8 * (jc == 2 ? 3 : jc == 3 ? 2 : jc), jc == 2 ? run : acc);   // x, y, [z at block 2 from lane 3], run at block 3 from lane 2
    run = fp_mul<F>(run, z);
    if (j < m) {
      const te_pre q = points[next_row];
      acc = q_madd<S>(acc, q.x, q.y, q.k, jc);
      while (bit < L && !((blind[32 * (size_t)p + (bit >> 3)] >> (bit & 7)) & 1)) bit++;     // the next set bit of the blinding
      next_row = keyset + bit; bit++;
    }
  }
  fp inv = fp_inv_nf<F>(run);                                                   // (binary GCD, fp256.h: the fixed power was ~30 % of this kernel)
  // pass 2 (backwards): affine coordinates (Montgomery form) and, one row behind, the sparse scalars of the two accumulator columns
  uint32_t *v = vals + (size_t)p * 257 * 16;
  uint32_t *b = bi + (size_t)p * 4 * MP; uint32_t *qv = sc + (size_t)p * 4 * MP * 8;
  fp prev = fp_zero(), res = fp_zero();                                        // lane 0 / 1: x / y of row i + 1; of the last row (the result)
#pragma unroll 1
  for (uint32_t i = m + 1; i-- > 0;) {
    const uint32_t *e = sp + (size_t)i * 32;
    // round 1: lane 3: zi = inv * pre_i; lane 2: inv * z_i (the next inv); lanes 0, 1: from_mont of the previous row's difference
    const fp z = load_fp(e + 16), pre = load_fp(e + 24), xy = load_fp(e + 8 * (jc & 1));
    fp one_plain = fp_zero(); one_plain.v[0] = 1;
    const fp a1 = jc >= 2 ? inv : fp_zero(), b1 = jc == 3 ? pre : z;
    const fp r1 = fp_mul<F>(a1, b1);
    const fp zi = qperm<3, 3, 3, 3>(r1);
    inv = qperm<2, 2, 2, 2>(r1);
    // round 2: lanes 0, 1: x_i zi, y_i zi
    const fp aff = fp_mul<F>(xy, zi);
    if (live && jc < 2) store_fp(v + (size_t)i * 16 + 8 * jc, aff);
    if (i < m) {                                                               // scalar of row i: from_mont(v_i - v_{i+1}), base N + pos_i + 1
      const fp d = fp_from_mont<F>(fp_sub<F>(aff, prev));
      if (live && jc < 2) store_fp(qv + (size_t)((2 + jc) * MP + i) * 8, d);
    }
    if (i == m) res = aff;
    prev = aff;
  }
  // the rest is cheap and done by lane 0 of the quad (result coordinates from the registers of lanes 0 and 1, not through memory)
  const fp resx = qperm<0, 0, 0, 0>(res), resy = qperm<1, 1, 1, 1>(res);
  if (!live || jc != 0) return;
  // instance = result - seed  (-(x, y) = (-x, y))
  te_ext r; r.x = resx; r.y = resy; r.t = fp_mul<F>(resx, resy); r.z = fp_one<F>();
  const te_aff inst = te_to_aff<S>(te_madd<S>(r, te_make_pre<S>(fp_neg<F>(seedx), seedy)));
  uint32_t *o = out + (size_t)p * 32;
  store_fp(o, resx); store_fp(o + 8, resy); store_fp(o + 16, inst.x); store_fp(o + 24, inst.y);
  // sparse vectors: bits | ip | ax | ay   (the host zeroed both arrays: unused entries are scalar 0 at base 0)
  fp one_plain = fp_zero(); one_plain.v[0] = 1;
  const fp minus1 = fp_from_mont<F>(fp_neg<F>(fp_one<F>()));
  for (uint32_t j = 0; j < m; j++) { b[j] = ppos[j]; store_fp(qv + (size_t)j * 8, one_plain); }
  b[MP] = N + cap; store_fp(qv + (size_t)MP * 8, one_plain);
  b[MP + 1] = N + ki + 1; store_fp(qv + (size_t)(MP + 1) * 8, minus1);
  for (uint32_t j = 0; j < m; j++) b[2 * MP + j] = b[3 * MP + j] = N + ppos[j] + 1;
  b[2 * MP + m] = b[3 * MP + m] = N + cap;
  store_fp(qv + (size_t)(2 * MP + m) * 8, fp_from_mont<F>(resx)); store_fp(qv + (size_t)(3 * MP + m) * 8, fp_from_mont<F>(resy));
  if (zk) for (uint32_t col = 0; col < 4; col++) for (uint32_t j = 0; j < 3; j++) {     // + zk_j * L_{cap+j}(tau) G
    b[col * MP + m + 1 + j] = cap + j;
    store_fp(qv + (size_t)(col * MP + m + 1 + j) * 8, fp_from_mont<F>(load_fp(zk + (((size_t)p * 4 + col) * 3 + j) * 8)));
  }
}
template <class S>
__global__ void k_ring_points_pre(const uint32_t *__restrict__ xy_mont, uint32_t n, te_pre *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = te_make_pre<S>(load_fp(xy_mont + (size_t)i * 16), load_fp(xy_mont + (size_t)i * 16 + 8));
}
template <class F>
__global__ void k_set_diag(uint32_t *__restrict__ mat, uint32_t n, uint32_t rows, uint32_t col0) {   // row i of the tile = unit vector e_(col0 + i)
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) store_fp(mat + ((size_t)i * n + col0 + i) * 8, fp_one<F>());
}

// ------------------------------------------------------------------------------------------------
// SHAKE128 + ark-transcript

struct ArkTranscript {
  HostShake128 h; bool has_len = false; uint32_t len = 0;
  void write(const void *d, size_t n) { h.update(d, n); len = (has_len ? len : 0) + (uint32_t)n; has_len = true; }
  void separate() { if (has_len) { uint8_t b[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len}; h.update(b, 4); has_len = false; } }
  void label(const char *l) { separate(); write(l, strlen(l)); separate(); }
  void label(const uint8_t *l, size_t n) { separate(); write(l, n); separate(); }
  void append(const std::vector<uint8_t> &d) { separate(); write(d.data(), d.size()); separate(); }
  void challenge48(const char *l, uint8_t out[48]) { label(l); write("challenge", 9); h.squeeze_copy(out, 48); separate(); }
};

// ------------------------------------------------------------------------------------------------
// suite-generic host helpers

template <class S, class G> struct RingTypes {
  using Fr = HostField<typename S::Fq>;              // Fr(pairing curve) = Fq(TE curve)
  using Te = HostTe<S>;
  using HG = HostG1<G>;
  using FqN = HostFieldN<typename G::Fq>;
  static constexpr int FQB = G::Fq::N * 4;            // bytes of a G1 coordinate
};

// host-side data parallelism over the proofs of a chunk: parallel_for of host_pool.h (one bounded persistent pool per process)

// int_BE(48 bytes) mod r, Montgomery form
// (hi 2^256 + lo) R = (hi R) R^2 / R + lo R: three Montgomery products -- sixteen challenges per proof on the prover AND the verifier;
// the byte-by-byte Horner form this replaces cost 96 products, 3.8 us, i.e. 60 us of a core per proof of the ~80 a proof took)
template <class F> static H256 fr_from_be48(const uint8_t b[48]) {
  using Fr = HostField<F>;
  H256 hi = {{0, 0, 0, 0}}, lo;
  for (int i = 0; i < 2; i++) { uint64_t v; memcpy(&v, b + 8 * i, 8); hi.l[1 - i] = __builtin_bswap64(v); }
  for (int i = 0; i < 4; i++) { uint64_t v; memcpy(&v, b + 16 + 8 * i, 8); lo.l[3 - i] = __builtin_bswap64(v); }
  static const H256 r2 = Fr::r2();
  return Fr::add(Fr::mul(Fr::mul(hi, r2), r2), Fr::mul(lo, r2));    // (mul reduces any 256-bit left operand: its result is below 2 p before the final subtraction)
}
template <class F> static H256 fr_pow(H256 a, uint64_t e) {
  using Fr = HostField<F>; H256 r = Fr::one();
  while (e) { if (e & 1) r = Fr::mul(r, a); a = Fr::sqr(a); e >>= 1; }
  return r;
}
template <class F> static H256 fr_small(uint64_t v) { return HostField<F>::to_mont(H256{{v, 0, 0, 0}}); }

struct G1Aff { uint8_t xy[96]; bool inf; };           // canonical LE x || y (FQB bytes each)

// ark-serialize G1 encodings: BLS12-381 zcash big-endian; BN254 arkworks little-endian
template <class G> static void g1_encode(const G1Aff &p, bool compressed, std::vector<uint8_t> &out) {
  constexpr int B = G::Fq::N * 4;
  using FqN = HostFieldN<typename G::Fq>;
  typename FqN::El y, half = FqN::from32(G::Fq::HALF), t;
  memcpy(y.l, p.xy + B, B);
  bool big = !p.inf && FqN::subb(t, half, y) != 0;    // y > (p-1)/2
  size_t o = out.size();
  if (B == 48) {                                      // zcash
    out.resize(o + (compressed ? B : 2 * B), 0);
    if (p.inf) { out[o] = compressed ? 0xC0 : 0x40; return; }
    for (int i = 0; i < B; i++) out[o + i] = p.xy[B - 1 - i];
    if (compressed) { out[o] |= 0x80; if (big) out[o] |= 0x20; }
    else for (int i = 0; i < B; i++) out[o + B + i] = p.xy[2 * B - 1 - i];
  } else {
    out.resize(o + (compressed ? B : 2 * B), 0);
    if (p.inf) { out[out.size() - 1] |= 0x40; return; }
    memcpy(&out[o], p.xy, compressed ? B : 2 * B);
    if (big) out[out.size() - 1] |= 0x80;
  }
}

}  // namespace avrf

using namespace avrf;

// ------------------------------------------------------------------------------------------------
// handles

// The table of all multiples of an SRS (msm.h G1DirectTable: 164 GB for the 6 145 powers of a ring-1024 setup at c = 15) is a property
// of (device, SRS): every setup over the same SRS on a device -- the four contexts bench.py proves with, their second lanes --
// shares ONE through this registry; the last setup to go frees it.
struct DirectEntry {
  G1DirectTable t; int device = 0, kind = 0; std::vector<uint8_t> srs_key;   // kind 0: the SRS powers, 1: the witness bases derived from them
  ~DirectEntry() { if (t.d) { (void)hipSetDevice(device); free_g1_direct_table(&t); } }
};
static std::mutex g_direct_mu;
static std::vector<std::weak_ptr<DirectEntry>> g_direct;

struct avrf_ring_setup {
  avrf_ctx *ctx; int suite; int curve; hipStream_t stream; int device;   // curve: pairing curve of the suite (0 BLS12-381, 1 BN254)
  size_t N, cap, keyset, L, n_srs;                    // n_srs = 0: verifier-only setup (PcsVerifierParams), no SRS on the device
  uint32_t *d_srs = nullptr;                          // n_srs Montgomery affine points
  uint32_t *d_srs_table = nullptr; int table_c = 0, table_nwin = 0;   // fixed-base window table over the SRS (batched commits)
  std::shared_ptr<DirectEntry> direct, direct_wit; bool direct_tried = false;   // the tables of all multiples (SRS powers; witness bases), built on first use
  uint32_t *d_wit_bases = nullptr;                    // the 2N + 1 witness bases the tables are built over
  int wit_c = 0, wit_nwin = 0;                        // window width of the witness table (sparse MSMs: few entries, small buckets)
  uint32_t *d_wit_table = nullptr;                    // same over [L_i(tau) G, i < N | prefix sums PS_k = sum_{i<k} L_i(tau) G, k <= N] (witness commits)
  void *host_lines = nullptr; void (*host_lines_free)(void *) = nullptr;   // host Miller-loop line tables of (g2, tau g2), built on first use
  G1Aff g1_0;                                         // powers_in_g1[0]
  std::vector<uint8_t> g2_raw;                        // powers_in_g2[0..2] exactly as in the SRS file
  std::vector<uint8_t> g1_raw;                        // the n_srs powers_in_g1 this setup keeps, serialize_uncompressed encoding
  std::vector<uint8_t> lag_raw;                       // L_i(tau) G, i < N (RingBuilderPcsParams), same encoding; filled by ensure_lagrange
  H256 w, w4;                                         // domain generators (Montgomery)
  uint32_t *d_tw_n = nullptr, *d_tw_n_inv = nullptr, *d_tw_4n = nullptr, *d_tw_4n_inv = nullptr;
  H256 ninv, n4inv;
  std::vector<std::pair<H256, H256>> h_pows;          // 2^i * BLINDING_BASE, affine Montgomery
  uint32_t *d_buf = nullptr; size_t buf_cap = 0;      // scratch for NTT batches / MSM scalars
  uint32_t *d_l4 = nullptr;                           // L_first | L_last evaluated on the 4N domain (2 x 4N)
  // per-chunk scratch: 0 witness evaluations (4 x 4N; later the opening quotients), 1 aggregated constraints (4N; later
  // the aggregated opening polynomial), 2 coefficients (4 x N), 3 parameter block, 4 quotient, 5 linearisation
  uint32_t *d_scr[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scr_cap[6] = {0, 0, 0, 0, 0, 0};
  MsmWorkspace ws;
  PairingTables ptab; bool ptab_ready = false;        // line tables of (g2, tau g2) for the device pairing checks (built on first use)
  // second lane for avrf_ring_prove: a shallow copy (same SRS tables, twiddles, constants) with its own stream, scratch
  // and MSM workspace, so two chunks of proofs are in flight and one hides the other's host rounds
  avrf_ring_setup *lane1 = nullptr;
};
struct avrf_ring_key {
  avrf_ring_setup *setup;
  size_t n_keys;
  std::vector<std::pair<H256, H256>> points;          // cap-1 points, affine Montgomery
  std::vector<H256> px, py, sel;                      // evaluations (N each, Montgomery)
  std::vector<H256> px_poly, py_poly, sel_poly;       // coefficients
  std::vector<H256> px4, py4, sel4;                   // evaluations on the 4N domain
  G1Aff C[3];
  uint32_t *d_fixed4 = nullptr;                       // px4 | py4 | sel4 on the device (3 x 4N)
  uint32_t *d_fixed_coef = nullptr;                   // px | py | sel coefficients on the device (3 x N)
  void *d_points_pre = nullptr;                       // the ring's points (keys, padding, blinding-base powers) as te_pre {x, y, d x y}: k_ring_witness_acc
};

// VerifierKeyBuilder (src/ring.rs:539-637): the ring commitment under construction
struct avrf_ring_vk_builder {
  avrf_ring_setup *setup;
  size_t curr = 0;                                    // keys appended so far
  G1Aff C[3];                                         // commitments to the x, y and selector columns of the partial ring
};

extern "C" hipStream_t avrf_ctx_stream_(avrf_ctx *c);
extern "C" int avrf_ctx_busy_(avrf_ctx *c);     // a three-call batch run is open on the context (capi.hip): its stream is taken
extern "C" int avrf_ctx_suite_(avrf_ctx *c);
extern "C" int avrf_ctx_device_(avrf_ctx *c);

namespace {

template <class S, class G> struct Ring {
  using T = RingTypes<S, G>;
  using Fr = typename T::Fr; using Te = typename T::Te;
  using F = typename S::Fq;
  static constexpr int FQB = T::FQB;

  static void ensure_buf(avrf_ring_setup *su, size_t bytes) {
    if (bytes <= su->buf_cap) return;
    if (su->d_buf) HIP_CHECK(hipFree(su->d_buf));
    HIP_CHECK(hipMalloc(&su->d_buf, bytes)); su->buf_cap = bytes;
  }
  static uint32_t *make_twiddles(H256 w, size_t n) {
    std::vector<H256> tw(n / 2);
    H256 x = Fr::one();
    for (size_t i = 0; i < n / 2; i++) { tw[i] = x; x = Fr::mul(x, w); }
    uint32_t *d; HIP_CHECK(hipMalloc(&d, (n / 2) * 32)); HIP_CHECK(hipMemcpy(d, tw.data(), (n / 2) * 32, hipMemcpyHostToDevice));
    return d;
  }
  // in-place (i)NTT of `batch` vectors of size n held on the host, via the device
  static void ntt(avrf_ring_setup *su, std::vector<H256> &v, size_t n, size_t batch, bool inverse) {
    ensure_buf(su, n * batch * 32);
    HIP_CHECK(hipMemcpyAsync(su->d_buf, v.data(), n * batch * 32, hipMemcpyHostToDevice, su->stream));
    const uint32_t *tw = n == su->N ? (inverse ? su->d_tw_n_inv : su->d_tw_n) : (inverse ? su->d_tw_4n_inv : su->d_tw_4n);
    fp sc; H256 k = n == su->N ? su->ninv : su->n4inv; memcpy(sc.v, k.l, 32);
    ntt_launch<F>(su->d_buf, (uint32_t)n, tw, (uint32_t)batch, inverse ? &sc : nullptr, su->stream);
    HIP_CHECK(hipMemcpyAsync(v.data(), su->d_buf, n * batch * 32, hipMemcpyDeviceToHost, su->stream));
    HIP_CHECK(hipStreamSynchronize(su->stream));
  }
  // KZG commit: sum coeffs[i] * powers_in_g1[i] on the device
  static G1Aff commit(avrf_ring_setup *su, const H256 *coeffs_mont, size_t n) {
    std::vector<H256> plain(n);
    for (size_t i = 0; i < n; i++) plain[i] = Fr::from_mont(coeffs_mont[i]);
    ensure_buf(su, n * 32);
    HIP_CHECK(hipMemcpyAsync(su->d_buf, plain.data(), n * 32, hipMemcpyHostToDevice, su->stream));
    G1Aff r; memset(&r, 0, sizeof r);
    msm_g1_device(su->curve, su->d_srs, su->d_buf, n, su->ws, su->stream, r.xy);
    r.inf = true; for (int i = 0; i < 2 * FQB; i++) if (r.xy[i]) r.inf = false;
    return r;
  }
  // `batch` KZG commits in one launch chain: coeffs = batch vectors of length n (Montgomery; pad with zeros)
  static void commit_batch(avrf_ring_setup *su, const H256 *coeffs_mont, size_t n, size_t batch, G1Aff *out) {
    std::vector<H256> plain(n * batch);
    for (size_t i = 0; i < n * batch; i++) plain[i] = Fr::from_mont(coeffs_mont[i]);
    ensure_buf(su, n * batch * 32);
    HIP_CHECK(hipMemcpyAsync(su->d_buf, plain.data(), n * batch * 32, hipMemcpyHostToDevice, su->stream));
    std::vector<uint8_t> xy(batch * 2 * FQB);
    msm_g1_device(su->curve, su->d_srs, su->d_buf, n, su->ws, su->stream, xy.data(), batch);
    for (size_t b = 0; b < batch; b++) {
      memset(&out[b], 0, sizeof(G1Aff)); memcpy(out[b].xy, &xy[b * 2 * FQB], 2 * FQB);
      out[b].inf = true; for (int i = 0; i < 2 * FQB; i++) if (out[b].xy[i]) out[b].inf = false;
    }
  }
  static H256 poly_eval(const std::vector<H256> &c, const H256 &x) {
    H256 acc = {{0, 0, 0, 0}};
    for (size_t i = c.size(); i-- > 0;) acc = Fr::add(Fr::mul(acc, x), c[i]);
    return acc;
  }
  static std::vector<H256> div_linear(const std::vector<H256> &c, const H256 &z) {   // (c(X) - c(z)) / (X - z)
    std::vector<H256> out(c.size() - 1); H256 acc = {{0, 0, 0, 0}};
    for (size_t i = c.size() - 1; i >= 1; i--) { acc = Fr::add(c[i], Fr::mul(acc, z)); out[i - 1] = acc; }
    return out;
  }
  static H256 challenge(ArkTranscript &t, const char *l) { uint8_t b[48]; t.challenge48(l, b); return fr_from_be48<F>(b); }
  static void push_le32(std::vector<uint8_t> &o, const H256 &mont) { H256 p = Fr::from_mont(mont); size_t k = o.size(); o.resize(k + 32); memcpy(&o[k], p.l, 32); }

  // ---- setup: parse `URS { powers_in_g1, powers_in_g2 }` (serialize_uncompressed), src/ring.rs:380-393,1412-1421
  static int setup_load(avrf_ctx *ctx, const uint8_t *srs, size_t len, size_t ring_size, avrf_ring_setup **out) {
    const size_t L = S::Fr::BITS;
    size_t need = ring_size + 4 + L, N = 1; while (N < need) N <<= 1;      // src/ring.rs:810-821
    const size_t pcs = 3 * N + 1;
    if (len < 8) return AVRF_INVALID_DATA;
    uint64_t cnt; memcpy(&cnt, srs, 8);
    const size_t e1 = 2 * FQB, e2 = 4 * FQB;
    // serialize_compressed form of the same object (RingSetup / PcsParams, src/ring.rs:484-521): decompress on the host pool
    // and continue with the equivalent uncompressed bytes
    std::vector<uint8_t> unc;
    if (cnt <= (len - 8) / FQB && len >= 8 + cnt * FQB + 8) {
      uint64_t c2; memcpy(&c2, srs + 8 + cnt * FQB, 8);
      if (len == 8 + cnt * FQB + 8 + c2 * 2 * FQB && len != 8 + cnt * e1 + 8 + c2 * e2) {
        if (cnt < pcs || c2 < 2) return AVRF_RING_CAPACITY_EXCEEDED;
        using HP = HostPairing<G>;
        std::vector<G1Aff> pts(pcs);
        std::atomic<int> bad{0};
        parallel_for(pcs, [&](size_t i) { if (!g1_decompress(srs + 8 + i * FQB, &pts[i])) bad = 1; });
        if (bad) return AVRF_INVALID_DATA;
        unc.resize(8);
        { uint64_t v = pcs; memcpy(unc.data(), &v, 8); }
        for (size_t i = 0; i < pcs; i++) g1_encode<G>(pts[i], false, unc);
        { uint64_t v = 2; size_t o = unc.size(); unc.resize(o + 8); memcpy(&unc[o], &v, 8); }
        for (int i = 0; i < 2; i++) {
          typename HP::G2 q;
          if (!HP::g2_decode_compressed(srs + 8 + cnt * FQB + 8 + (size_t)i * 2 * FQB, &q)) return AVRF_INVALID_DATA;
          size_t o = unc.size(); unc.resize(o + e2); HP::g2_encode(q, &unc[o]);
        }
        srs = unc.data(); len = unc.size(); memcpy(&cnt, srs, 8);
      }
    }
    if (len < 8 + cnt * e1 + 8) return AVRF_INVALID_DATA;
    uint64_t cnt2; memcpy(&cnt2, srs + 8 + cnt * e1, 8);
    if (len != 8 + cnt * e1 + 8 + cnt2 * e2) return AVRF_INVALID_DATA;
    if (cnt < pcs || cnt2 < 2) return AVRF_RING_CAPACITY_EXCEEDED;      // src/ring.rs:382-384
    avrf_ring_setup *su = new avrf_ring_setup();
    su->ctx = ctx; su->suite = S::ID; su->curve = pairing_curve_of(S::ID); su->stream = avrf_ctx_stream_(ctx); su->device = avrf_ctx_device_(ctx);
    su->N = N; su->cap = N - 3; su->L = L; su->keyset = su->cap - L - 1; su->n_srs = pcs;
    std::vector<uint8_t> le(pcs * e1);
    for (size_t i = 0; i < pcs; i++) {
      const uint8_t *p = srs + 8 + i * e1; uint8_t *q = &le[i * e1];
      if (FQB == 48) { bool inf = p[0] & 0x40; for (int k = 0; k < FQB; k++) { q[k] = inf ? 0 : p[FQB - 1 - k]; q[FQB + k] = inf ? 0 : p[2 * FQB - 1 - k]; } }
      else { bool inf = p[2 * FQB - 1] & 0x40; memcpy(q, p, e1); q[e1 - 1] &= 0x3f; if (inf) memset(q, 0, e1); }
    }
    memcpy(su->g1_0.xy, le.data(), e1); su->g1_0.inf = false;
    su->g2_raw.assign(srs + 8 + cnt * e1 + 8, srs + 8 + cnt * e1 + 8 + 2 * e2);
    su->g1_raw.assign(srs + 8, srs + 8 + pcs * e1);
    HIP_CHECK(hipSetDevice(su->device));
    uint8_t *d_le; uint32_t *d_flag; uint32_t flag = 0;
    HIP_CHECK(hipMalloc(&d_le, le.size())); HIP_CHECK(hipMalloc(&d_flag, 4)); HIP_CHECK(hipMalloc(&su->d_srs, pcs * e1));
    HIP_CHECK(hipMemcpy(d_le, le.data(), le.size(), hipMemcpyHostToDevice)); HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream));   // (the flag reset in stream order with the kernel: the null stream does not order with a non-blocking one)
    launch_g1_bases(su->curve, d_le, pcs, su->d_srs, d_flag, su->stream);
    HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, su->stream)); HIP_CHECK(hipStreamSynchronize(su->stream));
    HIP_CHECK(hipFree(d_le)); HIP_CHECK(hipFree(d_flag));
    if (flag) { HIP_CHECK(hipFree(su->d_srs)); delete su; return AVRF_INVALID_DATA; }
    {  // fixed-base window table: T[w][i] = 2^(c w) * tau^i G, all windows of a commit then share one bucket set
      // window width from the size of the commitments (3N + 1 coefficients): wider windows save bucket additions (k_accumulate:
      // one per table row and coefficient), narrower ones save bucket REDUCTION (k_bucket_sum + k_wsum: 2^(c-1) buckets per set).
      // Which wins depends on what else runs: ONE context alone prefers narrow windows (ring 1024, N = 2048, BLS12-381:
      // c = 12 / 11 / 10 / 9 -> 11.1 / 11.4 / 11.6 / 10.8 k proofs/s -- its latency-bound reductions sit on the critical path),
      // FOUR contexts sharing the chip, as bench.py and a loaded server run it, hide each other's reductions and pay for
      // additions only: c = 11 / 12 / 13 -> 13.8 / 14.4 / 13.8 k; BN254 ring 4096 (N = 8192): c = 12 / 13 -> 7.39 / 7.56 k
      // (tools/ring4_bench.py with AVRF_RING_TABLE_C swept).  The loaded regime decides: c = log2(3N + 1) rounded down
      // (381-bit curve), one less on the 254-bit curve whose additions are cheaper relative to its reductions.
      { int lg = 0; while (((size_t)2 << lg) <= pcs) lg++;                    // floor(log2(3N + 1)) = log2(N) + 1
        su->table_c = G::Fq::N > 8 ? lg : lg - 1;
        if (su->table_c < 8) su->table_c = 8; if (su->table_c > 13) su->table_c = 13; }
      if (const char *e = getenv("AVRF_RING_TABLE_C")) { int v = atoi(e); if (v >= 4 && v <= 14) su->table_c = v; }
      {
        su->table_nwin = (G::Fr::BITS + 1 + su->table_c - 1) / su->table_c;
        HIP_CHECK(hipMalloc(&su->d_srs_table, (size_t)su->table_nwin * pcs * e1));
        build_g1_table(su->curve, su->d_srs, pcs, su->table_c, su->table_nwin, su->d_srs_table, su->stream);
        HIP_CHECK(hipStreamSynchronize(su->stream));
      }
    }
    // domain
    H256 root = Fr::from32(G::ROOT_OF_UNITY);
    int lg = 0; while (((size_t)1 << lg) < N) lg++;
    H256 w4 = root; for (int i = 0; i < G::TWO_ADICITY - (lg + 2); i++) w4 = Fr::sqr(w4);
    su->w4 = w4; su->w = Fr::sqr(Fr::sqr(w4));
    su->d_tw_n = make_twiddles(su->w, N); su->d_tw_n_inv = make_twiddles(Fr::inv(su->w), N);
    su->d_tw_4n = make_twiddles(su->w4, 4 * N); su->d_tw_4n_inv = make_twiddles(Fr::inv(su->w4), 4 * N);
    su->ninv = Fr::inv(fr_small<F>(N)); su->n4inv = Fr::inv(fr_small<F>(4 * N));
    {  // Lagrange basis polynomials of rows 0 and cap-1, evaluated on the 4N domain (shared by every proof)
      const H256 zero = {{0, 0, 0, 0}};
      std::vector<H256> lfl(2 * N, zero); lfl[0] = Fr::one(); lfl[N + su->cap - 1] = Fr::one();
      ntt(su, lfl, N, 2, true);
      std::vector<H256> l4(2 * 4 * N, zero);
      for (size_t i = 0; i < N; i++) { l4[i] = lfl[i]; l4[4 * N + i] = lfl[N + i]; }
      ntt(su, l4, 4 * N, 2, false);
      HIP_CHECK(hipMalloc(&su->d_l4, l4.size() * 32)); HIP_CHECK(hipMemcpy(su->d_l4, l4.data(), l4.size() * 32, hipMemcpyHostToDevice));

    }
    // 2^i * H  (A.5)
    HostExt h; h.x = Fr::from32(S::B_X); h.y = Fr::from32(S::B_Y); h.t = Fr::mul(h.x, h.y); h.z = Fr::one();
    for (size_t i = 0; i < L; i++) {
      H256 zi = Fr::inv(h.z); su->h_pows.push_back({Fr::mul(h.x, zi), Fr::mul(h.y, zi)});
      h = Te::dbl(h);
    }
    *out = su;
    return AVRF_OK;
  }

  // ---- verifier-only setup (src/ring.rs:466-482 verifier_key_from_commitment: "verifier-only users: no SRS required"):
  // built from PcsVerifierParams = RawKzgVerifierKey { g1, g2, tau_in_g2 } (RingSetup::pcs_verifier_params, src/ring.rs:435)
  // in either ark-serialize mode.  Carries what the verifiers use -- domain, g1, (g2, tau g2) -- and no SRS: index / prove /
  // builder / serialisation of the full setup answer AVRF_SRS_LOOKUP_FAILED on such a handle.
  static int verifier_setup_load(avrf_ctx *ctx, const uint8_t *vp, size_t len, size_t ring_size, avrf_ring_setup **out) {
    using HP = HostPairing<G>;
    const size_t L = S::Fr::BITS, e1 = 2 * FQB, e2 = 4 * FQB;
    size_t need = ring_size + 4 + L, N = 1; while (N < need) N <<= 1;
    G1Aff g1; std::vector<uint8_t> g2(2 * e2);
    if (len == e1 + 2 * e2) {
      g1 = g1_from_raw(vp);
      memcpy(g2.data(), vp + e1, 2 * e2);
      typename HP::G2 q;
      for (int i = 0; i < 2; i++) if (!HP::g2_decode(g2.data() + (size_t)i * e2, &q) || q.inf || !HP::g2_on_twist(q)) return AVRF_INVALID_DATA;
    } else if (len == FQB + 2 * 2 * FQB) {
      if (!g1_decompress(vp, &g1)) return AVRF_INVALID_DATA;
      for (int i = 0; i < 2; i++) {
        typename HP::G2 q;
        if (!HP::g2_decode_compressed(vp + FQB + (size_t)i * 2 * FQB, &q)) return AVRF_INVALID_DATA;
        HP::g2_encode(q, g2.data() + (size_t)i * e2);
      }
    } else return AVRF_INVALID_DATA;
    // g1: in range, on the curve, in the prime-order subgroup, not the point at infinity (the uncompressed form carries both
    // coordinates, so the curve equation is checked here; BN254 G1 has cofactor 1).  The two G2 points are decoded and checked to
    // lie on the twist; like the reference's RingSetup deserialisation they get no G2 subgroup test -- PcsVerifierParams are
    // trusted-setup material published with the ring parameters, not per-proof input (src/ring.rs:466-474).
    if (g1.inf || !g1_on_curve_host(g1) || !g1_in_subgroup_host(g1)) return AVRF_INVALID_DATA;
    avrf_ring_setup *su = new avrf_ring_setup();
    su->ctx = ctx; su->suite = S::ID; su->curve = pairing_curve_of(S::ID); su->stream = avrf_ctx_stream_(ctx); su->device = avrf_ctx_device_(ctx);
    su->N = N; su->cap = N - 3; su->L = L; su->keyset = su->cap - L - 1; su->n_srs = 0;
    su->g1_0 = g1; su->g2_raw = g2;
    H256 root = Fr::from32(G::ROOT_OF_UNITY);
    int lg = 0; while (((size_t)1 << lg) < N) lg++;
    H256 w4 = root; for (int i = 0; i < G::TWO_ADICITY - (lg + 2); i++) w4 = Fr::sqr(w4);
    su->w4 = w4; su->w = Fr::sqr(Fr::sqr(w4));
    su->ninv = Fr::inv(fr_small<F>(N)); su->n4inv = Fr::inv(fr_small<F>(4 * N));
    *out = su;
    return AVRF_OK;
  }
  static int verifier_params_serialize(avrf_ring_setup *su, bool compress, std::vector<uint8_t> &o) {
    using HP = HostPairing<G>;
    g1_encode<G>(su->g1_0, compress, o);
    if (!compress) { o.insert(o.end(), su->g2_raw.begin(), su->g2_raw.end()); return AVRF_OK; }
    for (int i = 0; i < 2; i++) {
      typename HP::G2 q; HP::g2_decode(su->g2_raw.data() + (size_t)i * 4 * FQB, &q);
      size_t at = o.size(); o.resize(at + 2 * FQB); HP::g2_encode_compressed(q, &o[at]);
    }
    return AVRF_OK;
  }

  // ---- RingSetup::from_seed (src/ring.rs:359-366) as the reference derives it: S::Transcript::new(SUITE_ID), absorb_raw(seed),
  // to_rng() (src/utils/transcript.rs:61-92: every draw is the next bytes of the squeeze stream), then Kzg::setup(pcs_domain_size - 1,
  // rng) = w3f-pcs URS::generate: tau = Fr::rand, g1 = G1::rand, g2 = G2::rand, powers tau^i g1 / g2.  **UNPINNED**: the reference holds
  // no vector of a seeded setup; the draw order and the samplers are restated from the published code of ark-ff
  // (Fp::rand: N x next_u64 limbs, top limb masked to the modulus' bits, taken AS the Montgomery representation, rejected if >= p;
  // Fp2: c0 then c1), ark-ec (Projective::rand: x = BaseField::rand, greatest = next_u32's top bit, the larger / smaller root by the
  // field's Ord, times the cofactor) and rand 0.8, and checked against oracle/ring_py.py srs_from_seed, which restates the same.
  struct SeedStream {
    std::vector<uint8_t> buf; size_t pos = 0; uint8_t dig[64]; uint64_t blk = 0;
    const uint8_t *absorbed; size_t absorbed_len;
    void more(size_t n) {
      if constexpr (S::XOF_SHAKE) { HostShake128 h; h.update(absorbed, absorbed_len); buf.resize(pos + n + 4096); h.squeeze_copy(buf.data(), buf.size()); }
      else if constexpr (S::TR_SHA256) { HostSha256 h; h.update(absorbed, absorbed_len); buf.resize(pos + n + 4096); h.squeeze_copy(buf.data(), buf.size()); }
      else {                                                           // DigestXof over SHA-512: d = H(absorbed), block_i = H(d || LE64(i))
        if (!blk && buf.empty()) { HostSha512 h; h.update(absorbed, absorbed_len); h.final(dig); }
        while (buf.size() < pos + n) { HostSha512 h; h.update(dig, 64); uint8_t c[8]; for (int i = 0; i < 8; i++) c[i] = (uint8_t)(blk >> (8 * i)); h.update(c, 8);
          uint8_t o[64]; h.final(o); buf.insert(buf.end(), o, o + 64); blk++; }
      }
    }
    void take(uint8_t *out, size_t n) { if (pos + n > buf.size()) more(n); memcpy(out, buf.data() + pos, n); pos += n; }
    bool boolean() { uint8_t b[4]; take(b, 4); return (b[3] & 0x80) != 0; }                     // (rng.next_u32() as i32) < 0
    template <int L, int BITS> void field(uint64_t *v, const uint64_t *p) {                     // ark-ff Fp::rand: the limbs ARE the Montgomery form
      for (;;) {
        uint8_t b[8 * L]; take(b, sizeof b); memcpy(v, b, sizeof b);
        constexpr int shave = 64 * L - BITS;
        if (shave) v[L - 1] &= ~(uint64_t)0 >> shave;
        bool less = false;
        for (int i = L - 1; i >= 0; i--) if (v[i] != p[i]) { less = v[i] < p[i]; break; }
        if (less) return;
      }
    }
  };
  static int setup_from_seed(avrf_ctx *ctx, const uint8_t seed[32], size_t ring_size, avrf_ring_setup **out) {
    using HP = HostPairing<G>; using HG = typename T::HG;
    std::vector<uint8_t> ab(S::SUITE_ID, S::SUITE_ID + S::SUITE_ID_LEN); ab.insert(ab.end(), seed, seed + 32);
    SeedStream st; st.absorbed = ab.data(); st.absorbed_len = ab.size();
    // tau = Fr::rand: the drawn limbs are tau's Montgomery form
    H256 tau_m; { const H256 pr = Fr::P(); st.template field<4, G::Fr::BITS>(tau_m.l, pr.l); }
    const H256 tau = Fr::from_mont(tau_m);
    uint8_t tau_le[32]; memcpy(tau_le, tau.l, 32);
    // g1 = G1::rand: first x with x^3 + b a square, the root picked by `greatest`, times the cofactor
    std::vector<uint8_t> g1_urs, g2_urs(4 * FQB);
    for (;;) {
      const QEl pq = FqN::P(); QEl xm; st.template field<FqN::L, G::Fq::BITS>(xm.l, pq.l);
      const bool greatest = st.boolean();
      const QEl rhs = FqN::add(FqN::mul(FqN::sqr(xm), xm), FqN::from32(G::B));
      static const QEl e = [] { QEl v = FqN::P(), one = FqN::zero(); one.l[0] = 1; FqN::addc(v, v, one);   // (p + 1) / 4
                                for (int k = 0; k < 2; k++) for (int i = 0; i < FqN::L; i++) v.l[i] = (v.l[i] >> 1) | (i + 1 < FqN::L ? v.l[i + 1] << 63 : 0);
                                return v; }();
      QEl y = FqN::one();
      for (int i = 64 * FqN::L - 1; i >= 0; i--) { y = FqN::sqr(y); if ((e.l[i / 64] >> (i % 64)) & 1) y = FqN::mul(y, rhs); }
      if (!FqN::eq(FqN::sqr(y), rhs)) continue;
      QEl t; const QEl yp = FqN::from_mont(y), half = FqN::from32(G::Fq::HALF);
      const bool is_larger = FqN::subb(t, half, yp) != 0;                                       // y > (p - 1) / 2  <=>  y > -y
      if (is_larger != greatest) y = FqN::neg(y);
      typename HG::Pt pt; pt.x = xm; pt.y = y; pt.zz = FqN::one(); pt.zzz = FqN::one();
      static const uint64_t COF_BLS[2] = {0x8c00aaab0000aaabULL, 0x396c8c005555e156ULL};
      typename HG::Pt acc = HG::identity();
      if (FQB == 48) { for (int i = 127; i >= 0; i--) { acc = HG::dbl(acc); if ((COF_BLS[i / 64] >> (i % 64)) & 1) acc = HG::add(acc, pt); } }
      else acc = pt;                                                                            // BN254: cofactor 1
      G1Aff a; memset(&a, 0, sizeof a); HG::to_affine_bytes(acc, a.xy); a.inf = false;
      g1_encode<G>(a, false, g1_urs);
      break;
    }
    // g2 = G2::rand on the twist: x = (c0, c1), the root by the (c1, c0) order, times the cofactor
    for (;;) {
      const QEl pq = FqN::P(); typename HP::F2 x; st.template field<FqN::L, G::Fq::BITS>(x.a.l, pq.l); st.template field<FqN::L, G::Fq::BITS>(x.b.l, pq.l);
      const bool greatest = st.boolean();
      typename HP::F2 y;
      if (!HP::f2_sqrt(HP::f2_add(HP::f2_mul(HP::f2_sqr(x), x), HP::twist_b()), &y)) continue;
      if (HP::f2_is_largest(y) != greatest) y = HP::f2_neg(y);
      typename HP::G2 q; q.x = x; q.y = y; q.inf = false;
      static const uint64_t COF2_BLS[8] = {0xcf1c38e31c7238e5ULL, 0x1616ec6e786f0c70ULL, 0x21537e293a6691aeULL, 0xa628f1cb4d9e82efULL,
                                           0xa68a205b2e5a7ddfULL, 0xcd91de4547085abaULL, 0x091d50792876a202ULL, 0x05d543a95414e7f1ULL};
      static const uint64_t COF2_BN[4] = {0x345f2299c0f9fa8dULL, 0x06ceecda572a2489ULL, 0xb85045b68181585eULL, 0x30644e72e131a029ULL};
      const uint64_t *cof = FQB == 48 ? COF2_BLS : COF2_BN; const int cbits = FQB == 48 ? 512 : 256;
      typename HP::G2 acc; acc.inf = true; acc.x = HP::f2_zero(); acc.y = HP::f2_zero();
      for (int i = cbits - 1; i >= 0; i--) { acc = HP::g2_dbl(acc); if ((cof[i / 64] >> (i % 64)) & 1) acc = HP::g2_add(acc, q); }
      HP::g2_encode(acc, g2_urs.data());
      break;
    }
    const size_t n_g1 = avrf_ring_pcs_domain_size(S::ID, ring_size);
    std::vector<uint8_t> srs(8 + n_g1 * 2 * FQB + 8 + 2 * 4 * FQB);
    size_t len = 0;
    if (int e = srs_generate(ctx, tau_le, g1_urs.data(), g2_urs.data(), n_g1, srs.data(), srs.size(), &len)) return e;
    return setup_load(ctx, srs.data(), len, ring_size, out);
  }

  // ---- Kzg::setup (reached from RingSetup::from_rand / from_seed, src/ring.rs:359-374) with the trapdoor and the two
  // generators given explicitly: writes `URS { powers_in_g1: tau^i g1 (i < n_g1), powers_in_g2: [g2, tau g2] }` in the
  // serialize_uncompressed layout that setup_load reads.  The n_g1 fixed-base multiplications are ONE batched table MSM.
  static int srs_generate(avrf_ctx *ctx, const uint8_t tau_le[32], const uint8_t *g1_urs, const uint8_t *g2_urs, size_t n_g1,
                          uint8_t *out, size_t out_cap, size_t *out_len) {
    const size_t e1 = 2 * FQB, e2 = 4 * FQB, need = 8 + n_g1 * e1 + 8 + 2 * e2;
    if (out_len) *out_len = need;
    if (!out || out_cap < need) return AVRF_ERR_BAD_ARG;
    H256 tau = Fr::load_le(tau_le);
    if (Fr::geq_p(tau) || !n_g1) return AVRF_INVALID_DATA;
    hipStream_t stream = avrf_ctx_stream_(ctx);
    HIP_CHECK(hipSetDevice(avrf_ctx_device_(ctx)));
    // g1: URS entry -> canonical little-endian x || y
    uint8_t le[2 * 48];
    if (FQB == 48) { if (g1_urs[0] & 0xC0) return AVRF_INVALID_DATA; for (int k = 0; k < FQB; k++) { le[k] = g1_urs[FQB - 1 - k]; le[FQB + k] = g1_urs[2 * FQB - 1 - k]; } }
    else { if (g1_urs[e1 - 1] & 0x40) return AVRF_INVALID_DATA; memcpy(le, g1_urs, e1); le[e1 - 1] &= 0x3f; }   // 0x80: arkworks' sign-of-y flag
    std::vector<H256> pw(n_g1);                                        // tau^i, plain
    { H256 tm = Fr::to_mont(tau), run = Fr::one(); for (size_t i = 0; i < n_g1; i++) { pw[i] = Fr::from_mont(run); run = Fr::mul(run, tm); } }
    const int c = 4, nwin = (G::Fr::BITS + 1 + c - 1) / c;
    uint8_t *d_le; uint32_t *d_flag, *d_base, *d_table, *d_sc; uint32_t flag = 0;
    HIP_CHECK(hipMalloc(&d_le, e1)); HIP_CHECK(hipMalloc(&d_flag, 4)); HIP_CHECK(hipMalloc(&d_base, e1));
    HIP_CHECK(hipMalloc(&d_table, (size_t)nwin * e1)); HIP_CHECK(hipMalloc(&d_sc, n_g1 * 32));
    HIP_CHECK(hipMemcpy(d_le, le, e1, hipMemcpyHostToDevice)); HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, stream));   // (the flag reset in stream order with the kernel that sets it)
    HIP_CHECK(hipMemcpy(d_sc, pw.data(), n_g1 * 32, hipMemcpyHostToDevice));
    launch_g1_bases(pairing_curve_of(S::ID), d_le, 1, d_base, d_flag, stream);
    HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, stream)); HIP_CHECK(hipStreamSynchronize(stream));
    int st = AVRF_OK;
    std::vector<uint8_t> xy(n_g1 * e1);
    if (flag) st = AVRF_INVALID_DATA;
    else {
      build_g1_table(pairing_curve_of(S::ID), d_base, 1, c, nwin, d_table, stream);
      MsmWorkspace ws;
      msm_g1_fixed_device(pairing_curve_of(S::ID), d_table, c, 1, d_sc, 1, 1, ws, stream, xy.data(), n_g1);
      ws.release();
    }
    HIP_CHECK(hipFree(d_le)); HIP_CHECK(hipFree(d_flag)); HIP_CHECK(hipFree(d_base)); HIP_CHECK(hipFree(d_table)); HIP_CHECK(hipFree(d_sc));
    if (st) return st;
    uint64_t cnt = n_g1; memcpy(out, &cnt, 8);
    {
      std::vector<uint8_t> enc; enc.reserve(n_g1 * e1);
      for (size_t i = 0; i < n_g1; i++) {
        G1Aff a; memset(&a, 0, sizeof a); memcpy(a.xy, &xy[i * e1], e1);
        a.inf = true; for (size_t k = 0; k < e1; k++) if (a.xy[k]) a.inf = false;
        g1_encode<G>(a, false, enc);                                   // serialize_uncompressed (with arkworks' y flag on BN254)
      }
      memcpy(out + 8, enc.data(), n_g1 * e1);
    }
    using HP = HostPairing<G>;
    typename HP::G2 g2; HP::g2_decode(g2_urs, &g2);
    typename HP::G2 tg2 = HP::g2_mul(g2, tau.l);
    uint8_t *o2 = out + 8 + n_g1 * e1; cnt = 2; memcpy(o2, &cnt, 8);
    memcpy(o2 + 8, g2_urs, e2); HP::g2_encode(tg2, o2 + 8 + e2);
    return AVRF_OK;
  }

  // ---- ring_proof::index (A.5): fixed columns and their commitments
  static int index(avrf_ring_setup *su, const uint8_t *pks_xy, size_t n_keys, avrf_ring_key **out) {
    if (n_keys > su->keyset) return AVRF_RING_CAPACITY_EXCEEDED;       // src/ring.rs:400-402
    const size_t N = su->N;
    avrf_ring_key *k = new avrf_ring_key(); k->setup = su; k->n_keys = n_keys;
    H256 padx = Fr::from32(S::PAD_X), pady = Fr::from32(S::PAD_Y);
    for (size_t i = 0; i < n_keys; i++) {
      H256 x = Fr::load_le(pks_xy + 64 * i), y = Fr::load_le(pks_xy + 64 * i + 32);
      if (Fr::geq_p(x) || Fr::geq_p(y)) { delete k; return AVRF_INVALID_DATA; }
      k->points.push_back({Fr::to_mont(x), Fr::to_mont(y)});
    }
    for (size_t i = n_keys; i < su->keyset; i++) k->points.push_back({padx, pady});
    for (auto &p : su->h_pows) k->points.push_back(p);
    H256 zero = {{0, 0, 0, 0}};
    std::vector<H256> cols(3 * N, zero);
    for (size_t i = 0; i < k->points.size(); i++) { cols[i] = k->points[i].first; cols[N + i] = k->points[i].second; }
    for (size_t i = 0; i < su->keyset; i++) cols[2 * N + i] = Fr::one();
    k->px.assign(cols.begin(), cols.begin() + N); k->py.assign(cols.begin() + N, cols.begin() + 2 * N); k->sel.assign(cols.begin() + 2 * N, cols.end());
    ntt(su, cols, N, 3, true);
    k->px_poly.assign(cols.begin(), cols.begin() + N); k->py_poly.assign(cols.begin() + N, cols.begin() + 2 * N); k->sel_poly.assign(cols.begin() + 2 * N, cols.end());
    commit_batch(su, cols.data(), N, 3, k->C);
    HIP_CHECK(hipMalloc(&k->d_fixed_coef, 3 * N * 32)); HIP_CHECK(hipMemcpy(k->d_fixed_coef, cols.data(), 3 * N * 32, hipMemcpyHostToDevice));
    // evaluations of the fixed columns on the 4N domain (shared by every proof over this ring)
    std::vector<H256> e4(3 * 4 * N, zero);
    for (size_t i = 0; i < N; i++) { e4[i] = k->px_poly[i]; e4[4 * N + i] = k->py_poly[i]; e4[8 * N + i] = k->sel_poly[i]; }
    ntt(su, e4, 4 * N, 3, false);
    k->px4.assign(e4.begin(), e4.begin() + 4 * N); k->py4.assign(e4.begin() + 4 * N, e4.begin() + 8 * N); k->sel4.assign(e4.begin() + 8 * N, e4.end());
    HIP_CHECK(hipMalloc(&k->d_fixed4, e4.size() * 32)); HIP_CHECK(hipMemcpy(k->d_fixed4, e4.data(), e4.size() * 32, hipMemcpyHostToDevice));
    {  // the same points as te_pre for the device witness accumulation (k_ring_witness_acc)
      const size_t np = k->points.size();
      std::vector<H256> xy(2 * np);
      for (size_t i = 0; i < np; i++) { xy[2 * i] = k->points[i].first; xy[2 * i + 1] = k->points[i].second; }
      ensure_buf(su, np * 64);
      HIP_CHECK(hipMalloc(&k->d_points_pre, np * sizeof(te_pre)));
      HIP_CHECK(hipMemcpyAsync(su->d_buf, xy.data(), np * 64, hipMemcpyHostToDevice, su->stream));
      hipLaunchKernelGGL(k_ring_points_pre<S>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, su->stream, (const uint32_t *)su->d_buf, (uint32_t)np, (te_pre *)k->d_points_pre);
      HIP_CHECK(hipStreamSynchronize(su->stream));
    }
    *out = k;
    return AVRF_OK;
  }

  static void transcript_prelude(avrf_ring_key *k, ArkTranscript &t) {
    avrf_ring_setup *su = k->setup;
    t.label(S::SUITE_ID, S::SUITE_ID_LEN);
    t.label("vk");
    std::vector<uint8_t> vk; g1_encode<G>(su->g1_0, false, vk);
    vk.insert(vk.end(), su->g2_raw.begin(), su->g2_raw.end());
    for (int i = 0; i < 3; i++) g1_encode<G>(k->C[i], false, vk);
    t.append(vk);
  }

  // ---- VerifierKeyBuilder (src/ring.rs:539-637): start from the all-padding ring, then every appended key replaces a
  // padding point: C_x += (x - x_pad) L_i(tau) G, C_y likewise -- two sparse MSMs over the Lagrange-basis table per append
  static G1Aff g1_add_aff(const G1Aff &a, const G1Aff &b) {
    using HG = typename T::HG; using FqN = typename T::FqN;
    auto lift = [](const G1Aff &p) { typename HG::Pt q = HG::identity();
      if (!p.inf) { typename FqN::El x, y; memcpy(x.l, p.xy, FQB); memcpy(y.l, p.xy + FQB, FQB); q.x = FqN::to_mont(x); q.y = FqN::to_mont(y); q.zz = FqN::one(); q.zzz = FqN::one(); }
      return q; };
    typename HG::Pt r = HG::add(lift(a), lift(b));
    G1Aff o; memset(&o, 0, sizeof o); HG::to_affine_bytes(r, o.xy);
    o.inf = true; for (int i = 0; i < 2 * FQB; i++) if (o.xy[i]) o.inf = false;
    return o;
  }
  static int builder_new(avrf_ring_setup *su, avrf_ring_vk_builder **out) {
    avrf_ring_key *k = nullptr;
    int st = index(su, nullptr, 0, &k);                                // the ring of padding points only
    if (st) return st;
    avrf_ring_vk_builder *b = new avrf_ring_vk_builder(); b->setup = su;
    for (int i = 0; i < 3; i++) b->C[i] = k->C[i];
    if (k->d_fixed4) (void)hipFree(k->d_fixed4); if (k->d_fixed_coef) (void)hipFree(k->d_fixed_coef); delete k;
    ensure_lagrange(su);
    *out = b;
    return AVRF_OK;
  }
  static int builder_append(avrf_ring_vk_builder *b, const uint8_t *pks_xy, size_t n) {
    avrf_ring_setup *su = b->setup;
    if (n > su->keyset - b->curr) return AVRF_RING_CAPACITY_EXCEEDED;   // src/ring.rs:606-608; nothing appended
    if (!n) return AVRF_OK;
    const H256 padx = Fr::from32(S::PAD_X), pady = Fr::from32(S::PAD_Y);
    std::vector<H256> sc(2 * n); std::vector<uint32_t> bi(2 * n);
    for (size_t i = 0; i < n; i++) {
      H256 x = Fr::load_le(pks_xy + 64 * i), y = Fr::load_le(pks_xy + 64 * i + 32);
      if (Fr::geq_p(x) || Fr::geq_p(y)) return AVRF_INVALID_DATA;
      sc[i] = Fr::from_mont(Fr::sub(Fr::to_mont(x), padx)); sc[n + i] = Fr::from_mont(Fr::sub(Fr::to_mont(y), pady));
      bi[i] = bi[n + i] = (uint32_t)(b->curr + i);
    }
    uint32_t *d = dev_scratch(su, 1, 2 * n * 36);
    uint32_t *d_sc = d, *d_bi = d + 2 * n * 8;
    HIP_CHECK(hipMemcpyAsync(d_sc, sc.data(), 2 * n * 32, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemcpyAsync(d_bi, bi.data(), 2 * n * 4, hipMemcpyHostToDevice, su->stream));
    std::vector<G1Aff> D; commit_sparse(su, d_sc, d_bi, n, 2, D);
    b->C[0] = g1_add_aff(b->C[0], D[0]); b->C[1] = g1_add_aff(b->C[1], D[1]);
    b->curr += n;
    return AVRF_OK;
  }

  // ---- RingProver::prove with blinding disabled (A.7) for a batch of proofs over one ring, in lockstep:
  // every device stage (iNTT, KZG commits, NTT(4N) + constraint aggregation + iNTT(4N), openings) runs ONCE for
  // the whole chunk -- batched NTTs and one batched MSM launch chain over the shared SRS per round of the
  // Fiat-Shamir transcript -- while the O(N) per-proof bookkeeping runs on the host between the rounds.
  // out: per proof 4*G1 || 7*Fr || G1 || Fr || G1 || G1.
  struct ProofState {
    H256 seedx, seedy, resx, resy, instx, insty, al[7], zeta, ev[7], lin_zw, nu[8];
    G1Aff C[4], Cq, pi[2];
    ArkTranscript t;
  };
  static uint32_t *dev_scratch(avrf_ring_setup *su, int which, size_t bytes) {
    uint32_t **p = &su->d_scr[which]; size_t *cap = &su->scr_cap[which];
    if (bytes > *cap) { if (*p) HIP_CHECK(hipFree(*p)); HIP_CHECK(hipMalloc(p, bytes)); *cap = bytes; }
    return *p;
  }
  // batched commit of `batch` coefficient vectors on the device in Montgomery form: vector b starts at
  // d_coeffs_mont + b * stride elements, its first n coefficients are committed (the source is left untouched)
  static void commit_device(avrf_ring_setup *su, const uint32_t *d_coeffs_mont, size_t stride, size_t n, size_t batch, std::vector<G1Aff> &out) {
    // (the digit kernel of the MSM takes the Montgomery limbs as they are: no plain copy of the coefficient vectors)
    constexpr int mont_id = std::is_same<F, FqBandersnatch>::value ? 1 : 2;
    static_assert(std::is_same<F, FqBandersnatch>::value || std::is_same<F, FqBabyJubJub>::value, "scalar field of the KZG commitments");
    std::vector<uint8_t> xy(batch * 2 * FQB);
    static const bool trace = getenv("AVRF_RING_TRACE") != nullptr;
    struct timespec t0; if (trace) { HIP_CHECK(hipStreamSynchronize(su->stream)); clock_gettime(CLOCK_MONOTONIC, &t0); }
    // many vectors at once and the table of all multiples is there: one gathered point per (coefficient, row), no buckets (msm.hip)
    if (su->direct && batch >= 32) msm_g1_direct_device(su->direct->t, d_coeffs_mont, n, stride, su->ws, su->stream, xy.data(), batch, mont_id);
    else msm_g1_fixed_device(su->curve, su->d_srs_table, su->table_c, su->n_srs, d_coeffs_mont, n, stride, su->ws, su->stream, xy.data(), batch, nullptr, mont_id);
    if (trace) { struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
      fprintf(stderr, "    commit n=%zu batch=%zu: %.3f ms wall, accumulate %.3f ms (c=%d seg=%d)\n", n, batch,
              (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, su->ws.accum_ms_last, su->ws.last_plan.c, su->ws.last_plan.lpb); }
    out.resize(batch);
    for (size_t b = 0; b < batch; b++) {
      memset(&out[b], 0, sizeof(G1Aff)); memcpy(out[b].xy, &xy[b * 2 * FQB], 2 * FQB);
      out[b].inf = true; for (int i = 0; i < 2 * FQB; i++) if (out[b].xy[i]) out[b].inf = false;
    }
  }
  // Lagrange-basis SRS for the witness columns, built on first use: L_i(tau) G = commit(iNTT(e_i)) (N commits in one
  // batched MSM), then the prefix sums PS_k on the host and a window table over [L_0 .. L_{N-1} | PS_0 .. PS_N].
  static void ensure_lagrange(avrf_ring_setup *su) {
    if (su->d_wit_table) return;
    const size_t N = su->N, nb = 2 * N + 1;
    // in tiles of 256 unit vectors: scratch is O(256 N) instead of the N x N matrix (2 GiB at N = 8192, 137 GB at 2^16)
    const size_t TILE = N < 256 ? N : 256;
    uint32_t *d_mat = dev_scratch(su, 0, TILE * N * 32);
    std::vector<G1Aff> lag; lag.reserve(N);
    for (size_t i0 = 0; i0 < N; i0 += TILE) {
      const size_t rows = N - i0 < TILE ? N - i0 : TILE;
      HIP_CHECK(hipMemsetAsync(d_mat, 0, rows * N * 32, su->stream));
      hipLaunchKernelGGL(k_set_diag<F>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, su->stream, d_mat, (uint32_t)N, (uint32_t)rows, (uint32_t)i0);
      { fp sc; memcpy(sc.v, su->ninv.l, 32); ntt_launch<F>(d_mat, (uint32_t)N, su->d_tw_n_inv, (uint32_t)rows, &sc, su->stream); }
      std::vector<G1Aff> part; commit_device(su, d_mat, N, N, rows, part);
      lag.insert(lag.end(), part.begin(), part.end());
    }
    su->lag_raw.clear();
    for (size_t i = 0; i < N; i++) g1_encode<G>(lag[i], false, su->lag_raw);
    using HG = typename T::HG; using FqN = typename T::FqN;
    std::vector<typename HG::Pt> ps(N + 1);
    ps[0] = HG::identity();
    for (size_t i = 0; i < N; i++) {
      typename HG::Pt q = HG::identity();
      if (!lag[i].inf) { typename FqN::El x, y; memcpy(x.l, lag[i].xy, FQB); memcpy(y.l, lag[i].xy + FQB, FQB);
        q.x = FqN::to_mont(x); q.y = FqN::to_mont(y); q.zz = FqN::one(); q.zzz = FqN::one(); }
      ps[i + 1] = HG::add(ps[i], q);
    }
    std::vector<uint8_t> le(nb * 2 * FQB);
    for (size_t i = 0; i < N; i++) memcpy(&le[i * 2 * FQB], lag[i].xy, 2 * FQB);
    HG::to_affine_bytes_batch(ps.data(), N + 1, &le[N * 2 * FQB]);
    uint8_t *d_le; uint32_t *d_flag, *d_bases;
    HIP_CHECK(hipMalloc(&d_le, le.size())); HIP_CHECK(hipMalloc(&d_flag, 4)); HIP_CHECK(hipMalloc(&d_bases, nb * 2 * FQB));
    HIP_CHECK(hipMemcpy(d_le, le.data(), le.size(), hipMemcpyHostToDevice)); HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream));
    launch_g1_bases(su->curve, d_le, nb, d_bases, d_flag, su->stream);
    su->wit_c = 7;
    if (const char *e = getenv("AVRF_RING_WIT_C")) { int v = atoi(e); if (v >= 4 && v <= 14) su->wit_c = v; }
    su->wit_nwin = (G::Fr::BITS + 1 + su->wit_c - 1) / su->wit_c;
    HIP_CHECK(hipMalloc(&su->d_wit_table, (size_t)su->wit_nwin * nb * 2 * FQB));
    build_g1_table(su->curve, d_bases, nb, su->wit_c, su->wit_nwin, su->d_wit_table, su->stream);
    HIP_CHECK(hipStreamSynchronize(su->stream));
    HIP_CHECK(hipFree(d_le)); HIP_CHECK(hipFree(d_flag));
    su->d_wit_bases = d_bases;
  }
  // `batch` sparse commits over the witness table: vector b = m (base index, plain scalar) pairs
  static void commit_sparse(avrf_ring_setup *su, const uint32_t *d_scalars_plain, const uint32_t *d_base_idx, size_t m, size_t batch, std::vector<G1Aff> &out) {
    std::vector<uint8_t> xy(batch * 2 * FQB);
    if (su->direct_wit && batch >= 32) msm_g1_direct_device(su->direct_wit->t, d_scalars_plain, m, m, su->ws, su->stream, xy.data(), batch, 0, d_base_idx);
    else msm_g1_fixed_device(su->curve, su->d_wit_table, su->wit_c, 2 * su->N + 1, d_scalars_plain, m, m, su->ws, su->stream, xy.data(), batch, d_base_idx);
    out.resize(batch);
    for (size_t b = 0; b < batch; b++) {
      memset(&out[b], 0, sizeof(G1Aff)); memcpy(out[b].xy, &xy[b * 2 * FQB], 2 * FQB);
      out[b].inf = true; for (int i = 0; i < 2 * FQB; i++) if (out[b].xy[i]) out[b].inf = false;
    }
  }
  static fp fp_zero_host() { fp r; memset(&r, 0, sizeof r); return r; }

  static avrf_ring_setup *second_lane(avrf_ring_setup *su) {
    if (su->lane1) return su->lane1;
    ensure_lagrange(su);                                               // so that the copy sees the witness table
    avrf_ring_setup *l = new avrf_ring_setup(*su);
    l->lane1 = nullptr; l->d_buf = nullptr; l->buf_cap = 0; l->ws = MsmWorkspace(); l->host_lines = nullptr; l->host_lines_free = nullptr;
    for (int i = 0; i < 6; i++) { l->d_scr[i] = nullptr; l->scr_cap[i] = 0; }
    HIP_CHECK(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
    su->lane1 = l;
    return l;
  }
  static int prove_chunk(avrf_ring_key *k, avrf_ring_setup *su, size_t n, const uint32_t *key_index, const uint8_t *blindings, bool hiding, uint8_t *out) {
    const size_t N = su->N, cap = su->cap, M = 4 * N, plen = 4 * FQB + 7 * 32 + FQB + 32 + 2 * FQB;
    const H256 one = Fr::one();
    static const bool trace = getenv("AVRF_RING_TRACE") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    auto cpu_now = [] { struct timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t_prev = now(), c_prev = cpu_now();                         // (trace: wall time and the PROCESS's CPU time of every phase, workers included)
    auto lap = [&](const char *what) { if (trace) { double t = now(), c = cpu_now(); fprintf(stderr, "  ring_prove[%zu] %-32s %8.3f ms wall %8.3f ms cpu\n", n, what, t - t_prev, c - c_prev); t_prev = t; c_prev = c; } };
    for (size_t i = 0; i < n; i++) if (key_index[i] >= k->n_keys) return AVRF_ERR_BAD_ARG;
    std::vector<ProofState> st(n);
    const H256 w_last = fr_pow<F>(su->w, cap - 1);
    // ---- round 0: the witness in sparse form (A.7 step 1).  Rows with bit 1: the signer's key and the set bits
    // of the blinding; the accumulator column only changes there, so it is cnt+1 points; the four KZG commits are
    // sparse MSMs over the Lagrange-basis SRS and its prefix sums (same group elements as the coefficient-form commits).
    ensure_lagrange(su);
    constexpr size_t MP = 264;                                         // padded entries per sparse vector (L + 2 + 3 zk rows <= 264)
    // hiding (RingContext with blinding, src/ring.rs:277-295): the last 3 rows of every witness column are uniformly
    // random field elements (w3f-ring-proof `private_column`); they only add 3 sparse terms to each commitment
    std::vector<H256> zk;
    if (hiding) {
      std::vector<uint8_t> rnd(n * 12 * 48);
      for (size_t o = 0; o < rnd.size();) { ssize_t g = getrandom(&rnd[o], rnd.size() - o, 0); if (g <= 0) return AVRF_ERR_BAD_ARG; o += (size_t)g; }
      zk.resize(n * 12);
      for (size_t i = 0; i < n * 12; i++) zk[i] = fr_from_be48<F>(&rnd[i * 48]);
    }
    // ---- round 0 + 1 (device): witness accumulation (k_ring_witness_acc), columns, coefficients, their 4N evaluations, 4n sparse
    // commits in one MSM chain.  Back to the host: result and instance of every proof (for the transcript and the constraints).
    uint32_t *d_coef = dev_scratch(su, 2, n * 4 * N * 32), *d_e4 = dev_scratch(su, 0, std::max(n * 4 * M * 32, n * 257 * 128));
    std::vector<H256> wout(n * 4);
    {
      const size_t b_pos = n * 256 * 4, b_cnt = n * 4, b_val = n * 257 * 64, b_sc = n * 4 * MP * 32, b_bi = n * 4 * MP * 4, b_zk = n * 12 * 32, b_bl = n * 32, b_out = n * 128;
      uint32_t *d_w = dev_scratch(su, 1, b_sc + b_val + b_zk + b_pos + b_bi + 2 * b_cnt + b_bl + b_out);
      uint32_t *d_sc = d_w, *d_val = d_sc + b_sc / 4, *d_zk = d_val + b_val / 4, *d_pos = d_zk + b_zk / 4, *d_bi = d_pos + b_pos / 4, *d_cnt = d_bi + b_bi / 4,
               *d_ki = d_cnt + n, *d_bl = d_ki + n, *d_out = d_bl + b_bl / 4;
      if (hiding) HIP_CHECK(hipMemcpyAsync(d_zk, zk.data(), b_zk, hipMemcpyHostToDevice, su->stream));
      HIP_CHECK(hipMemcpyAsync(d_ki, key_index, b_cnt, hipMemcpyHostToDevice, su->stream));
      HIP_CHECK(hipMemcpyAsync(d_bl, blindings, b_bl, hipMemcpyHostToDevice, su->stream));
      HIP_CHECK(hipMemsetAsync(d_sc, 0, b_sc, su->stream));
      HIP_CHECK(hipMemsetAsync(d_bi, 0, b_bi, su->stream));
      fp sx, sy; { const H256 ax = Fr::from32(S::ACC_X), ay = Fr::from32(S::ACC_Y); memcpy(sx.v, ax.l, 32); memcpy(sy.v, ay.l, 32); }
      hipLaunchKernelGGL(k_ring_witness_acc<S>, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, su->stream, (const te_pre *)k->d_points_pre, (const uint32_t *)d_ki,
                         (const uint8_t *)d_bl, (uint32_t)n, (uint32_t)su->keyset, (uint32_t)su->L, (uint32_t)N, (uint32_t)cap, sx, sy,
                         hiding ? (const uint32_t *)d_zk : nullptr, d_e4, d_pos, d_cnt, d_val, d_sc, d_bi, d_out);
      HIP_CHECK(hipMemcpyAsync(wout.data(), d_out, b_out, hipMemcpyDeviceToHost, su->stream));
      hipLaunchKernelGGL(k_ring_witness_cols<F>, dim3((unsigned)((N + 255) / 256), (unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_pos,
                         (const uint32_t *)d_cnt, (const uint32_t *)d_ki, (const uint32_t *)d_val, hiding ? (const uint32_t *)d_zk : nullptr, (uint32_t)N, (uint32_t)cap, d_coef);
      { fp sc; memcpy(sc.v, su->ninv.l, 32); ntt_launch<F>(d_coef, (uint32_t)N, su->d_tw_n_inv, (uint32_t)(4 * n), &sc, su->stream); }
      ntt_launch2<F>(d_coef, (uint32_t)N, d_e4, (uint32_t)M, su->d_tw_4n, (uint32_t)(4 * n), nullptr, su->stream);   // zero-extended to 4N
      std::vector<G1Aff> C; commit_sparse(su, d_sc, d_bi, MP, 4 * n, C);                                         // (waits for the stream)
      const H256 ax = Fr::from32(S::ACC_X), ay = Fr::from32(S::ACC_Y);
      for (size_t p = 0; p < n; p++) {
        for (int i = 0; i < 4; i++) st[p].C[i] = C[4 * p + i];
        st[p].seedx = ax; st[p].seedy = ay; st[p].resx = wout[4 * p]; st[p].resy = wout[4 * p + 1]; st[p].instx = wout[4 * p + 2]; st[p].insty = wout[4 * p + 3];
      }
    }
    lap("intt + ntt4n + 4n commits");
    ArkTranscript t0; transcript_prelude(k, t0);                       // shared by every proof over this ring
    parallel_for(n, [&](size_t p) {
      ProofState &ps = st[p];
      ps.t = t0;
      { std::vector<uint8_t> b; push_le32(b, ps.instx); push_le32(b, ps.insty); ps.t.label("instance"); ps.t.append(b); }
      { std::vector<uint8_t> b; for (int i = 0; i < 4; i++) g1_encode<G>(ps.C[i], false, b); ps.t.label("committed_cols"); ps.t.append(b); }
      for (int i = 0; i < 7; i++) ps.al[i] = challenge(ps.t, "constraints_aggregation");
    }, 64);                                                            // (3 us per proof on one thread: a worker is worth waking for ~200 us)
    lap("transcript: alphas");
    // per-chunk parameter block on the device: RingConsts[n] | zeta[n] | nu[8n] | ev[7n] | lin_zw[n]
    const size_t qlen = 3 * N + 1, olen = 3 * N;
    const size_t rc_bytes = (n * sizeof(RingConsts) + 31) / 32 * 32;
    uint32_t *d_par = dev_scratch(su, 3, rc_bytes + n * (1 + 8 + 7 + 1) * 32);
    RingConsts *d_rc = (RingConsts *)d_par;
    uint32_t *d_zeta = d_par + rc_bytes / 4, *d_nu = d_zeta + n * 8, *d_ev = d_nu + n * 64, *d_linzw = d_ev + n * 56;
    // ---- round 2 (device): constraint aggregation on the 4N domain, iNTT(4N), * Z_zk / (X^N - 1), n quotient commits
    uint32_t *d_agg = dev_scratch(su, 1, n * M * 32), *d_q = dev_scratch(su, 4, n * qlen * 32);
    {
      std::vector<RingConsts> rc(n);
      auto setfp = [](fp &d, const H256 &v) { memcpy(d.v, v.l, 32); };
      for (size_t p = 0; p < n; p++) {
        for (int i = 0; i < 7; i++) setfp(rc[p].alpha[i], st[p].al[i]);
        setfp(rc[p].w_last, w_last); setfp(rc[p].seedx, st[p].seedx); setfp(rc[p].seedy, st[p].seedy); setfp(rc[p].resx, st[p].resx); setfp(rc[p].resy, st[p].resy);
      }
      HIP_CHECK(hipMemcpyAsync(d_rc, rc.data(), n * sizeof(RingConsts), hipMemcpyHostToDevice, su->stream));
      hipLaunchKernelGGL(k_ring_constraints<S>, dim3((M + 255) / 256, (unsigned)n), dim3(256), 0, su->stream, d_e4, k->d_fixed4, su->d_l4, su->d_tw_4n,
                         (const RingConsts *)d_rc, (uint32_t)M, d_agg);
      fp sc; memcpy(sc.v, su->n4inv.l, 32);
      ntt_launch<F>(d_agg, (uint32_t)M, su->d_tw_4n_inv, (uint32_t)n, &sc, su->stream);
      Fp3 z;                                                           // Z(X) = prod_j (X - w^(N-3+j)) = X^3 + z2 X^2 + z1 X + z0
      { H256 r0 = fr_pow<F>(su->w, N - 3), r1 = Fr::mul(r0, su->w), r2 = Fr::mul(r1, su->w);
        H256 z2 = Fr::neg(Fr::add(Fr::add(r0, r1), r2)), z1 = Fr::add(Fr::add(Fr::mul(r0, r1), Fr::mul(r0, r2)), Fr::mul(r1, r2)), z0 = Fr::neg(Fr::mul(Fr::mul(r0, r1), r2));
        setfp(z.v[0], z0); setfp(z.v[1], z1); setfp(z.v[2], z2); }
      hipLaunchKernelGGL(k_ring_quotient<F>, dim3((unsigned)((qlen + 255) / 256), (unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_agg, (uint32_t)N,
                         (uint32_t)qlen, z, d_q);
      lap("constraints + quotient (launch)");
      std::vector<G1Aff> C; commit_device(su, d_q, qlen, qlen, n, C);
      for (size_t p = 0; p < n; p++) st[p].Cq = C[p];
    }
    lap("n quotient commits");
    // ---- round 3: evaluation point (host transcript), evaluations + linearisation (device)
    std::vector<H256> zs(n);
    {
      parallel_for(n, [&](size_t p) {
        ProofState &ps = st[p];
        std::vector<uint8_t> b; g1_encode<G>(ps.Cq, false, b); ps.t.label("quotient"); ps.t.append(b);
        ps.zeta = challenge(ps.t, "evaluation_point"); zs[p] = ps.zeta;
      }, 256);
      HIP_CHECK(hipMemcpyAsync(d_zeta, zs.data(), n * 32, hipMemcpyHostToDevice, su->stream));
    }
    uint32_t *d_lin = dev_scratch(su, 5, n * N * 32);
    fp w_dev; memcpy(w_dev.v, su->w.l, 32);
    hipLaunchKernelGGL(k_ring_evals<F>, dim3(7, (unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_coef, (const uint32_t *)k->d_fixed_coef,
                       (const uint32_t *)d_zeta, (uint32_t)N, d_ev);
    hipLaunchKernelGGL(k_ring_lin<S>, dim3((unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_coef, (const RingConsts *)d_rc, (const uint32_t *)d_zeta,
                       (const uint32_t *)d_ev, w_dev, (uint32_t)N, d_lin, d_linzw);
    {
      std::vector<H256> evh(n * 8);                                    // ev[7n] | lin_zw[n] are contiguous on the device
      HIP_CHECK(hipMemcpyAsync(evh.data(), d_ev, n * 8 * 32, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipStreamSynchronize(su->stream));
      std::vector<H256> nus(n * 8);
      parallel_for(n, [&](size_t p) {
        ProofState &ps = st[p];
        for (int i = 0; i < 7; i++) ps.ev[i] = evh[p * 7 + i];
        ps.lin_zw = evh[n * 7 + p];
        { std::vector<uint8_t> b; for (int i = 0; i < 7; i++) push_le32(b, ps.ev[i]); ps.t.label("register_evaluations"); ps.t.append(b); }
        { std::vector<uint8_t> bb; push_le32(bb, ps.lin_zw); ps.t.label("shifted_linearization_evaluation"); ps.t.append(bb); }
        for (int i = 0; i < 8; i++) { ps.nu[i] = challenge(ps.t, "kzg_aggregation"); nus[p * 8 + i] = ps.nu[i]; }
      }, 64);
      HIP_CHECK(hipMemcpyAsync(d_nu, nus.data(), n * 8 * 32, hipMemcpyHostToDevice, su->stream));
      HIP_CHECK(hipStreamSynchronize(su->stream));                     // nus goes out of scope
    }
    lap("evals + linearisation");
    // ---- round 4 (device): aggregated opening at zeta, opening of lin at zeta*w, 2n opening commits
    {
      uint32_t *d_aggz = d_agg;                                        // the 4N aggregate is dead: reuse (qlen <= 4N)
      uint32_t *d_open = dev_scratch(su, 0, n * 4 * M * 32);           // = d_e4 (dead): q1 (n x olen) | q2 (n x N)
      uint32_t *d_q1 = d_open, *d_q2 = d_open + n * olen * 8;
      hipLaunchKernelGGL(k_ring_aggz<F>, dim3((unsigned)((qlen + 255) / 256), (unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_coef,
                         (const uint32_t *)k->d_fixed_coef, (const uint32_t *)d_q, (const uint32_t *)d_nu, (uint32_t)N, (uint32_t)qlen, d_aggz);
      fp one_m; memcpy(one_m.v, one.l, 32);
      hipLaunchKernelGGL(k_ring_divlin<F>, dim3((unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_aggz, (uint32_t)qlen, (uint32_t)qlen,
                         (const uint32_t *)d_zeta, one_m, d_q1, (uint32_t)olen);
      // small chunks: BOTH openings in one MSM chain (2n vectors of stride 3N, the short one zero-padded: zero digits are dropped
      // by the sort) -- a chain of 32-128 proofs is latency-bound (k_wsum_blk: ~50 sequential G1 additions per bucket set, 0.85 ms
      // whatever the number of sets), so the second chain cost as much as the first; large chunks keep two exact-length chains
      const bool merged = n <= 128;
      const size_t q2_stride = merged ? olen : N;
      if (merged) HIP_CHECK(hipMemsetAsync(d_q2, 0, n * olen * 32, su->stream));
      hipLaunchKernelGGL(k_ring_divlin<F>, dim3((unsigned)n), dim3(256), 0, su->stream, (const uint32_t *)d_lin, (uint32_t)N, (uint32_t)N,
                         (const uint32_t *)d_zeta, w_dev, d_q2, (uint32_t)q2_stride);
      std::vector<G1Aff> C1, C2;
      if (merged) {
        commit_device(su, d_q1, olen, olen, 2 * n, C1);
        for (size_t p = 0; p < n; p++) { st[p].pi[0] = C1[p]; st[p].pi[1] = C1[n + p]; }
      } else {
        commit_device(su, d_q1, olen, olen, n, C1);
        commit_device(su, d_q2, N, N - 1, n, C2);
        for (size_t p = 0; p < n; p++) { st[p].pi[0] = C1[p]; st[p].pi[1] = C2[p]; }
      }
    }
    lap("2n opening commits");
    for (size_t p = 0; p < n; p++) {
      const ProofState &ps = st[p];
      std::vector<uint8_t> pr;
      for (int i = 0; i < 4; i++) g1_encode<G>(ps.C[i], true, pr);
      for (int i = 0; i < 7; i++) push_le32(pr, ps.ev[i]);
      g1_encode<G>(ps.Cq, true, pr); push_le32(pr, ps.lin_zw); g1_encode<G>(ps.pi[0], true, pr); g1_encode<G>(ps.pi[1], true, pr);
      if (pr.size() != plen) return AVRF_ERR_BAD_ARG;
      memcpy(out + plen * p, pr.data(), plen);
    }
    lap("proof bytes");
    return AVRF_OK;
  }

  // ---- G1 decompression (ark-serialize compressed forms); p = 3 mod 4 for both curves
  using FqN = typename T::FqN; using QEl = typename FqN::El;
  static bool g1_decompress(const uint8_t *b, G1Aff *out) {
    memset(out, 0, sizeof *out);
    uint8_t le[48]; bool big, inf;
    if (FQB == 48) { inf = b[0] & 0x40; big = b[0] & 0x20; if (!(b[0] & 0x80)) return false; for (int i = 0; i < FQB; i++) le[i] = b[FQB - 1 - i]; le[FQB - 1] &= 0x1f; }
    else { inf = b[FQB - 1] & 0x40; big = b[FQB - 1] & 0x80; memcpy(le, b, FQB); le[FQB - 1] &= 0x3f; }
    if (inf) {                                                          // canonical encoding only: no sort flag, every other bit zero
      if (big) return false;
      for (int i = 0; i < FQB; i++) if (le[i]) return false;
      out->inf = true; return true;
    }
    QEl x; memcpy(x.l, le, FQB);
    QEl t; if (FqN::subb(t, x, FqN::P()) == 0) return false;           // x >= p
    QEl xm = FqN::to_mont(x);
    QEl rhs = FqN::add(FqN::mul(FqN::sqr(xm), xm), FqN::from32(G::B));
    static const QEl e = [] { QEl v = FqN::P(), one = FqN::zero(); one.l[0] = 1; FqN::addc(v, v, one);   // (p + 1) / 4
                              for (int k = 0; k < 2; k++) for (int i = 0; i < FqN::L; i++) v.l[i] = (v.l[i] >> 1) | (i + 1 < FqN::L ? v.l[i + 1] << 63 : 0);
                              return v; }();
    QEl y = FqN::one();
    for (int i = 64 * FqN::L - 1; i >= 0; i--) { y = FqN::sqr(y); if ((e.l[i / 64] >> (i % 64)) & 1) y = FqN::mul(y, rhs); }
    if (!FqN::eq(FqN::sqr(y), rhs)) return false;                       // not on the curve
    QEl yp = FqN::from_mont(y), half = FqN::from32(G::Fq::HALF);
    bool is_big = FqN::subb(t, half, yp) != 0;
    if (is_big != big) { y = FqN::neg(y); yp = FqN::from_mont(y); }
    memcpy(out->xy, x.l, FQB); memcpy(out->xy + FQB, yp.l, FQB);
    return true;
  }
  // The same subgroup test as k_g1_subgroup_bls (phi(P) = [-z^2] P) on the host, for small batches: one lane of the device
  // needs 2.8 ms for the two 64-bit ladders however few points there are; a host core does them in ~0.1 ms per point.
  static bool g1_in_subgroup_host(const G1Aff &a) {
    if (FQB != 48 || a.inf) return true;                              // BN254 G1 has cofactor 1
    using HG = typename T::HG;
    static const uint32_t BETA[12] = {0x798a64e8, 0x30f1361b, 0x7ece5a2a, 0xf3b8ddab, 0xc61577f7, 0x16a8ca3a,
                                      0x74fd029b, 0xc26a2ff8, 0x60701c6e, 0x3636b766, 0x241b6160, 0x051ba4ab};
    QEl beta; memcpy(beta.l, BETA, sizeof BETA > sizeof beta.l ? sizeof beta.l : sizeof BETA);
    QEl x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y); memcpy(x.l, a.xy, FQB); memcpy(y.l, a.xy + FQB, FQB);
    typename HG::Pt p; p.x = FqN::to_mont(x); p.y = FqN::to_mont(y); p.zz = FqN::one(); p.zzz = FqN::one();
    const uint64_t z = 0xd201000000010000ull;
    typename HG::Pt q = p;
    for (int b = 62; b >= 0; b--) { q = HG::dbl(q); if ((z >> b) & 1) q = HG::add(q, p); }
    const typename HG::Pt q1 = q;
    for (int b = 62; b >= 0; b--) { q = HG::dbl(q); if ((z >> b) & 1) q = HG::add(q, q1); }
    if (HG::is_identity(q)) return false;
    return FqN::eq(q.x, FqN::mul(FqN::mul(beta, p.x), q.zz)) && FqN::eq(q.y, FqN::neg(FqN::mul(p.y, q.zzz)));
  }

  // coordinates < p and y^2 = x^3 + b (the point at infinity passes): the check of an UNCOMPRESSED G1 entry that ark-serialize's
  // Validate::Yes makes before the subgroup test (a decompressed point is on the curve by construction)
  static bool g1_on_curve_host(const G1Aff &a) {
    if (a.inf) return true;
    QEl x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y); memcpy(x.l, a.xy, FQB); memcpy(y.l, a.xy + FQB, FQB);
    if (FqN::geq(x, FqN::P()) || FqN::geq(y, FqN::P())) return false;
    const QEl xm = FqN::to_mont(x), ym = FqN::to_mont(y);
    return FqN::eq(FqN::sqr(ym), FqN::add(FqN::mul(FqN::sqr(xm), xm), FqN::from32(G::B)));
  }

  static G1Aff g1_msm(avrf_ring_setup *su, const std::vector<uint8_t> &bases_xy, const std::vector<H256> &scalars_plain,
                      bool check_subgroup = false, bool *bad_points = nullptr) {
    const size_t n = scalars_plain.size();
    G1Aff r; memset(&r, 0, sizeof r); r.inf = true;
    if (!n) return r;
    // staging in the setup's grow-only scratch (no allocation on the verification path)
    const size_t pb = (n * 2 * FQB + 255) / 256 * 256, sb = (n * 32 + 255) / 256 * 256;
    uint8_t *base = (uint8_t *)dev_scratch(su, 1, 2 * pb + sb + 256);
    uint8_t *d_xy = base; uint32_t *d_b = (uint32_t *)(base + pb), *d_s = (uint32_t *)(base + 2 * pb), *d_flag = (uint32_t *)(base + 2 * pb + sb);
    HIP_CHECK(hipMemcpyAsync(d_xy, bases_xy.data(), n * 2 * FQB, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemcpyAsync(d_s, scalars_plain.data(), n * 32, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream));
    launch_g1_bases(su->curve, d_xy, n, d_b, d_flag, su->stream);
    if (check_subgroup) launch_g1_subgroup_check(su->curve, d_b, n, d_flag, su->stream);   // Validate::Yes of the deserialised points
    msm_g1_device(su->curve, d_b, d_s, n, su->ws, su->stream, r.xy);       // (synchronises the stream)
    if (bad_points) { uint32_t f = 0; HIP_CHECK(hipMemcpy(&f, d_flag, 4, hipMemcpyDeviceToHost)); *bad_points = f != 0; }
    r.inf = true; for (int i = 0; i < 2 * FQB; i++) if (r.xy[i]) r.inf = false;
    return r;
  }

  // the verifier's two sums over one launch chain: vector 0 = scalars_a over bases_a, vector 1 = scalars_b over bases_b, as two
  // scalar vectors over the concatenated bases (a zero scalar has no digits and costs nothing downstream).  One chain instead
  // of two is 0.25 ms of launch latency less per verification call, and the host's two bit-sum Horners run side by side.
  static void g1_msm2(avrf_ring_setup *su, const std::vector<uint8_t> &bases_a, const std::vector<H256> &scalars_a,
                      const std::vector<uint8_t> &bases_b, const std::vector<H256> &scalars_b, bool check_subgroup_a, bool *bad_points,
                      G1Aff *out_a, G1Aff *out_b) {
    const size_t na = scalars_a.size(), nbv = scalars_b.size(), n = na + nbv;
    const size_t pb = (n * 2 * FQB + 255) / 256 * 256, sb = (2 * n * 32 + 255) / 256 * 256;
    uint8_t *base = (uint8_t *)dev_scratch(su, 1, 2 * pb + sb + 256);
    uint8_t *d_xy = base; uint32_t *d_b = (uint32_t *)(base + pb), *d_s = (uint32_t *)(base + 2 * pb), *d_flag = (uint32_t *)(base + 2 * pb + sb);
    HIP_CHECK(hipMemcpyAsync(d_xy, bases_a.data(), na * 2 * FQB, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemcpyAsync(d_xy + na * 2 * FQB, bases_b.data(), nbv * 2 * FQB, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemsetAsync(d_s, 0, 2 * n * 32, su->stream));
    HIP_CHECK(hipMemcpyAsync(d_s, scalars_a.data(), na * 32, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemcpyAsync(d_s + (n + na) * 8, scalars_b.data(), nbv * 32, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream));
    launch_g1_bases(su->curve, d_xy, n, d_b, d_flag, su->stream);
    if (check_subgroup_a) launch_g1_subgroup_check(su->curve, d_b, na, d_flag, su->stream);   // Validate::Yes of the deserialised points
    uint8_t xy[2][2 * FQB];
    msm_g1_device(su->curve, d_b, d_s, n, su->ws, su->stream, &xy[0][0], 2, n);            // (synchronises the stream)
    if (bad_points) { uint32_t f = 0; HIP_CHECK(hipMemcpy(&f, d_flag, 4, hipMemcpyDeviceToHost)); *bad_points = f != 0; }
    G1Aff *outs[2] = {out_a, out_b};
    for (int v = 0; v < 2; v++) {
      memset(outs[v], 0, sizeof(G1Aff)); memcpy(outs[v]->xy, xy[v], 2 * FQB);
      outs[v]->inf = true; for (int i = 0; i < 2 * FQB; i++) if (xy[v][i]) outs[v]->inf = false;
    }
  }

  // ---- CanonicalSerialize of the setup objects (src/ring.rs:484-542): RingSetup = its PcsParams (URS { powers_in_g1,
  // powers_in_g2 }), RingBuilderPcsParams = Vec<G1Affine> (the SRS in Lagrangian form); both in either ark-serialize mode
  static G1Aff g1_from_raw(const uint8_t *p) {                         // one serialize_uncompressed G1 entry
    G1Aff a; memset(&a, 0, sizeof a);
    if (FQB == 48) { a.inf = p[0] & 0x40; if (!a.inf) for (int k = 0; k < FQB; k++) { a.xy[k] = p[FQB - 1 - k]; a.xy[FQB + k] = p[2 * FQB - 1 - k]; } }
    else { a.inf = p[2 * FQB - 1] & 0x40; if (!a.inf) { memcpy(a.xy, p, 2 * FQB); a.xy[2 * FQB - 1] &= 0x3f; } }
    return a;
  }
  static void put_points(const std::vector<uint8_t> &raw, size_t n, bool compress, std::vector<uint8_t> &o) {
    { uint64_t v = n; size_t at = o.size(); o.resize(at + 8); memcpy(&o[at], &v, 8); }
    if (!compress) { o.insert(o.end(), raw.begin(), raw.begin() + n * 2 * FQB); return; }
    for (size_t i = 0; i < n; i++) g1_encode<G>(g1_from_raw(&raw[i * 2 * FQB]), true, o);
  }
  static int setup_serialize(avrf_ring_setup *su, bool compress, std::vector<uint8_t> &o) {
    using HP = HostPairing<G>;
    put_points(su->g1_raw, su->n_srs, compress, o);
    { uint64_t v = 2; size_t at = o.size(); o.resize(at + 8); memcpy(&o[at], &v, 8); }
    if (!compress) { o.insert(o.end(), su->g2_raw.begin(), su->g2_raw.end()); return AVRF_OK; }
    for (int i = 0; i < 2; i++) {
      typename HP::G2 q; HP::g2_decode(su->g2_raw.data() + (size_t)i * 4 * FQB, &q);
      size_t at = o.size(); o.resize(at + 2 * FQB); HP::g2_encode_compressed(q, &o[at]);
    }
    return AVRF_OK;
  }
  static int builder_params_serialize(avrf_ring_setup *su, bool compress, std::vector<uint8_t> &o) {
    ensure_lagrange(su);
    put_points(su->lag_raw, su->N, compress, o);
    return AVRF_OK;
  }

  // ---- n independent KZG pairing checks on the device: ok[i] = [ e(A_i, g2) * e(B_i, tau g2) == 1 ]   (pairing.hip)
  static void ensure_pairing(avrf_ring_setup *su) {
    if (su->ptab_ready) return;
    su->ptab.build(su->curve, su->g2_raw.data(), 2, su->stream);
    su->ptab_ready = true;
  }
  // the Miller-loop lines of the setup's two fixed G2 arguments, tabulated once for the host pairing (host_pairing.h G2Lines)
  using HP = HostPairing<G>;
  static const typename HP::G2Lines *host_lines(avrf_ring_setup *su) {
    static std::mutex lines_mu;
    std::lock_guard<std::mutex> lk(lines_mu);
    if (!su->host_lines) {
      typename HP::G2 q[2];
      const size_t g2len = su->g2_raw.size() / 2;
      HP::g2_decode(su->g2_raw.data(), &q[0]); HP::g2_decode(su->g2_raw.data() + g2len, &q[1]);
      auto *t = new typename HP::G2Lines[2];
      t[0] = HP::g2_lines(q[0]); t[1] = HP::g2_lines(q[1]);
      su->host_lines = t;
      su->host_lines_free = [](void *p) { delete[] static_cast<typename HP::G2Lines *>(p); };
    }
    return static_cast<const typename HP::G2Lines *>(su->host_lines);
  }
  // Few independent checks: ONE wave of the device pairing kernel needs 11.7 ms whatever it carries, a host core 1.0 ms per check
  // (line tables), so up to two checks per pool thread (at most 16) are finished on the host pool from the device's G1 sums.
  static bool few_checks(size_t n) { size_t t = HostPool::get().size(); if (t > 8) t = 8; return n <= 2 * t; }
  // pts: n x {A, B} Montgomery affine points as launch_g1_bases / launch_g1_lincomb write them ((0, 0) = infinity)
  static void host_pair_checks(avrf_ring_setup *su, const uint32_t *pts, size_t n, int32_t *ok) {
    const typename HP::G2Lines *lines = host_lines(su);
    constexpr size_t FW = FQB / 4;
    parallel_for(n, [&](size_t it) {
      QEl px[2], py[2]; bool pinf[2];
      for (int i = 0; i < 2; i++) {
        const uint32_t *w = pts + (2 * it + i) * 2 * FW;
        memset(&px[i], 0, sizeof(QEl)); memset(&py[i], 0, sizeof(QEl));
        memcpy(px[i].l, w, FQB); memcpy(py[i].l, w + FW, FQB);
        bool z = true; for (size_t k = 0; k < 2 * FW; k++) z = z && w[k] == 0;
        pinf[i] = z;
      }
      ok[it] = HP::product_is_one_lines(px, py, pinf, lines, 2) ? 1 : 0;
    });
  }
  static int pairing_check(avrf_ring_setup *su, size_t n, const uint8_t *a_xy, const uint8_t *b_xy, int32_t *ok_out) {
    if (!n) return AVRF_OK;
    const size_t e1 = 2 * FQB;
    std::vector<uint8_t> le(2 * n * e1);
    for (size_t i = 0; i < n; i++) { memcpy(&le[(2 * i) * e1], a_xy + i * e1, e1); memcpy(&le[(2 * i + 1) * e1], b_xy + i * e1, e1); }
    const size_t pb = (2 * n * e1 + 255) / 256 * 256, ob = (n * 4 + 255) / 256 * 256;
    uint8_t *base = (uint8_t *)dev_scratch(su, 1, 2 * pb + ob + 256);
    uint8_t *d_le = base; uint32_t *d_pts = (uint32_t *)(base + pb); int32_t *d_ok = (int32_t *)(base + 2 * pb); uint32_t *d_flag = (uint32_t *)(base + 2 * pb + ob);
    HIP_CHECK(hipMemcpyAsync(d_le, le.data(), 2 * n * e1, hipMemcpyHostToDevice, su->stream));
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream));
    launch_g1_bases(su->curve, d_le, 2 * n, d_pts, d_flag, su->stream);
    uint32_t flag = 0;
    if (few_checks(n)) {
      std::vector<uint32_t> pts(2 * n * e1 / 4);
      HIP_CHECK(hipMemcpyAsync(pts.data(), d_pts, 2 * n * e1, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipStreamSynchronize(su->stream)); HIP_CHECK(hipGetLastError());
      if (flag) return AVRF_INVALID_DATA;
      host_pair_checks(su, pts.data(), n, ok_out);
      return AVRF_OK;
    }
    ensure_pairing(su);
    launch_pairing_check(su->ptab, d_pts, n, d_ok, su->stream);
    HIP_CHECK(hipMemcpyAsync(ok_out, d_ok, n * 4, hipMemcpyDeviceToHost, su->stream));
    HIP_CHECK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, su->stream));
    HIP_CHECK(hipStreamSynchronize(su->stream)); HIP_CHECK(hipGetLastError());
    return flag ? AVRF_INVALID_DATA : AVRF_OK;
  }

  // ---- RingVerifier::verify / multi-ring batch verifier (src/ring.rs:242,693-735).
  // Every item's two KZG openings are folded with 128-bit randomisers into
  //   e(sum r (C - v g1 + z pi), g2) * e(-sum r pi, tau g2) == 1 :
  // two G1 MSMs on the GPU (10 n + 1 and 2 n terms) and one 2-pairing check on the host.
  // each_status != nullptr: the n items are verified INDEPENDENTLY (n x RingVerifier::verify): per-item status, one 2-pairing
  // check per item on the device (k_g1_lincomb + pairing.hip) instead of one randomised check for the whole batch.
  static constexpr size_t DEVICE_DECOMPRESS_MIN = 128;   // items; below, the pool's cores finish sooner than a device round trip
  static int verify_batch(avrf_ring_setup *su, size_t n, const uint8_t *commitments, const uint32_t *ring_of_item, size_t n_rings,
                          const uint8_t *instances_xy, const uint8_t *proofs, int32_t *each_status = nullptr) {
    if (n == 0) return AVRF_OK;
    static const bool trace = getenv("AVRF_RING_TRACE") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t_prev = now();
    auto lap = [&](const char *what) { if (trace) { double t = now(); fprintf(stderr, "  ring_verify[%zu] %-28s %8.3f ms\n", n, what, t - t_prev); t_prev = t; } };
    const size_t N = su->N, cap = su->cap, plen = 4 * FQB + 7 * 32 + FQB + 32 + 2 * FQB, clen = 3 * FQB;
    const H256 one = Fr::one();
    std::vector<G1Aff> fixed(3 * n_rings);
    for (size_t i = 0; i < 3 * n_rings; i++) if (!g1_decompress(commitments + FQB * i, &fixed[i])) return AVRF_INVALID_DATA;
    // Validate::Yes (subgroup) of the deserialised G1 points: on the host pool for small batches, on the device otherwise
    const bool host_subgroup = FQB == 48 && 10 * n + 3 * n_rings <= 512;
    if (host_subgroup) {
      std::atomic<int> badc{0};
      parallel_for(3 * n_rings, [&](size_t i) { if (!g1_in_subgroup_host(fixed[i])) badc = 1; });
      if (badc) return AVRF_INVALID_DATA;
    }
    (void)clen;
    // randomisers: SHAKE128 over everything the batch contains
    // (the statement is bound in full: sizes, the verifier key, which ring every proof is checked against)
    HostShake128 rh; rh.update("avrf-ring-batch", 15);
    { uint64_t hdr[3] = {(uint64_t)n, (uint64_t)n_rings, (uint64_t)N}; rh.update(hdr, sizeof hdr); }
    rh.update(su->g1_0.xy, 2 * FQB); rh.update(su->g2_raw.data(), su->g2_raw.size());
    for (size_t i = 0; i < n; i++) { uint32_t ri = ring_of_item ? ring_of_item[i] : 0; rh.update(&ri, 4); }
    rh.update(commitments, 3 * FQB * n_rings); rh.update(instances_xy, 64 * n); rh.update(proofs, plen * n);
    std::vector<uint8_t> rnd(32 * n); rh.squeeze_copy(rnd.data(), rnd.size());
    // per item: 10 terms of the first MSM (3 fixed + 4 witness + C_q + pi1 + pi2) and 2 of the second; the
    // items are independent until the MSMs, so the host part runs on the thread pool
    std::vector<uint8_t> b1((10 * n + 1) * 2 * FQB), b2(2 * n * 2 * FQB); std::vector<H256> s1(10 * n + 1), s2(2 * n);
    auto put = [&](std::vector<uint8_t> &b, std::vector<H256> &sv, size_t slot, const G1Aff &p, const H256 &k_mont) {
      memcpy(&b[slot * 2 * FQB], p.xy, 2 * FQB); sv[slot] = Fr::from_mont(k_mont); };
    std::vector<H256> gsc(n);
    std::atomic<int> status{AVRF_OK};
    const H256 w_last = fr_pow<F>(su->w, cap - 1), ninv = su->ninv;
    H256 wz[3]; for (int j = 0; j < 3; j++) wz[j] = fr_pow<F>(su->w, N - 3 + j);
    H256 seedx = Fr::from32(S::ACC_X), seedy = Fr::from32(S::ACC_Y);
    std::vector<int32_t> item_st(each_status ? n : 0, 0);
    auto fail = [&](size_t it, int st) { if (each_status) item_st[it] = st; else status = st; };
    // the 7 G1 points of every proof: decompression (a 381-bit square root each) and, for small batches, the subgroup test,
    // one host task per POINT so that a single verification spreads over the pool
    const size_t g1_off[7] = {0, (size_t)FQB, 2 * (size_t)FQB, 3 * (size_t)FQB, 4 * (size_t)FQB + 7 * 32, 5 * (size_t)FQB + 8 * 32, 6 * (size_t)FQB + 8 * 32};
    std::vector<G1Aff> dec(7 * n); std::vector<uint8_t> dec_ok(7 * n, 0);
    if (n >= DEVICE_DECOMPRESS_MIN) {
      // large batches: the square roots on the device (k_g1_decompress), one lane per point; the y coordinates come back
      // because the transcript replay below absorbs the uncompressed points
      const size_t np = 7 * n, cb = (np * FQB + 255) / 256 * 256, xb = (np * 2 * FQB + 255) / 256 * 256;
      std::vector<uint8_t> comp(np * FQB), xy(np * 2 * FQB);
      for (size_t k = 0; k < np; k++) memcpy(&comp[k * FQB], proofs + plen * (k / 7) + g1_off[k % 7], FQB);
      uint8_t *base = (uint8_t *)dev_scratch(su, 1, cb + xb + np + 256);
      HIP_CHECK(hipMemcpyAsync(base, comp.data(), np * FQB, hipMemcpyHostToDevice, su->stream));
      launch_g1_decompress(su->curve, base, np, base + cb, base + cb + xb, su->stream);
      HIP_CHECK(hipMemcpyAsync(xy.data(), base + cb, np * 2 * FQB, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipMemcpyAsync(dec_ok.data(), base + cb + xb, np, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipStreamSynchronize(su->stream)); HIP_CHECK(hipGetLastError());
      for (size_t k = 0; k < np; k++) {
        memset(&dec[k], 0, sizeof dec[k]);
        if (dec_ok[k] == 2) dec[k].inf = true; else memcpy(dec[k].xy, &xy[k * 2 * FQB], 2 * FQB);
      }
      lap("G1 decompression (device)");
    } else
    parallel_for(7 * n, [&](size_t k) {
      const size_t it = k / 7, j = k % 7;
      bool ok = g1_decompress(proofs + plen * it + g1_off[j], &dec[k]);
      if (ok && host_subgroup) ok = g1_in_subgroup_host(dec[k]);
      dec_ok[k] = ok;
    });
    // the transcript up to the verifier key is the same for every proof checked against one ring: absorbed once per ring
    std::vector<ArkTranscript> t_ring(n_rings);
    for (size_t r = 0; r < n_rings; r++) {
      ArkTranscript &t = t_ring[r];
      t.label(S::SUITE_ID, S::SUITE_ID_LEN); t.label("vk");
      std::vector<uint8_t> vk; g1_encode<G>(su->g1_0, false, vk); vk.insert(vk.end(), su->g2_raw.begin(), su->g2_raw.end());
      for (int i = 0; i < 3; i++) g1_encode<G>(fixed[3 * r + i], false, vk);
      t.append(vk);
    }
    parallel_for(n, [&](size_t it) {
      const uint8_t *pr = proofs + plen * it;
      const uint32_t ring = ring_of_item ? ring_of_item[it] : 0;
      if (ring >= n_rings) { status = AVRF_ERR_BAD_ARG; return; }
      const G1Aff *fx = &fixed[3 * ring];
      G1Aff C[4], Cq, pi1, pi2; H256 ev[7], lin_zw;
      for (int j = 0; j < 7; j++) if (!dec_ok[7 * it + j]) { fail(it, AVRF_INVALID_DATA); return; }
      size_t off = 0;
      for (int i = 0; i < 4; i++) { C[i] = dec[7 * it + i]; off += FQB; }
      for (int i = 0; i < 7; i++) { H256 v = Fr::load_le(pr + off); if (Fr::geq_p(v)) { fail(it, AVRF_INVALID_DATA); return; } ev[i] = Fr::to_mont(v); off += 32; }
      Cq = dec[7 * it + 4]; off += FQB;
      { H256 v = Fr::load_le(pr + off); if (Fr::geq_p(v)) { fail(it, AVRF_INVALID_DATA); return; } lin_zw = Fr::to_mont(v); off += 32; }
      pi1 = dec[7 * it + 5]; off += FQB;
      pi2 = dec[7 * it + 6]; off += FQB;
      H256 ix = Fr::load_le(instances_xy + 64 * it), iy = Fr::load_le(instances_xy + 64 * it + 32);
      if (Fr::geq_p(ix) || Fr::geq_p(iy)) { fail(it, AVRF_INVALID_DATA); return; }
      H256 ixm = Fr::to_mont(ix), iym = Fr::to_mont(iy);
      // transcript replay
      ArkTranscript t = t_ring[ring];
      { std::vector<uint8_t> b(instances_xy + 64 * it, instances_xy + 64 * it + 64); t.label("instance"); t.append(b); }
      { std::vector<uint8_t> b; for (int i = 0; i < 4; i++) g1_encode<G>(C[i], false, b); t.label("committed_cols"); t.append(b); }
      H256 al[7]; for (int i = 0; i < 7; i++) al[i] = challenge(t, "constraints_aggregation");
      { std::vector<uint8_t> b; g1_encode<G>(Cq, false, b); t.label("quotient"); t.append(b); }
      H256 zeta = challenge(t, "evaluation_point");
      { std::vector<uint8_t> b(pr + 4 * FQB, pr + 4 * FQB + 7 * 32); t.label("register_evaluations"); t.append(b); }
      { std::vector<uint8_t> b(pr + 5 * FQB + 7 * 32, pr + 5 * FQB + 8 * 32); t.label("shifted_linearization_evaluation"); t.append(b); }
      H256 nu[8]; for (int i = 0; i < 8; i++) nu[i] = challenge(t, "kzg_aggregation");
      // q(zeta) from the evaluations (A.8)
      const H256 x2 = ev[0], y2 = ev[1], sel = ev[2], b = ev[3], ip = ev[4], x1 = ev[5], y1 = ev[6];
      const H256 nl = Fr::sub(zeta, w_last), omb = Fr::sub(one, b);
      H256 zn = zeta; for (size_t k = 1; k < N; k <<= 1) zn = Fr::sqr(zn);
      const H256 zn1 = Fr::sub(zn, one);
      if (Fr::is_zero(zn1)) { fail(it, AVRF_VERIFICATION_FAILURE); return; }
      HostExt sd; sd.x = seedx; sd.y = seedy; sd.t = Fr::mul(seedx, seedy); sd.z = one;
      HostExt in; in.x = ixm; in.y = iym; in.t = Fr::mul(ixm, iym); in.z = one;
      HostExt rs = Te::add(sd, in);
      // the item's four inversions -- 1 / (zeta - 1), 1 / (zeta - w^(cap-1)) of the two Lagrange values, 1 / z of seed + instance,
      // 1 / (zeta^N - 1) -- as ONE (Montgomery's trick; a fixed-exponent inversion is 380 products, 13 us).  zeta^N != 1 was checked, so
      // the first, second and fourth are non-zero; z = 0 (an instance off the curve) keeps the old meaning: 1 / 0 = 0
      const bool z_zero = Fr::is_zero(rs.z);
      const H256 v0 = Fr::sub(zeta, one), v1 = nl, v2 = z_zero ? one : rs.z, v3 = zn1;
      const H256 p01 = Fr::mul(v0, v1), p012 = Fr::mul(p01, v2);
      H256 tinv = Fr::inv(Fr::mul(p012, v3));
      const H256 i3 = Fr::mul(tinv, p012); tinv = Fr::mul(tinv, v3);
      const H256 i2 = Fr::mul(tinv, p01); tinv = Fr::mul(tinv, v2);
      const H256 i1 = Fr::mul(tinv, v0), i0 = Fr::mul(tinv, v1);
      const H256 zn1n = Fr::mul(zn1, ninv);
      const H256 lf = Fr::mul(zn1n, i0), ll = Fr::mul(Fr::mul(w_last, zn1n), i1);      // L_i(zeta) = w^i (zeta^N - 1) / (N (zeta - w^i))
      const H256 rzi = z_zero ? H256{{0, 0, 0, 0}} : i2; const H256 resx = Fr::mul(rs.x, rzi), resy = Fr::mul(rs.y, rzi);
      const H256 x1y1 = Fr::mul(x1, y1), x2y2 = Fr::mul(x2, y2);
      H256 rest[7];
      rest[0] = Fr::mul(Fr::neg(Fr::add(ip, Fr::mul(sel, b))), nl);
      rest[1] = Fr::mul(Fr::sub(Fr::mul(b, Fr::neg(Fr::add(x1y1, x2y2))), Fr::mul(omb, x1)), nl);
      rest[2] = Fr::mul(Fr::sub(Fr::mul(b, Fr::sub(x2y2, x1y1)), Fr::mul(omb, y1)), nl);
      rest[3] = Fr::mul(b, omb);
      rest[4] = Fr::add(Fr::mul(lf, Fr::sub(x1, seedx)), Fr::mul(ll, Fr::sub(x1, resx)));
      rest[5] = Fr::add(Fr::mul(lf, Fr::sub(y1, seedy)), Fr::mul(ll, Fr::sub(y1, resy)));
      rest[6] = Fr::add(Fr::mul(lf, ip), Fr::mul(ll, Fr::sub(ip, one)));
      H256 aggz = lin_zw; for (int i = 0; i < 7; i++) aggz = Fr::add(aggz, Fr::mul(al[i], rest[i]));
      H256 zk = one; for (int j = 0; j < 3; j++) zk = Fr::mul(zk, Fr::sub(zeta, wz[j]));
      const H256 qz = Fr::mul(Fr::mul(aggz, zk), i3);
      H256 vagg = Fr::mul(nu[7], qz); for (int i = 0; i < 7; i++) vagg = Fr::add(vagg, Fr::mul(nu[i], ev[i]));
      const H256 a_coef = S::A_KIND == 1 ? Fr::neg(fr_small<F>(5)) : S::A_KIND == 2 ? Fr::neg(one) : one;
      const H256 k1 = Fr::add(Fr::mul(b, Fr::add(Fr::mul(y1, y2), Fr::mul(a_coef, Fr::mul(x1, x2)))), omb);
      const H256 k2 = Fr::add(Fr::mul(b, Fr::sub(Fr::mul(x1, y2), Fr::mul(x2, y1))), omb);
      const H256 zw = Fr::mul(zeta, su->w);
      // randomisers r1, r2 (128 bits each)
      H256 r1 = Fr::to_mont(H256{{0, 0, 0, 0}}), r2 = r1;
      { H256 a = {{0, 0, 0, 0}}, c = {{0, 0, 0, 0}}; memcpy(a.l, &rnd[32 * it], 16); memcpy(c.l, &rnd[32 * it + 16], 16); r1 = Fr::to_mont(a); r2 = Fr::to_mont(c); }
      if (n == 1) r1 = one;
      const size_t o = 10 * it;
      for (int j = 0; j < 3; j++) put(b1, s1, o + j, fx[j], Fr::mul(r1, nu[j]));
      put(b1, s1, o + 3, C[0], Fr::mul(r1, nu[3]));
      put(b1, s1, o + 4, C[1], Fr::add(Fr::mul(r1, nu[4]), Fr::mul(r2, Fr::mul(nl, al[0]))));
      put(b1, s1, o + 5, C[2], Fr::add(Fr::mul(r1, nu[5]), Fr::mul(r2, Fr::mul(nl, Fr::mul(al[1], k1)))));
      put(b1, s1, o + 6, C[3], Fr::add(Fr::mul(r1, nu[6]), Fr::mul(r2, Fr::mul(nl, Fr::mul(al[2], k2)))));
      put(b1, s1, o + 7, Cq, Fr::mul(r1, nu[7]));
      put(b1, s1, o + 8, pi1, Fr::mul(r1, zeta)); put(b1, s1, o + 9, pi2, Fr::mul(r2, zw));
      gsc[it] = Fr::add(Fr::mul(r1, vagg), Fr::mul(r2, lin_zw));
      put(b2, s2, 2 * it, pi1, Fr::neg(r1)); put(b2, s2, 2 * it + 1, pi2, Fr::neg(r2));
    }, 4);
    if (status != AVRF_OK) return status;
    lap("decode + transcripts (host)");
    // ONE independent verification is a batch of one: its two 11- / 2-term sums go through the MSM engine and the pairing
    // check runs on the host (2.4 ms; the per-item device form below is paced by lone 381-bit double-and-add chains: 5 ms)
    if (each_status && n == 1 && item_st[0]) { each_status[0] = item_st[0]; return AVRF_OK; }
    if (each_status && n > 1) {
      // 13 (base, scalar) pairs per item: its 10 terms of the first argument, the g1 term, its 2 terms of the second
      const size_t e1 = 2 * FQB, TPI = 13;
      std::vector<uint8_t> bb(n * TPI * e1); std::vector<H256> ss(n * TPI);
      for (size_t it = 0; it < n; it++) {
        memcpy(&bb[(it * TPI) * e1], &b1[(10 * it) * e1], 10 * e1);
        for (int j = 0; j < 10; j++) ss[it * TPI + j] = s1[10 * it + j];
        memcpy(&bb[(it * TPI + 10) * e1], su->g1_0.xy, e1); ss[it * TPI + 10] = Fr::from_mont(Fr::neg(gsc[it]));
        memcpy(&bb[(it * TPI + 11) * e1], &b2[(2 * it) * e1], 2 * e1);
        ss[it * TPI + 11] = s2[2 * it]; ss[it * TPI + 12] = s2[2 * it + 1];
      }
      const bool host_pair = few_checks(n);
      if (!host_pair) ensure_pairing(su);
      const size_t nb = n * TPI;
      const size_t pb = (nb * e1 + 255) / 256 * 256, sb = (nb * 32 + 255) / 256 * 256, ob = (n * 2 * e1 + 255) / 256 * 256, kb = (n * 4 + 255) / 256 * 256;
      uint8_t *base = (uint8_t *)dev_scratch(su, 1, 2 * pb + sb + ob + 2 * kb + 256);
      uint8_t *d_xy = base; uint32_t *d_b = (uint32_t *)(base + pb), *d_s = (uint32_t *)(base + 2 * pb), *d_pts = (uint32_t *)(base + 2 * pb + sb);
      int32_t *d_ok = (int32_t *)(base + 2 * pb + sb + ob), *d_rec = (int32_t *)(base + 2 * pb + sb + ob + kb);
      uint32_t *d_flag = (uint32_t *)(base + 2 * pb + sb + ob + 2 * kb);
      HIP_CHECK(hipMemcpyAsync(d_xy, bb.data(), nb * e1, hipMemcpyHostToDevice, su->stream));
      if (su->curve == 0) parallel_for(n, [&](size_t it) { for (size_t j = 0; j < TPI; j++) g1_glv_split_bls(ss[it * TPI + j].l); });   // k -> (k mod z^2, k div z^2)
      HIP_CHECK(hipMemcpyAsync(d_s, ss.data(), nb * 32, hipMemcpyHostToDevice, su->stream));
      HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, su->stream)); HIP_CHECK(hipMemsetAsync(d_rec, 0, n * 4, su->stream));
      launch_g1_bases(su->curve, d_xy, nb, d_b, d_flag, su->stream);
      if (!host_subgroup) launch_g1_subgroup_check(su->curve, d_b, nb, d_flag, su->stream, d_rec, (uint32_t)TPI);
      launch_g1_lincomb(su->curve, d_b, d_s, n, (uint32_t)TPI, 11, d_pts, su->stream, su->curve == 0);
      std::vector<int32_t> okv(n), rec(n); uint32_t range_flag = 0;
      std::vector<uint32_t> pts(host_pair ? n * 2 * e1 / 4 : 0);
      if (host_pair) HIP_CHECK(hipMemcpyAsync(pts.data(), d_pts, n * 2 * e1, hipMemcpyDeviceToHost, su->stream));
      else {
        launch_pairing_check(su->ptab, d_pts, n, d_ok, su->stream);
        HIP_CHECK(hipMemcpyAsync(okv.data(), d_ok, n * 4, hipMemcpyDeviceToHost, su->stream));
      }
      HIP_CHECK(hipMemcpyAsync(rec.data(), d_rec, n * 4, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipMemcpyAsync(&range_flag, d_flag, 4, hipMemcpyDeviceToHost, su->stream));
      HIP_CHECK(hipStreamSynchronize(su->stream)); HIP_CHECK(hipGetLastError());
      lap(host_pair ? "per-item G1 sums (device)" : "per-item G1 sums + pairing checks (device)");
      if (range_flag & 1) return AVRF_INVALID_DATA;      // a base with a coordinate >= p (launch_g1_bases): no item to pin it on
      if (host_pair) { host_pair_checks(su, pts.data(), n, okv.data()); lap("pairing checks (host pool)"); }
      for (size_t it = 0; it < n; it++)
        each_status[it] = item_st[it] ? item_st[it] : rec[it] ? AVRF_INVALID_DATA : okv[it] ? AVRF_OK : AVRF_VERIFICATION_FAILURE;
      return AVRF_OK;
    }
    H256 g1_scalar = {{0, 0, 0, 0}};
    for (size_t it = 0; it < n; it++) g1_scalar = Fr::sub(g1_scalar, gsc[it]);
    put(b1, s1, 10 * n, su->g1_0, g1_scalar);
    // b1 holds every deserialised G1 point of the batch (ring commitments, proof commitments, opening proofs): its bases are
    // subgroup-checked on the device before they are used (ark-serialize Validate::Yes; BLS12-381 G1 has a large cofactor)
    bool bad = false;
    G1Aff acc1, acc2;
    g1_msm2(su, b1, s1, b2, s2, !host_subgroup, &bad, &acc1, &acc2);
    if (bad) { if (each_status) { each_status[0] = AVRF_INVALID_DATA; return AVRF_OK; } return AVRF_INVALID_DATA; }
    lap("two G1 MSMs (device, one chain)");
    const typename HP::G2Lines *lines = host_lines(su);
    QEl px[2], py[2]; bool pinf[2] = {acc1.inf, acc2.inf};
    const G1Aff *accs[2] = {&acc1, &acc2};
    for (int i = 0; i < 2; i++) { QEl x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y); memcpy(x.l, accs[i]->xy, FQB); memcpy(y.l, accs[i]->xy + FQB, FQB); px[i] = FqN::to_mont(x); py[i] = FqN::to_mont(y); }
    const bool ok = HP::product_is_one_lines_par(px, py, pinf, lines, 2, [](size_t k, auto fn) { parallel_for(k, fn); });   // the two Miller loops side by side
    lap("2-pairing check (host)");
    if (each_status) { each_status[0] = ok ? AVRF_OK : AVRF_VERIFICATION_FAILURE; return AVRF_OK; }
    return ok ? AVRF_OK : AVRF_VERIFICATION_FAILURE;
  }
};

using RingB = Ring<SuiteBandersnatch, G1Bls12381>;
using RingJ = Ring<SuiteBabyJubJub, G1Bn254>;
using RingK = Ring<SuiteJubJub, G1Bls12381>;        // JubJub-SHA512-TAI over BLS12-381 (src/suites/jubjub.rs:76-95)
using RingX = Ring<SuiteBandersnatchShake, G1Bls12381>;   // Bandersnatch with the SHAKE128 transcript (src/suites/bandersnatch_shake128.rs)
using RingW = Ring<SuiteBandersnatchSW, G1Bls12381>;   // Bandersnatch-SW: the ring proof runs on the TEMapping of the keys (src/ring.rs:75-81)
template <class R> struct RingTag { using type = R; };
template <class F> static auto with_ring(int suite, F &&f) {
  switch (suite) { case 1: return f(RingTag<RingJ>{}); case 2: return f(RingTag<RingK>{}); case 4: return f(RingTag<RingW>{}); case 5: return f(RingTag<RingX>{}); default: return f(RingTag<RingB>{}); }
}

}  // namespace

template <class F> static int guarded(F f) {
  try { return f(); }
  catch (const avrf::HipFailure &e) { fprintf(stderr, "avrf: HIP error %s at %s:%d\n", hipGetErrorString(e.err), e.file, e.line); return AVRF_ERR_NO_DEVICE; }
  catch (const std::bad_alloc &) { return AVRF_ERR_NO_DEVICE; }
}

extern "C" {

int avrf_ring_setup_load(avrf_ctx *ctx, const uint8_t *srs, size_t srs_len, size_t ring_size, avrf_ring_setup **out) {
  if (!ctx || !srs || !out || ring_size == 0) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(ctx)) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (!ring_suite(avrf_ctx_suite_(ctx))) return AVRF_ERR_BAD_ARG;      // not a RingSuite (Ed25519)
  return guarded([&] { return with_ring(avrf_ctx_suite_(ctx), [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::setup_load(ctx, srs, srs_len, ring_size, out); }); });
}
int avrf_ring_srs_generate(avrf_ctx *ctx, const uint8_t *tau, const uint8_t *g1, const uint8_t *g2, size_t n_g1, uint8_t *out, size_t out_cap, size_t *out_len) {
  if (!ctx || !tau || !g1 || !g2 || !ring_suite(avrf_ctx_suite_(ctx))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(ctx)) return AVRF_ERR_BAD_ARG;
  return guarded([&] { return with_ring(avrf_ctx_suite_(ctx), [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::srs_generate(ctx, tau, g1, g2, n_g1, out, out_cap, out_len); }); });
}
int avrf_ring_setup_from_seed(avrf_ctx *ctx, const uint8_t *seed, size_t ring_size, avrf_ring_setup **out) {
  if (!ctx || !seed || !out || !ring_suite(avrf_ctx_suite_(ctx))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(ctx)) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  return guarded([&] { return with_ring(avrf_ctx_suite_(ctx), [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::setup_from_seed(ctx, seed, ring_size, out); }); });
}
size_t avrf_ring_pcs_domain_size(int suite, size_t ring_size) {       /* pcs_domain_size, src/ring.rs:810-817: 3 * piop_domain + 1 */
  const size_t L = with_suite(suite, [&](auto tag) { using S = typename decltype(tag)::type; return (size_t)S::Fr::BITS; });
  size_t need = ring_size + 4 + L, N = 1; while (N < need) N <<= 1;
  return 3 * N + 1;
}
void avrf_ring_setup_free(avrf_ring_setup *su) {
  if (!su) return;
  (void)hipSetDevice(su->device);
  void *d[] = {su->d_srs, su->d_tw_n, su->d_tw_n_inv, su->d_tw_4n, su->d_tw_4n_inv, su->d_buf, su->d_l4, su->d_srs_table, su->d_wit_table, su->d_wit_bases,
               su->d_scr[0], su->d_scr[1], su->d_scr[2], su->d_scr[3], su->d_scr[4], su->d_scr[5]};
  for (void *p : d) if (p) (void)hipFree(p);
  su->ws.release();
  su->ptab.release();
  if (su->host_lines && su->host_lines_free) su->host_lines_free(su->host_lines);
  if (avrf_ring_setup *l = su->lane1) {                                // only what the lane owns
    void *o[] = {l->d_buf, l->d_scr[0], l->d_scr[1], l->d_scr[2], l->d_scr[3], l->d_scr[4], l->d_scr[5]};
    for (void *p : o) if (p) (void)hipFree(p);
    l->ws.release();
    (void)hipStreamDestroy(l->stream);
    delete l;
  }
  delete su;
}
size_t avrf_ring_max_ring_size(const avrf_ring_setup *su) { return su ? su->keyset : 0; }
size_t avrf_ring_domain_size(const avrf_ring_setup *su) { return su ? su->N : 0; }
size_t avrf_ring_proof_len(const avrf_ring_setup *su) { return su ? (su->curve == 0 ? 592 : 480) : 0; }
size_t avrf_ring_commitment_len(const avrf_ring_setup *su) { return su ? (su->curve == 0 ? 144 : 96) : 0; }
int avrf_ring_setup_suite(const avrf_ring_setup *su) { return su ? su->suite : -1; }
avrf_ring_setup *avrf_ring_key_setup(const avrf_ring_key *key) { return key ? key->setup : nullptr; }
int avrf_ring_setup_plan(const avrf_ring_setup *su, int32_t out[4]) {
  if (!su || !out) return AVRF_ERR_BAD_ARG;
  out[0] = su->table_c; out[1] = su->table_nwin; out[2] = su->wit_c; out[3] = su->wit_nwin;   // the witness table is built on first proof
  // once a batched prove call has built the tables of all multiples, those are what the commitments of a batch run over
  if (su->direct) { out[0] = su->direct->t.c; out[1] = su->direct->t.rows; }
  if (su->direct_wit) { out[2] = su->direct_wit->t.c; out[3] = su->direct_wit->t.rows; }
  return AVRF_OK;
}

int avrf_ring_index(avrf_ring_setup *su, const uint8_t *pks_xy, size_t n_keys, avrf_ring_key **out, uint8_t *commitment_out) {
  if (!su || !out || (n_keys && !pks_xy)) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (!su->n_srs) return AVRF_SRS_LOOKUP_FAILED;                       // verifier-only setup
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  int st = guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::index(su, pks_xy, n_keys, out); }); });
  if (st == AVRF_OK && commitment_out) {
    std::vector<uint8_t> b;
    for (int i = 0; i < 3; i++) { if (su->curve == 0) g1_encode<G1Bls12381>((*out)->C[i], true, b); else g1_encode<G1Bn254>((*out)->C[i], true, b); }
    memcpy(commitment_out, b.data(), b.size());
  }
  return st;
}
void avrf_ring_key_free(avrf_ring_key *k) { if (!k) return; if (k->d_fixed4) (void)hipFree(k->d_fixed4); if (k->d_fixed_coef) (void)hipFree(k->d_fixed_coef); if (k->d_points_pre) (void)hipFree(k->d_points_pre); delete k; }

int avrf_ring_vk_builder_new(avrf_ring_setup *su, avrf_ring_vk_builder **out) {
  if (!su || !out) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (!su->n_srs) return AVRF_SRS_LOOKUP_FAILED;
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  return guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::builder_new(su, out); }); });
}
void avrf_ring_vk_builder_free(avrf_ring_vk_builder *b) { delete b; }
size_t avrf_ring_vk_builder_free_slots(const avrf_ring_vk_builder *b) { return b ? b->setup->keyset - b->curr : 0; }
int avrf_ring_vk_builder_append(avrf_ring_vk_builder *b, const uint8_t *pks_xy, size_t n) {
  if (!b || (n && !pks_xy)) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(b->setup->ctx)) return AVRF_ERR_BAD_ARG;
  if (hipSetDevice(b->setup->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  return guarded([&] { return with_ring(b->setup->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::builder_append(b, pks_xy, n); }); });
}
int avrf_ring_vk_builder_finalize(const avrf_ring_vk_builder *b, uint8_t *commitment_out) {
  if (!b || !commitment_out) return AVRF_ERR_BAD_ARG;
  std::vector<uint8_t> o;
  for (int i = 0; i < 3; i++) { if (b->setup->curve == 0) g1_encode<G1Bls12381>(b->C[i], true, o); else g1_encode<G1Bn254>(b->C[i], true, o); }
  memcpy(commitment_out, o.data(), o.size());
  return AVRF_OK;
}

// The table of all multiples of this setup's SRS: found in the registry or built (once per device and SRS, ~2 s), sized to the HBM that
// is free -- the widest window c <= 16 whose table fits min(what is left of AVRF_RING_TABLE_GB (default 232, both tables together), free - 40 GB); no table when even the
// bucket form's own width does not fit, when AVRF_RING_DIRECT=0, or when the allocation fails (the bucket form then runs as before).
static void ensure_direct(avrf_ring_setup *su) {
  if (su->direct_tried || !su->n_srs) return;
  su->direct_tried = true;
  if (const char *e = getenv("AVRF_RING_DIRECT")) if (atoi(e) == 0) return;
  with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; R_::ensure_lagrange(su); return 0; });   // the witness bases
  std::lock_guard<std::mutex> lk(g_direct_mu);
  double budget_gb = 232.0;
  if (const char *e = getenv("AVRF_RING_TABLE_GB")) budget_gb = atof(e);
  // kind 0 over the SRS powers, then kind 1 over the witness bases with what is left of the budget
  for (int kind = 0; kind < 2; kind++) {
    std::shared_ptr<DirectEntry> &slot = kind ? su->direct_wit : su->direct;
    const size_t nb = kind ? 2 * su->N + 1 : su->n_srs;
    const uint32_t *bases = kind ? su->d_wit_bases : su->d_srs;
    const int c_min = kind ? su->wit_c : su->table_c;
    if (!bases) continue;
    if (kind) if (const char *e = getenv("AVRF_RING_DIRECT_WIT")) if (atoi(e) == 0) continue;   // (A/B knob: the witness commits stay on the bucket form)
    for (auto it = g_direct.begin(); it != g_direct.end();) {
      std::shared_ptr<DirectEntry> e = it->lock();
      if (!e) { it = g_direct.erase(it); continue; }
      if (e->device == su->device && e->kind == kind && e->t.curve == su->curve && e->t.n == nb && e->srs_key == su->g1_raw) { slot = e; break; }
      ++it;
    }
    if (slot) { budget_gb -= slot->t.bytes * 1e-9; continue; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    double usable = (double)free_b - 40e9; if (usable > budget_gb * 1e9) usable = budget_gb * 1e9;
    int c = 0;
    for (int cc = 16; cc >= c_min && cc >= 8; cc--) {
      G1DirectTable shape;
      if ((double)g1_direct_table_shape(su->curve, nb, cc, &shape) <= usable && shape.points < 0x7fffffffull) { c = cc; break; }
    }
    if (!c) continue;
    auto e = std::make_shared<DirectEntry>();
    e->device = su->device; e->kind = kind; e->srs_key = su->g1_raw;
    try { build_g1_direct_table(su->curve, bases, nb, c, &e->t, su->stream); }
    catch (const HipFailure &) { (void)hipGetLastError(); continue; }       // (the entry's destructor frees what was allocated)
    g_direct.push_back(e);
    slot = e;
    budget_gb -= e->t.bytes * 1e-9;
  }
  if (su->lane1) { su->lane1->direct = su->direct; su->lane1->direct_wit = su->direct_wit; su->lane1->direct_tried = true; su->lane1->d_wit_bases = su->d_wit_bases; }
}

int avrf_ring_prove(avrf_ring_key *k, size_t n, const uint32_t *key_index, const uint8_t *blindings, int blinding_mode, uint8_t *proofs_out) {
  if (!k || (n && (!key_index || !blindings || !proofs_out))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(k->setup->ctx)) return AVRF_ERR_BAD_ARG;
  if (blinding_mode != 0 && blinding_mode != 1) return AVRF_ERR_BAD_ARG;
  if (hipSetDevice(k->setup->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  const size_t plen = k->setup->curve == 0 ? 592 : 480;
  // proofs proved in lockstep per device round: at most 512, and a call of up to 1024 proofs is cut in TWO so that both lanes
  // work (the host rounds of one chunk hide behind the device rounds of the other) -- a rank of an 8-GPU node proves 128 proofs
  // per context (BASELINE configs[3]: 4096 proofs / 8 ranks / 4 contexts), which as ONE chunk on one lane left the device
  // idle during every transcript round: 8.99 -> 10.0 k proofs/s for 512 proofs on 4 contexts with 2 host cores
  size_t chunk = 512;
  if (n >= 32 && n <= 1024) chunk = (n + 1) / 2;
  if (const char *e = getenv("AVRF_RING_CHUNK")) { long v = atol(e); if (v >= 1 && v <= 4096) chunk = (size_t)v; }   // (test hook: re-read per call)
  avrf_ring_setup *su = k->setup;
  if (n >= 64) if (int st = guarded([&] { ensure_direct(su); return (int)AVRF_OK; })) return st;
  auto run = [&](avrf_ring_setup *lane, size_t i) {
    const size_t m = n - i < chunk ? n - i : chunk;
    return guarded([&] {
      return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::prove_chunk(k, lane, m, key_index + i, blindings + 32 * i, blinding_mode == 1, proofs_out + plen * i); }); });
  };
  const char *le = getenv("AVRF_RING_LANES"); const bool one_lane = le && atoi(le) == 1;
  if (n <= chunk || one_lane) {
    for (size_t i = 0; i < n; i += chunk) if (int st = run(su, i)) return st;
    return AVRF_OK;
  }
  // two chunks in flight: chunks alternate between the setup's own stream/scratch and its second lane
  avrf_ring_setup *lanes[2] = {su, nullptr};
  if (int st = guarded([&] { lanes[1] = with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::second_lane(su); }); return (int)AVRF_OK; })) return st;
  std::atomic<int> status{AVRF_OK};
  std::thread th[2];
  for (int t = 0; t < 2; t++) th[t] = std::thread([&, t] {
    if (hipSetDevice(su->device) != hipSuccess) { status = AVRF_ERR_NO_DEVICE; return; }
    for (size_t i = (size_t)t * chunk; i < n && status == AVRF_OK; i += 2 * chunk) if (int st = run(lanes[t], i)) status = st;
  });
  for (auto &x : th) x.join();
  return status;
}

static int copy_out(const std::vector<uint8_t> &v, uint8_t *out, size_t cap, size_t *out_len) {
  if (out_len) *out_len = v.size();
  if (!out || cap < v.size()) return AVRF_ERR_BAD_ARG;
  memcpy(out, v.data(), v.size());
  return AVRF_OK;
}
int avrf_ring_verifier_setup_load(avrf_ctx *ctx, const uint8_t *params, size_t params_len, size_t ring_size, avrf_ring_setup **out) {
  if (!ctx || !params || !out || ring_size == 0) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(ctx)) return AVRF_ERR_BAD_ARG;
  *out = nullptr;
  if (!ring_suite(avrf_ctx_suite_(ctx))) return AVRF_ERR_BAD_ARG;
  return guarded([&] { return with_ring(avrf_ctx_suite_(ctx), [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::verifier_setup_load(ctx, params, params_len, ring_size, out); }); });
}
int avrf_ring_pcs_verifier_params_serialize(avrf_ring_setup *su, int compress, uint8_t *out, size_t out_cap, size_t *out_len) {
  if (!su) return AVRF_ERR_BAD_ARG;
  std::vector<uint8_t> v;
  int st = guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::verifier_params_serialize(su, compress != 0, v); }); });
  return st ? st : copy_out(v, out, out_cap, out_len);
}
int avrf_ring_setup_serialize(avrf_ring_setup *su, int compress, uint8_t *out, size_t out_cap, size_t *out_len) {
  if (!su) return AVRF_ERR_BAD_ARG;
  if (!su->n_srs) return AVRF_SRS_LOOKUP_FAILED;
  std::vector<uint8_t> v;
  int st = guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::setup_serialize(su, compress != 0, v); }); });
  return st ? st : copy_out(v, out, out_cap, out_len);
}
int avrf_ring_builder_params_serialize(avrf_ring_setup *su, int compress, uint8_t *out, size_t out_cap, size_t *out_len) {
  if (!su) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  if (!su->n_srs) return AVRF_SRS_LOOKUP_FAILED;
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  std::vector<uint8_t> v;
  int st = guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::builder_params_serialize(su, compress != 0, v); }); });
  return st ? st : copy_out(v, out, out_cap, out_len);
}

int avrf_ring_verify_each(avrf_ring_setup *su, size_t n, const uint8_t *ring_commitments, size_t n_rings, const uint32_t *ring_of_item,
                          const uint8_t *instances_xy, const uint8_t *ring_proofs, int32_t *status_out) {
  if (!su || (n && (!ring_commitments || !n_rings || !instances_xy || !ring_proofs || !status_out))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  return guarded([&] {
    return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::verify_batch(su, n, ring_commitments, ring_of_item, n_rings, instances_xy, ring_proofs, status_out); }); });
}

int avrf_ring_pairing_check(avrf_ring_setup *su, size_t n, const uint8_t *a_xy, const uint8_t *b_xy, int32_t *ok_out) {
  if (!su || (n && (!a_xy || !b_xy || !ok_out))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  return guarded([&] { return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::pairing_check(su, n, a_xy, b_xy, ok_out); }); });
}

int avrf_ring_batch_verify(avrf_ring_setup *su, size_t n, const uint8_t *ring_commitments, size_t n_rings, const uint32_t *ring_of_item,
                           const uint8_t *instances_xy, const uint8_t *ring_proofs) {
  if (!su || (n && (!ring_commitments || !n_rings || !instances_xy || !ring_proofs))) return AVRF_ERR_BAD_ARG;
  if (avrf_ctx_busy_(su->ctx)) return AVRF_ERR_BAD_ARG;
  if (hipSetDevice(su->device) != hipSuccess) return AVRF_ERR_NO_DEVICE;
  return guarded([&] {
    return with_ring(su->suite, [&](auto r_) { using R_ = typename decltype(r_)::type; return R_::verify_batch(su, n, ring_commitments, ring_of_item, n_rings, instances_xy, ring_proofs); }); });
}

}  // extern "C"
