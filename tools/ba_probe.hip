// tools/ba_probe.hip -- probe for batch-affine bucket accumulation on gfx950 (BLS12-381 G1): pairs of affine points are
// added with ONE field inversion per lane shared by K additions (Montgomery's trick; 5M + 1S per addition plus the share of
// the inversion) against the XYZZ mixed addition of k_accumulate (8M + 2S).  Prints the cost per addition of both.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed -I ark_vrf_amd/csrc -o /tmp/ba_probe tools/ba_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../ark_vrf_amd/csrc/te.h"
#include "../ark_vrf_amd/csrc/curves.h"
#include "batch_affine.h"

using namespace avrf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using C = G1Bls12381; using CV = G1Curve<C>; using Fq = C::Fq; constexpr int N = Fq::N;

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_gen(uint32_t *tab, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int w = 0; w < 2 * N; w++) tab[(size_t)i * 2 * N + w] = mix(i * 64 + w + 12345);
  tab[(size_t)i * 2 * N + N - 1] &= 0x0fffffffu; tab[(size_t)i * 2 * N + 2 * N - 1] &= 0x0fffffffu;
  if (i % 1000 == 7) for (int w = 0; w < 2 * N; w++) tab[(size_t)i * 2 * N + w] = 0;      // a few points at infinity
}
__global__ void k_idx(uint32_t *idx, uint32_t m, uint32_t T) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint32_t h = mix(i * 2 + 1);
  idx[i] = (h % T) | (mix(h) & 0x80000000u);
  if (i % 513 == 5) idx[i] = 0xffffffffu;                                                   // padding entries
}

__global__ void __launch_bounds__(256, 2) k_chain(const uint32_t *tab, const uint32_t *idx, uint32_t per, uint32_t lanes, uint32_t *out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  CV::acc_t acc = CV::identity();
  for (uint32_t i = 0; i < per; i++) {
    uint32_t e = idx[(size_t)t * per + i];
    if (e == 0xffffffffu) continue;
    acc = CV::madd(acc, CV::load_base(tab + (size_t)(e & 0x7fffffffu) * 2 * N), (e & 0x80000000u) != 0);
  }
  CV::store_acc(out + (size_t)t * 4 * N, acc);
}

__global__ void __launch_bounds__(256) k_inv_check(const uint32_t *tab, uint32_t n, uint32_t *bad, int which, uint32_t *sink) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fpn<N> a = fn_load<N>(tab + (size_t)i * 2 * N);
  if (fn_is_zero(a)) return;
  fpn<N> r;
  if (which == 0) r = fn_inv<Fq>(a);
  else if (which == 1) r = fn_inv_gcd<Fq>(a);
  else { r = fn_inv_gcd<Fq>(a); if (!fn_eq(fn_mul<Fq>(r, a), fn_one<Fq>())) atomicAdd(bad, 1u); }
  sink[i] = r.v[0];
}

// reference for pair p: XYZZ mixed addition, normalised with the Fermat inversion
__global__ void __launch_bounds__(64) k_check(const uint32_t *tab, const uint32_t *idx, const uint32_t *dst, uint32_t ncheck, uint32_t *bad, const uint32_t *flagged) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ncheck) return;
  bool ia, ib;
  CV::base_t A = ba_load_entry<C>(tab, idx[2 * p], ia), B = ba_load_entry<C>(tab, idx[2 * p + 1], ib);
  CV::acc_t r = CV::madd(CV::from_affine(A), B, false);
  fpn<N> x = fn_zero<N>(), y = fn_zero<N>();
  if (!CV::is_identity(r)) { x = fn_mul<Fq>(r.x, fn_inv<Fq>(r.zz)); y = fn_mul<Fq>(r.y, fn_inv<Fq>(r.zzz)); }
  fpn<N> gx = fn_load<N>(dst + (size_t)p * 2 * N), gy = fn_load<N>(dst + (size_t)p * 2 * N + N);
  if (!fn_eq(x, gx) || !fn_eq(y, gy)) { if (!(!ia && !ib && fn_eq(A.x, B.x))) atomicAdd(bad, 1u); }
}

int main(int argc, char **argv) {
  const uint32_t T = 135190;                    // table entries (22 windows x 6145 bases)
  const uint32_t npairs = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 24);
  uint32_t *tab, *idx, *dst, *scr, *flag, *out, *sink;
  CK(hipMalloc(&tab, (size_t)T * 2 * N * 4)); CK(hipMalloc(&idx, (size_t)npairs * 2 * 4));
  CK(hipMalloc(&dst, (size_t)npairs * 2 * N * 4)); CK(hipMalloc(&scr, (size_t)npairs * N * 4 + 65536 * 256 * N * 4));
  CK(hipMalloc(&flag, 16)); CK(hipMalloc(&out, (size_t)(npairs / 16) * 4 * N * 4)); CK(hipMalloc(&sink, (size_t)T * 4));
  hipLaunchKernelGGL(k_gen, dim3((T + 255) / 256), dim3(256), 0, 0, tab, T);
  hipLaunchKernelGGL(k_idx, dim3((2 * npairs + 255) / 256), dim3(256), 0, 0, idx, 2 * npairs, T);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // inversions
  for (int which = 0; which < 3; which++) {
    CK(hipMemset(flag, 0, 16));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_inv_check, dim3((T + 255) / 256), dim3(256), 0, 0, tab, T, flag, which, sink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t bad; CK(hipMemcpy(&bad, flag, 4, hipMemcpyDeviceToHost));
    printf("inversion %s: %u values %.3f ms  (bad %u)\n", which == 0 ? "fermat" : which == 1 ? "gcd" : "gcd+check", T, ms, bad);
  }
  // baseline chains: 2 * npairs entries
  for (uint32_t per : {32u, 64u, 128u}) {
    uint32_t lanes = 2 * npairs / per;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chain, dim3((lanes + 255) / 256), dim3(256), 0, 0, tab, idx, per, lanes, out);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("madd chain per=%u: %u additions in %.3f ms -> %.2f G add/s\n", per, 2 * npairs, ms, 2.0 * npairs / ms * 1e-6);
  }
  for (uint32_t K : {16u, 32u, 64u, 128u, 256u}) {
    const uint32_t waves = (uint32_t)(((size_t)npairs + 64 * K - 1) / (64 * K));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(flag, 0, 16));
      CK(hipEventRecord(e0));
      ba_launch_round<C>(tab, idx, npairs, K, dst, scr, flag, 0);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    uint32_t fl; CK(hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost));
    CK(hipMemset(flag + 1, 0, 4));
    const uint32_t ncheck = npairs < 200000 ? npairs : 200000;
    hipLaunchKernelGGL(k_check, dim3((ncheck + 63) / 64), dim3(64), 0, 0, tab, idx, dst, ncheck, flag + 1, flag);
    CK(hipDeviceSynchronize());
    uint32_t bad; CK(hipMemcpy(&bad, flag + 1, 4, hipMemcpyDeviceToHost));
    printf("batch-affine K=%u: %u additions (%u waves) in %.3f ms -> %.2f G add/s  (flag %u, mismatches %u of %u)\n", K, npairs, waves, ms,
           (double)npairs / ms * 1e-6, fl, bad, ncheck);
  }
  // second round: dense input (the first round's output)
  {
    const uint32_t np2 = npairs / 2, K = 64;
    uint32_t *dst2; CK(hipMalloc(&dst2, (size_t)np2 * 2 * N * 4));
    CK(hipEventRecord(e0));
    ba_launch_round<C>(dst, nullptr, np2, K, dst2, scr, flag, 0);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("batch-affine dense K=%u: %u additions in %.3f ms -> %.2f G add/s\n", K, np2, ms, (double)np2 / ms * 1e-6);
  }
  return 0;
}
