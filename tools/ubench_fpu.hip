// tools/ubench_fpu.hip -- the unsaturated-limb field / point arithmetic (ark_vrf_amd/csrc/fpu*.h) against the saturated
// forms it replaces inside k_accumulate: agreement on the device (chains of mixed additions compared as projective points)
// and throughput per chip at 1 .. 8 waves per SIMD.  The gate of round 5: fu_mul >= 150 G/s (fp_mul: 126), teu_madd >= 17 G/s
// (te_madd: 14.4), fu_mul<FqBls12381> >= 70 G/s (fn_mul: 58.8).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -enable-ipra=0 -Iark_vrf_amd/csrc -o /tmp/ubench_fpu tools/ubench_fpu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../ark_vrf_amd/csrc/fpu_te.h"
#include "../ark_vrf_amd/csrc/curves.h"

using namespace avrf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class F> __global__ void k_fumul(uint32_t *out, int iters, uint32_t seed) {
  constexpr int L = UL<F>::L;
  fu<L> a, b;
  for (int i = 0; i < L; i++) { a.v[i] = (int32_t)((seed * (i + 1) + threadIdx.x) & UL<F>::MASK); b.v[i] = (int32_t)((seed * (i + 7) + blockIdx.x) & UL<F>::MASK); }
  a.v[L - 1] &= 0xffff; b.v[L - 1] &= 0xffff;
  for (int i = 0; i < iters; i++) { a = fu_mul<F>(a, b); b = fu_mul<F>(b, a); }
  uint32_t r = 0;
  for (int i = 0; i < L; i++) r ^= (uint32_t)a.v[i] ^ (uint32_t)b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> __global__ void k_fusqr(uint32_t *out, int iters, uint32_t seed) {
  constexpr int L = UL<F>::L;
  fu<L> a, b;
  for (int i = 0; i < L; i++) { a.v[i] = (int32_t)((seed * (i + 1) + threadIdx.x) & UL<F>::MASK); b.v[i] = (int32_t)((seed * (i + 7) + blockIdx.x) & UL<F>::MASK); }
  a.v[L - 1] &= 0xffff; b.v[L - 1] &= 0xffff;
  b.v[0] ^= (int32_t)(threadIdx.x & 15);
  for (int i = 0; i < iters; i++) { a = fu_sqr<F>(a); b = fu_sqr<F>(b); }
  uint32_t r = 0;
  for (int i = 0; i < L; i++) r ^= (uint32_t)a.v[i] ^ (uint32_t)b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> __global__ void k_fmul_sat(uint32_t *out, int iters, uint32_t seed) {
  constexpr int N = UL<F>::N;
  fpn<N> a, b;
  for (int i = 0; i < N; i++) { a.v[i] = seed * (i + 1) + threadIdx.x; b.v[i] = seed * (i + 7) + blockIdx.x; }
  a.v[N - 1] &= 0x00ffffff; b.v[N - 1] &= 0x00ffffff;
  for (int i = 0; i < iters; i++) {
    mont_mul_ps<N, F>(a.v, a.v, b.v); mont_mul_ps<N, F>(b.v, b.v, a.v);
  }
  uint32_t r = 0;
  for (int i = 0; i < N; i++) r ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// mode 0: saturated te_madd, mode 1: teu_madd; the base alternates between G and B with signs from the lane / iteration
template <class S, int MODE> __global__ void __launch_bounds__(256, 2) k_madd(uint32_t *out, int iters, uint32_t seed, int store) {
  using Fq = typename S::Fq;
  te_pre g, b;
  g.x = fp_const<Fq>(S::G_X); g.y = fp_const<Fq>(S::G_Y); g.k = fp_const<Fq>(S::G_K);
  b.x = fp_const<Fq>(S::B_X); b.y = fp_const<Fq>(S::B_Y); b.k = fp_const<Fq>(S::B_K);
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  te_ext res;
  if (MODE == 0) {
    te_ext p = te_identity<S>();
    for (int i = 0; i < iters; i++) {
      const bool neg = ((lane * 2654435761u + i * 40503u + seed) >> 13) & 1;
      te_pre q = ((lane + i) & 2) ? g : b;
      if (neg) q = te_pre_neg<S>(q);
      p = te_madd<S>(p, q);
    }
    res = p;
  } else {
    te_acc_u<S> p = teu_identity<S>();
    for (int i = 0; i < iters; i++) {
      const bool neg = ((lane * 2654435761u + i * 40503u + seed) >> 13) & 1;
      const te_pre q = ((lane + i) & 2) ? g : b;
      p = teu_madd<S>(p, q, neg);
    }
    res = teu_to_ext<S>(p);
  }
  if (store) store_ext(reinterpret_cast<te_ext *>(out) + lane, res);
  else {
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= res.x.v[i] ^ res.y.v[i] ^ res.z.v[i] ^ res.t.v[i];
    out[lane] = r + seed;
  }
}
// are two extended points the same projective point, and is each one consistent (X Y = T Z)?
template <class S> __global__ void k_same_point(const te_ext *a, const te_ext *b, uint32_t n, uint32_t *bad) {
  using Fq = typename S::Fq;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const te_ext p = load_ext(a + i), q = load_ext(b + i);
  bool ok = fp_eq(fp_mul<Fq>(p.x, q.z), fp_mul<Fq>(q.x, p.z)) && fp_eq(fp_mul<Fq>(p.y, q.z), fp_mul<Fq>(q.y, p.z));
  ok = ok && fp_eq(fp_mul<Fq>(q.x, q.y), fp_mul<Fq>(q.t, q.z)) && !fp_is_zero(q.z);
  ok = ok && !ge_p<Fq>(q.x) && !ge_p<Fq>(q.y) && !ge_p<Fq>(q.t) && !ge_p<Fq>(q.z);
  if (!ok) atomicAdd(bad, 1u);
}

// G1: mode 0 the saturated G1Curve::madd, mode 1 the k_accumulate policy (AccumG1U: g1u_madd, raw partial out, reader's conversion back)
template <class C, int MODE> __global__ void __launch_bounds__(256, 2) k_g1madd(uint32_t *out, const uint32_t *bases, int nbases, int iters, uint32_t seed, int store) {
  using CV = G1Curve<C>; using AC = AccumG1U<C>; constexpr int N = C::Fq::N;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  typename CV::acc_t res;
  if (MODE == 0) {
    typename CV::acc_t p = CV::identity();
    for (int i = 0; i < iters; i++) {
      const uint32_t h = lane * 2654435761u + i * 40503u + seed;
      p = CV::madd(p, CV::load_base(bases + (size_t)((h >> 3) % nbases) * 2 * N), (h >> 13) & 1);
    }
    res = p;
  } else {
    typename AC::acc_t p = AC::identity();
    for (int i = 0; i < iters; i++) {
      const uint32_t h = lane * 2654435761u + i * 40503u + seed;
      p = AC::madd(p, CV::load_base(bases + (size_t)((h >> 3) % nbases) * 2 * N), (h >> 13) & 1);
    }
    alignas(16) uint32_t tmp[AC::PART_WORDS];                  // through the partial-sum format, as k_accumulate -> k_bucket_sum
    AC::store_part(tmp, p);
    res = AC::load_part(tmp);
  }
  if (store) CV::store_acc(out + (size_t)lane * 4 * N, res);
  else {
    uint32_t r = 0;
    for (int i = 0; i < N; i++) r ^= res.x.v[i] ^ res.y.v[i] ^ res.zz.v[i] ^ res.zzz.v[i];
    out[lane] = r + seed;
  }
}
// table of nbases points: entry i = i G (entry 0 = infinity, entry 4 = entry 3: the exceptional cases), from an affine generator in PLAIN words
template <class C> __global__ void k_g1_table(uint32_t *bases, int nbases, const uint32_t *g) {
  using CV = G1Curve<C>; using Fq = typename C::Fq; constexpr int N = Fq::N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbases) return;
  typename CV::base_t q;
  q.x = fn_to_mont<Fq>(fn_load<N>(g)); q.y = fn_to_mont<Fq>(fn_load<N>(g + N));
  const int k = i == 4 ? 3 : i;
  typename CV::acc_t p = CV::identity();
  for (int bit = 15; bit >= 0; bit--) { p = CV::dbl(p); if ((k >> bit) & 1) p = CV::madd(p, q, false); }
  fe<Fq> x = fn_zero<N>(), y = fn_zero<N>();
  if (k) { const fe<Fq> izz = fn_inv<Fq>(p.zz), izzz = fn_inv<Fq>(p.zzz); x = fn_mul<Fq>(p.x, izz); y = fn_mul<Fq>(p.y, izzz); }
  fn_store<N>(bases + (size_t)i * 2 * N, x); fn_store<N>(bases + (size_t)i * 2 * N + N, y);
}
template <class C> __global__ void k_same_g1(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t *bad) {
  using CV = G1Curve<C>; using Fq = typename C::Fq;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto p = CV::load_acc(a + (size_t)i * CV::ACC_WORDS), q = CV::load_acc(b + (size_t)i * CV::ACC_WORDS);
  bool ok;
  if (CV::is_identity(p) || CV::is_identity(q)) ok = CV::is_identity(p) && CV::is_identity(q);
  else {
    ok = fn_eq(fn_mul<Fq>(p.x, q.zz), fn_mul<Fq>(q.x, p.zz)) && fn_eq(fn_mul<Fq>(p.y, q.zzz), fn_mul<Fq>(q.y, p.zzz));
    ok = ok && fn_eq(fn_mul<Fq>(fn_sqr<Fq>(q.zz), q.zz), fn_sqr<Fq>(q.zzz));
    ok = ok && !fn_ge_p<Fq>(q.x) && !fn_ge_p<Fq>(q.y) && !fn_ge_p<Fq>(q.zz) && !fn_ge_p<Fq>(q.zzz);
  }
  if (!ok) atomicAdd(bad, 1u);
}

template <class K> double time_kernel(K launch, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best * 1e-3;
}

template <class S> void check_te(const char *name, uint32_t *d0, uint32_t *d1, uint32_t *dbad) {
  const int blocks = 64, threads = 256, n = blocks * threads;
  for (int iters : {1, 2, 7, 40}) {
    CK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL((k_madd<S, 0>), dim3(blocks), dim3(threads), 0, 0, d0, iters, 99u, 1);
    hipLaunchKernelGGL((k_madd<S, 1>), dim3(blocks), dim3(threads), 0, 0, d1, iters, 99u, 1);
    hipLaunchKernelGGL(k_same_point<S>, dim3(blocks), dim3(threads), 0, 0, (const te_ext *)d0, (const te_ext *)d1, (uint32_t)n, dbad);
    uint32_t bad; CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    printf("teu_madd vs te_madd <%s>, %d additions per lane, %d lanes: %u mismatches\n", name, iters, n, bad);
  }
}
template <class C> void check_g1(const char *name, uint32_t *d0, uint32_t *d1, uint32_t *dbad, uint32_t *dbases, int nbases, const uint32_t *g_plain, uint32_t *dg) {
  constexpr int N = C::Fq::N;
  CK(hipMemcpy(dg, g_plain, 2 * N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_g1_table<C>, dim3((nbases + 63) / 64), dim3(64), 0, 0, dbases, nbases, (const uint32_t *)dg);
  const int blocks = 16, threads = 256, n = blocks * threads;
  for (int iters : {1, 2, 3, 7, 30}) {
    CK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL((k_g1madd<C, 0>), dim3(blocks), dim3(threads), 0, 0, d0, (const uint32_t *)dbases, nbases, iters, 99u, 1);
    hipLaunchKernelGGL((k_g1madd<C, 1>), dim3(blocks), dim3(threads), 0, 0, d1, (const uint32_t *)dbases, nbases, iters, 99u, 1);
    hipLaunchKernelGGL(k_same_g1<C>, dim3(blocks), dim3(threads), 0, 0, (const uint32_t *)d0, (const uint32_t *)d1, (uint32_t)n, dbad);
    uint32_t bad; CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    printf("g1u_madd vs G1Curve::madd <%s>, %d additions per lane over %d bases (infinity, repeats -> doublings, cancellations), %d lanes: %u mismatches\n", name, iters, nbases, n, bad);
  }
}

int main(int argc, char **argv) {
  const bool quick = argc > 1;
  uint32_t *out, *d0, *d1, *dbad, *dbases, *dg;
  CK(hipMalloc(&out, 256 * 32 * 256 * 4 * 4)); CK(hipMalloc(&d0, 64 * 256 * 192)); CK(hipMalloc(&d1, 64 * 256 * 192)); CK(hipMalloc(&dbad, 4));
  CK(hipMalloc(&dbases, 8192 * 96)); CK(hipMalloc(&dg, 96));
  uint32_t *dg381, *dg254; CK(hipMalloc(&dg381, 96)); CK(hipMalloc(&dg254, 96));
  check_te<SuiteBandersnatch>("Bandersnatch a=-5", d0, d1, dbad);
  check_te<SuiteBabyJubJub>("BabyJubJub a=1", d0, d1, dbad);
  check_te<SuiteJubJub>("JubJub a=-1", d0, d1, dbad);
  check_te<SuiteEd25519>("Ed25519 a=-1", d0, d1, dbad);
  static const uint32_t g381[24] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu, 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u,
                                    0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu, 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
  static const uint32_t g254[16] = {1, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0};
  CK(hipMemcpy(dg381, g381, 96, hipMemcpyHostToDevice)); CK(hipMemcpy(dg254, g254, 64, hipMemcpyHostToDevice));
  check_g1<G1Bls12381>("BLS12-381", d0, d1, dbad, dbases, 6, g381, dg);
  check_g1<G1Bn254>("BN254", d0, d1, dbad, dbases, 6, g254, dg);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d\n", prop.name, prop.multiProcessorCount);
  for (int wpc = 4; wpc <= 32; wpc *= 2) {
    if (quick && wpc != 8 && wpc != 32) continue;
    int blocks = 256 * wpc / 4, threads = 256, fit = 512;
    printf("--- %d waves/CU (%d blocks x %d)\n", wpc, blocks, threads);
    double t;
#define RATE(label, K, per, unit) t = time_kernel([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); }); \
    printf("  %-34s %8.2f %s\n", label, (double)blocks * threads * fit * per / t * 1e-9, unit)
    RATE("fp_mul<FqBandersnatch> (saturated)", (k_fmul_sat<FqBandersnatch>), 2, "Gmul/s");
    RATE("fu_mul<FqBandersnatch> 9x29", (k_fumul<FqBandersnatch>), 2, "Gmul/s");
    RATE("fu_sqr<FqBandersnatch> 9x29", (k_fusqr<FqBandersnatch>), 2, "Gsqr/s");
    RATE("fu_mul<FqBabyJubJub> 9x29", (k_fumul<FqBabyJubJub>), 2, "Gmul/s");
    RATE("fn_mul<FqBn254> (saturated)", (k_fmul_sat<FqBn254>), 2, "Gmul/s");
    RATE("fu_mul<FqBn254> 9x29", (k_fumul<FqBn254>), 2, "Gmul/s");
    RATE("fn_mul<FqBls12381> (saturated)", (k_fmul_sat<FqBls12381>), 2, "Gmul/s");
    RATE("fu_mul<FqBls12381> 14x28", (k_fumul<FqBls12381>), 2, "Gmul/s");
    RATE("fu_sqr<FqBls12381> 14x28", (k_fusqr<FqBls12381>), 2, "Gsqr/s");
#undef RATE
    if (wpc <= 8) {
      const int it = 48, nb = 8192;
      hipLaunchKernelGGL(k_g1_table<G1Bls12381>, dim3(nb / 64), dim3(64), 0, 0, dbases, nb, (const uint32_t *)dg381);
      t = time_kernel([&] { hipLaunchKernelGGL((k_g1madd<G1Bls12381, 0>), dim3(blocks), dim3(threads), 0, 0, out, (const uint32_t *)dbases, nb, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "g1 madd<Bls12381> (saturated)", (double)blocks * threads * it / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL((k_g1madd<G1Bls12381, 1>), dim3(blocks), dim3(threads), 0, 0, out, (const uint32_t *)dbases, nb, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "g1u_madd<Bls12381> 14x28", (double)blocks * threads * it / t * 1e-9);
      hipLaunchKernelGGL(k_g1_table<G1Bn254>, dim3(nb / 64), dim3(64), 0, 0, dbases, nb, (const uint32_t *)dg254);
      t = time_kernel([&] { hipLaunchKernelGGL((k_g1madd<G1Bn254, 0>), dim3(blocks), dim3(threads), 0, 0, out, (const uint32_t *)dbases, nb, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "g1 madd<Bn254> (saturated)", (double)blocks * threads * it / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL((k_g1madd<G1Bn254, 1>), dim3(blocks), dim3(threads), 0, 0, out, (const uint32_t *)dbases, nb, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "g1u_madd<Bn254> 9x29", (double)blocks * threads * it / t * 1e-9);
    }
    if (wpc <= 16) {
      const int it = 128;
      t = time_kernel([&] { hipLaunchKernelGGL((k_madd<SuiteBandersnatch, 0>), dim3(blocks), dim3(threads), 0, 0, out, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "te_madd<Bandersnatch> (saturated)", (double)blocks * threads * it / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL((k_madd<SuiteBandersnatch, 1>), dim3(blocks), dim3(threads), 0, 0, out, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "teu_madd<Bandersnatch> 9x29", (double)blocks * threads * it / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL((k_madd<SuiteBabyJubJub, 0>), dim3(blocks), dim3(threads), 0, 0, out, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "te_madd<BabyJubJub> (saturated)", (double)blocks * threads * it / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL((k_madd<SuiteBabyJubJub, 1>), dim3(blocks), dim3(threads), 0, 0, out, it, 777u, 0); });
      printf("  %-34s %8.3f Gadd/s\n", "teu_madd<BabyJubJub> 9x29", (double)blocks * threads * it / t * 1e-9);
    }
  }
  return 0;
}
