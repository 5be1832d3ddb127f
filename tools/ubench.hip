// tools/ubench.hip -- integer-ALU micro-benchmarks for gfx950 (the local guides list no
// integer-multiply rates; SURVEY.md §7 "Hard parts").  Measures what bounds the field
// arithmetic: v_mad_u64_u32 / v_mul_lo / v_mul_hi / v_add_co issue rates, f64 FMA rate,
// Montgomery-multiplication and mixed-addition throughput per chip at several occupancies.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../ark_vrf_amd/csrc/te.h"
#include "../ark_vrf_amd/csrc/curves.h"

using namespace avrf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP> __global__ void k_op(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5, dm = 1.0000001, da = 0.5;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // mad_u64_u32, 8 independent chains
      c0 = (uint64_t)(uint32_t)c0 * b + c0; c1 = (uint64_t)(uint32_t)c1 * b + c1; c2 = (uint64_t)(uint32_t)c2 * b + c2; c3 = (uint64_t)(uint32_t)c3 * b + c3;
      c4 = (uint64_t)(uint32_t)c4 * b + c4; c5 = (uint64_t)(uint32_t)c5 * b + c5; c6 = (uint64_t)(uint32_t)c6 * b + c6; c7 = (uint64_t)(uint32_t)c7 * b + c7;
    } else if (OP == 1) {  // mul_lo
      uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = c4, x5 = c5, x6 = c6, x7 = c7;
      x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b;
      c0 = x0; c1 = x1; c2 = x2; c3 = x3; c4 = x4; c5 = x5; c6 = x6; c7 = x7;
    } else if (OP == 2) {  // mul_hi
      c0 = __umulhi((uint32_t)c0, b) + 1; c1 = __umulhi((uint32_t)c1, b) + 1; c2 = __umulhi((uint32_t)c2, b) + 1; c3 = __umulhi((uint32_t)c3, b) + 1;
      c4 = __umulhi((uint32_t)c4, b) + 1; c5 = __umulhi((uint32_t)c5, b) + 1; c6 = __umulhi((uint32_t)c6, b) + 1; c7 = __umulhi((uint32_t)c7, b) + 1;
    } else if (OP == 3) {  // 64-bit add (add_co + addc)
      c0 += c1; c1 += c2; c2 += c3; c3 += c4; c4 += c5; c5 += c6; c6 += c7; c7 += c0;
    } else if (OP == 4) {  // f64 fma
      d0 = fma(d0, dm, da); d1 = fma(d1, dm, da); d2 = fma(d2, dm, da); d3 = fma(d3, dm, da);
      d4 = fma(d4, dm, da); d5 = fma(d5, dm, da); d6 = fma(d6, dm, da); d7 = fma(d7, dm, da);
    } else if (OP == 5) {  // mad_u32_u24
      uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = c4, x5 = c5, x6 = c6, x7 = c7;
      x0 = __umul24(x0, b) + x1; x1 = __umul24(x1, b) + x2; x2 = __umul24(x2, b) + x3; x3 = __umul24(x3, b) + x4;
      x4 = __umul24(x4, b) + x5; x5 = __umul24(x5, b) + x6; x6 = __umul24(x6, b) + x7; x7 = __umul24(x7, b) + x0;
      c0 = x0; c1 = x1; c2 = x2; c3 = x3; c4 = x4; c5 = x5; c6 = x6; c7 = x7;
    }
  }
  uint64_t r = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  double dr = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ (uint32_t)dr;
}

// The multiply-add the Montgomery multipliers are made of, issued by hand so that the compiler can neither fold nor reorder it
// (k_op<0> above is NOT such a stream: its eight chains compile to three v_mad_u64_u32 per iteration -- see the disassembly count
// in profiles/r4_ubench.txt -- which is why its "rate" contradicted the mixed-addition kernels in round 3).  Sixteen independent
// 64-bit accumulators, distinct multiplicand registers, one multiplier operand in an SGPR (SG = 1, as the modulus limbs are) or
// in a VGPR; CARRY = 1 follows every multiply-add with the v_addc_co_u32 of the product-scanning form.
template <int SG, int CARRY> __global__ void k_mad_asm(uint32_t *out, int iters, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  uint64_t a8 = a0 * 17, a9 = a0 * 19, a10 = a0 * 21, a11 = a0 * 23, a12 = a0 * 25, a13 = a0 * 27, a14 = a0 * 29, a15 = a0 * 31;
  uint32_t x0 = seed * 3 + blockIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, c = 0;
  const uint32_t ys = seed | 1; uint32_t yv = ys + (threadIdx.x & 1);
#define MAD_(acc, x) do { if (SG) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(ys) : "vcc"); \
                          else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(yv) : "vcc"); \
                          if (CARRY) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(c) : : "vcc"); } while (0)
  for (int i = 0; i < iters; i++) {
    MAD_(a0, x0); MAD_(a1, x1); MAD_(a2, x2); MAD_(a3, x3); MAD_(a4, x0); MAD_(a5, x1); MAD_(a6, x2); MAD_(a7, x3);
    MAD_(a8, x0); MAD_(a9, x1); MAD_(a10, x2); MAD_(a11, x3); MAD_(a12, x0); MAD_(a13, x1); MAD_(a14, x2); MAD_(a15, x3);
  }
#undef MAD_
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ a8 ^ a9 ^ a10 ^ a11 ^ a12 ^ a13 ^ a14 ^ a15;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ c;
}

// Where does the multiply-add + carry PAIR lose time?  MODE 0: every pair's carry goes into its OWN third word (no chain through
// one register); MODE 1: carries leave through SGPR pairs and are consumed four multiply-adds later (software-pipelined carries);
// MODE 2: the real column shape -- a dependent chain mad -> mad on ONE accumulator with its carries into one third word, four
// such columns interleaved instruction by instruction; MODE 3: the same four columns one after the other (what the generated
// multiplier does today).
template <int MODE> __global__ void k_mad_pairs(uint32_t *out, int iters, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  uint32_t x0 = seed * 3 + blockIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const uint32_t ys = seed | 1;
#define PV_(acc, c, x) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c) : "v"(x), "s"(ys) : "vcc")
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      PV_(a0, c0, x0); PV_(a1, c1, x1); PV_(a2, c2, x2); PV_(a3, c3, x3); PV_(a4, c4, x0); PV_(a5, c5, x1); PV_(a6, c6, x2); PV_(a7, c7, x3);
      PV_(a0, c0, x1); PV_(a1, c1, x2); PV_(a2, c2, x3); PV_(a3, c3, x0); PV_(a4, c4, x1); PV_(a5, c5, x2); PV_(a6, c6, x3); PV_(a7, c7, x0);
    } else if (MODE == 1) {
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %12, %0\n\tv_mad_u64_u32 %1, s[22:23], %9, %12, %1\n\t"
                   "v_mad_u64_u32 %2, s[24:25], %10, %12, %2\n\tv_mad_u64_u32 %3, s[26:27], %11, %12, %3\n\t"
                   "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\t"
                   "v_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(ys)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %12, %0\n\tv_mad_u64_u32 %1, s[22:23], %9, %12, %1\n\t"
                   "v_mad_u64_u32 %2, s[24:25], %10, %12, %2\n\tv_mad_u64_u32 %3, s[26:27], %11, %12, %3\n\t"
                   "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\t"
                   "v_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                   : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x1), "v"(x2), "v"(x3), "v"(x0), "s"(ys)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %12, %0\n\tv_mad_u64_u32 %1, s[22:23], %9, %12, %1\n\t"
                   "v_mad_u64_u32 %2, s[24:25], %10, %12, %2\n\tv_mad_u64_u32 %3, s[26:27], %11, %12, %3\n\t"
                   "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\t"
                   "v_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x2), "v"(x3), "v"(x0), "v"(x1), "s"(ys)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %12, %0\n\tv_mad_u64_u32 %1, s[22:23], %9, %12, %1\n\t"
                   "v_mad_u64_u32 %2, s[24:25], %10, %12, %2\n\tv_mad_u64_u32 %3, s[26:27], %11, %12, %3\n\t"
                   "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\t"
                   "v_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                   : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x3), "v"(x0), "v"(x1), "v"(x2), "s"(ys)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (MODE == 2) {      // four dependent columns, interleaved
      PV_(a0, c0, x0); PV_(a1, c1, x1); PV_(a2, c2, x2); PV_(a3, c3, x3); PV_(a0, c0, x1); PV_(a1, c1, x2); PV_(a2, c2, x3); PV_(a3, c3, x0);
      PV_(a0, c0, x2); PV_(a1, c1, x3); PV_(a2, c2, x0); PV_(a3, c3, x1); PV_(a0, c0, x3); PV_(a1, c1, x0); PV_(a2, c2, x1); PV_(a3, c3, x2);
    } else {                      // the same four columns one after the other
      PV_(a0, c0, x0); PV_(a0, c0, x1); PV_(a0, c0, x2); PV_(a0, c0, x3); PV_(a1, c1, x1); PV_(a1, c1, x2); PV_(a1, c1, x3); PV_(a1, c1, x0);
      PV_(a2, c2, x2); PV_(a2, c2, x3); PV_(a2, c2, x0); PV_(a2, c2, x1); PV_(a3, c3, x3); PV_(a3, c3, x0); PV_(a3, c3, x1); PV_(a3, c3, x2);
    }
  }
#undef PV_
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}

template <class F> __global__ void k_fmul(uint32_t *out, int iters, uint32_t seed) {
  fp a, b;
  for (int i = 0; i < 8; i++) { a.v[i] = seed * (i + 1) + threadIdx.x; b.v[i] = seed * (i + 7) + blockIdx.x; }
  a.v[7] &= 0x0fffffff; b.v[7] &= 0x0fffffff;
  for (int i = 0; i < iters; i++) { a = fp_mul<F>(a, b); b = fp_mul<F>(b, a); }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> __global__ void k_fadd(uint32_t *out, int iters, uint32_t seed) {
  fp a, b;
  for (int i = 0; i < 8; i++) { a.v[i] = seed * (i + 1) + threadIdx.x; b.v[i] = seed * (i + 7) + blockIdx.x; }
  a.v[7] &= 0x0fffffff; b.v[7] &= 0x0fffffff;
  for (int i = 0; i < iters; i++) { a = fp_add<F>(a, b); b = fp_sub<F>(b, a); }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class S> __global__ void k_madd(uint32_t *out, int iters, uint32_t seed) {
  using Fq = typename S::Fq;
  te_ext p = te_identity<S>();
  te_pre q; q.x = fp_const<Fq>(S::G_X); q.y = fp_const<Fq>(S::G_Y); q.k = fp_const<Fq>(S::G_K);
  q.x.v[0] ^= (threadIdx.x & 1);  // keep the compiler from hoisting everything
  for (int i = 0; i < iters; i++) p = te_madd<S>(p, q);
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= p.x.v[i] ^ p.y.v[i] ^ p.z.v[i] ^ p.t.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}

// N-limb field (KZG G1 base fields): VARIANT 0 = fn_mul_cios (operand scanning), 1 = fn_mul (product scanning)
template <class F, int VARIANT> __global__ void k_fnmul(uint32_t *out, int iters, uint32_t seed) {
  constexpr int N = F::N;
  fpn<N> a, b;
  for (int i = 0; i < N; i++) { a.v[i] = seed * (i + 1) + threadIdx.x; b.v[i] = seed * (i + 7) + blockIdx.x; }
  a.v[N - 1] &= 0x00ffffff; b.v[N - 1] &= 0x00ffffff;
  for (int i = 0; i < iters; i++) {
    if (VARIANT == 0) { a = fn_mul_cios<F>(a, b); b = fn_mul_cios<F>(b, a); }
    else { a = fn_mul<F>(a, b); b = fn_mul<F>(b, a); }
  }
  uint32_t r = 0;
  for (int i = 0; i < N; i++) r ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class C> __global__ void __launch_bounds__(256, 2) k_g1madd(uint32_t *out, int iters, uint32_t seed) {
  using CV = G1Curve<C>; constexpr int N = C::Fq::N;
  typename CV::acc_t p = CV::identity();
  typename CV::base_t q; q.x = fn_one<typename C::Fq>(); q.y = fn_one<typename C::Fq>();
  q.x.v[0] ^= (threadIdx.x & 1); q.y.v[1] ^= seed;
  for (int i = 0; i < iters; i++) { p = CV::madd(p, q, (i & 1) != 0); q.x.v[0] += 2; }
  uint32_t r = 0;
  for (int i = 0; i < N; i++) r ^= p.x.v[i] ^ p.y.v[i] ^ p.zz.v[i] ^ p.zzz.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}

template <class K> double time_kernel(K launch, int reps = 3) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm-up
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best * 1e-3;
}

struct FqBandersnatch8 : FqBandersnatch { static constexpr int N = 8; };

int main() {
  {  // fn_mul_ps agrees with fn_mul
    uint32_t *o0, *o1; CK(hipMalloc(&o0, 64 * 256 * 4)); CK(hipMalloc(&o1, 64 * 256 * 4));
    hipLaunchKernelGGL((k_fnmul<FqBls12381, 0>), dim3(64), dim3(256), 0, 0, o0, 16, 99u);
    hipLaunchKernelGGL((k_fnmul<FqBls12381, 1>), dim3(64), dim3(256), 0, 0, o1, 16, 99u);
    static uint32_t h0[64 * 256], h1[64 * 256];
    CK(hipMemcpy(h0, o0, sizeof h0, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, o1, sizeof h1, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 64 * 256; i++) bad += h0[i] != h1[i];
    printf("fn_mul vs fn_mul_cios (12 limbs): %d mismatches\n", bad);
    hipLaunchKernelGGL((k_fnmul<FqBandersnatch8, 0>), dim3(64), dim3(256), 0, 0, o0, 16, 99u);
    hipLaunchKernelGGL((k_fnmul<FqBandersnatch8, 1>), dim3(64), dim3(256), 0, 0, o1, 16, 99u);
    CK(hipMemcpy(h0, o0, sizeof h0, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, o1, sizeof h1, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < 64 * 256; i++) bad += h0[i] != h1[i];
    printf("fn_mul vs fn_mul_cios (8 limbs): %d mismatches\n", bad);
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  uint32_t *out; CK(hipMalloc(&out, 256 * 32 * 256 * 4 * 4));
  const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "add64 (add_co+addc)", "v_fma_f64", "mul24+add"};
  int iters = 4096;
  for (int wpc = 4; wpc <= 32; wpc *= 2) {   // waves per CU
    int blocks = 256 * wpc / 4, threads = 256;
    printf("--- %d waves/CU (%d blocks x %d)\n", wpc, blocks, threads);
    for (int op = 0; op < 6; op++) {
      double t;
      switch (op) {
        case 0: t = time_kernel([&] { hipLaunchKernelGGL(k_op<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
        case 1: t = time_kernel([&] { hipLaunchKernelGGL(k_op<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
        case 2: t = time_kernel([&] { hipLaunchKernelGGL(k_op<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
        case 3: t = time_kernel([&] { hipLaunchKernelGGL(k_op<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
        case 4: t = time_kernel([&] { hipLaunchKernelGGL(k_op<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
        default: t = time_kernel([&] { hipLaunchKernelGGL(k_op<5>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); break;
      }
      double ops = (double)blocks * threads * iters * 8;
      printf("  %-22s %8.2f Gop/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", names[op], ops / t * 1e-9, ops / t / 2.4e9 / 256);
    }
    {
      const double mops = (double)blocks * threads * iters * 16;
      double tm = time_kernel([&] { hipLaunchKernelGGL((k_mad_asm<1, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (asm stream, 16 independent accumulators, SGPR multiplier)\n", "mad_u64_u32 asm/s", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL((k_mad_asm<0, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (asm stream, VGPR multiplier)\n", "mad_u64_u32 asm/v", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL((k_mad_asm<1, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (each followed by v_addc_co_u32: the product-scanning pair)\n", "mad+addc asm pairs", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL(k_mad_pairs<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (pairs, 8 independent accumulators AND third words, carries through vcc)\n", "pairs independent", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL(k_mad_pairs<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (pairs, carries through SGPR pairs, consumed 4 multiply-adds later)\n", "pairs sgpr-carry d4", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL(k_mad_pairs<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (4 dependent columns interleaved)\n", "pairs 4 cols interl.", mops / tm * 1e-9);
      tm = time_kernel([&] { hipLaunchKernelGGL(k_mad_pairs<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); });
      printf("  %-22s %8.2f Gop/s  (4 dependent columns one after the other: today's multiplier)\n", "pairs 4 cols serial", mops / tm * 1e-9);
    }
    int fit = 512;
    double t = time_kernel([&] { hipLaunchKernelGGL(k_fmul<FqBandersnatch>, dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fp_mul<FqBandersnatch>", (double)blocks * threads * fit * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL(k_fmul<FqBabyJubJub>, dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fp_mul<FqBabyJubJub>", (double)blocks * threads * fit * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL((k_fnmul<FqBls12381, 0>), dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fn_mul_cios<Bls12381>", (double)blocks * threads * fit * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL((k_fnmul<FqBls12381, 1>), dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fn_mul<FqBls12381>", (double)blocks * threads * fit * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL((k_fnmul<FqBandersnatch8, 0>), dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fn_mul_cios<8 limbs>", (double)blocks * threads * fit * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL((k_fnmul<FqBandersnatch8, 1>), dim3(blocks), dim3(threads), 0, 0, out, fit, 777u); });
    printf("  %-22s %8.2f Gmul/s\n", "fn_mul<8 limbs>", (double)blocks * threads * fit * 2 / t * 1e-9);
    if (wpc <= 8) {
      t = time_kernel([&] { hipLaunchKernelGGL(k_g1madd<G1Bls12381>, dim3(blocks), dim3(threads), 0, 0, out, 64, 777u); });
      printf("  %-22s %8.3f Gadd/s\n", "g1_madd<Bls12381>", (double)blocks * threads * 64 / t * 1e-9);
      t = time_kernel([&] { hipLaunchKernelGGL(k_g1madd<G1Bn254>, dim3(blocks), dim3(threads), 0, 0, out, 64, 777u); });
      printf("  %-22s %8.3f Gadd/s\n", "g1_madd<Bn254>", (double)blocks * threads * 64 / t * 1e-9);
    }
    t = time_kernel([&] { hipLaunchKernelGGL(k_fadd<FqBandersnatch>, dim3(blocks), dim3(threads), 0, 0, out, fit * 4, 777u); });
    printf("  %-22s %8.2f Gop/s\n", "fp_add/sub", (double)blocks * threads * fit * 4 * 2 / t * 1e-9);
    t = time_kernel([&] { hipLaunchKernelGGL(k_madd<SuiteBandersnatch>, dim3(blocks), dim3(threads), 0, 0, out, 128, 777u); });
    printf("  %-22s %8.3f Gadd/s  (%.1f us per add per lane)\n", "te_madd<Bandersnatch>", (double)blocks * threads * 128 / t * 1e-9, t / 128 * 1e6);
  }
  return 0;
}
