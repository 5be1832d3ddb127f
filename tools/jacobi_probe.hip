// tools/jacobi_probe.hip -- the two-symbol macro-step Jacobi of fp256.h (fp_jacobi2_nf) against Euler's criterion a^((p-1)/2) and against the
// single-bit binary form it replaced, on random, structured and adversarial inputs, with timings.  Stand-alone:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -enable-ipra=0 -Iark_vrf_amd/csrc -o /tmp/jacobi_probe tools/jacobi_probe.hip && /tmp/jacobi_probe
// exit status 0 iff every symbol agrees.  (tests/test_gpu_jacobi.py builds and runs it on the GPU box.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include "fp256.h"
using namespace avrf;

// the form fp_jacobi2_nf replaced: one fused single-bit step per iteration on all eight limbs (~60 instructions x ~1.4 x 255 iterations)
template <class F> AVRF_DN int jacobi_single_bit(fp a) {
  fp n = fp_const<F>(F::P);
  uint32_t t = 0;
#pragma unroll 1
  while (!fp_is_zero(a)) {
    const bool odd = (a.v[0] & 1u) != 0;
    fp d1, d2;
    const bool lt = sub8(d1, a, n) != 0;
    sub8(d2, n, a);
    const bool sw = odd && lt;
    t ^= sw ? ((a.v[0] & n.v[0]) >> 1) & 1u : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t an = odd ? (lt ? d2.v[i] : d1.v[i]) : a.v[i];
      n.v[i] = sw ? a.v[i] : n.v[i];
      a.v[i] = an;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) a.v[i] = (a.v[i] >> 1) | (a.v[i + 1] << 31);
    a.v[7] >>= 1;
    t ^= ((n.v[0] >> 1) ^ (n.v[0] >> 2)) & 1u;
  }
  uint32_t o = n.v[0] ^ 1u;
#pragma unroll
  for (int i = 1; i < 8; i++) o |= n.v[i];
  return o ? 0 : ((t & 1u) ? -1 : 1);
}
template <class F> AVRF_DN int jacobi_euler(fp a) {               // a in Montgomery form or not: chi(a R) = chi(a); a^((p-1)/2) of the value as given
  if (fp_is_zero(a)) return 0;
  fp e = fp_const<F>(F::P);                                        // (p - 1) / 2
  e.v[0] &= ~1u;
  for (int i = 0; i < 7; i++) e.v[i] = (e.v[i] >> 1) | (e.v[i + 1] << 31);
  e.v[7] >>= 1;
  fp r = fp_one<F>();
  for (int i = 255; i >= 0; i--) { r = fp_mul_nf<F>(r, r); if ((e.v[i >> 5] >> (i & 31)) & 1) r = fp_mul_nf<F>(r, a); }
  return fp_eq(r, fp_one<F>()) ? 1 : -1;
}
template <class F, int WHICH> __global__ void __launch_bounds__(128) k_jac(const uint32_t *in, uint32_t n, int *out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fp a, b;
  for (int i = 0; i < 8; i++) { a.v[i] = in[16 * (size_t)j + i]; b.v[i] = in[16 * (size_t)j + 8 + i]; }
  int ja, jb;
  if constexpr (WHICH == 0) fp_jacobi2_nf<F>(a, b, &ja, &jb);
  else if constexpr (WHICH == 1) { ja = jacobi_single_bit<F>(a); jb = jacobi_single_bit<F>(b); }
  else { ja = jacobi_euler<F>(a); jb = jacobi_euler<F>(b); }
  out[2 * (size_t)j] = ja; out[2 * (size_t)j + 1] = jb;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint64_t rs = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }

template <class F> int run(const char *name, uint32_t n_rand) {
  std::vector<uint32_t> in;
  auto push = [&](const uint32_t *v) { for (int i = 0; i < 8; i++) in.push_back(v[i]); };
  auto lt_p = [&](const uint32_t *v) { for (int i = 7; i >= 0; i--) { if (v[i] != F::P[i]) return v[i] < F::P[i]; } return false; };
  std::vector<std::vector<uint32_t>> vals;
  // structured: 0, small values, p - small, powers of two and their neighbours, zero low words (the z = 30 cap), long runs of ones, values whose
  // top limbs equal p's (compares decided in the low limbs)
  for (uint32_t s = 0; s < 40; s++) { std::vector<uint32_t> v(8, 0); v[0] = s; vals.push_back(v); }
  for (uint32_t s = 1; s < 40; s++) { std::vector<uint32_t> v(F::P, F::P + 8); uint64_t b = s; for (int i = 0; i < 8 && b; i++) { uint64_t d = (uint64_t)v[i] - b; v[i] = (uint32_t)d; b = (d >> 63) & 1; } vals.push_back(v); }
  for (int bit = 0; bit < 256; bit++) for (int d = -1; d <= 1; d++) {
    std::vector<uint32_t> v(8, 0); v[bit >> 5] = 1u << (bit & 31);
    if (d == 1) v[0] |= 1u; if (d == -1) { for (int i = 0; i < (bit >> 5); i++) v[i] = 0xffffffffu; v[bit >> 5] = (1u << (bit & 31)) - 1u; }
    if (lt_p(v.data())) vals.push_back(v);
  }
  for (int k = 1; k < 8; k++) for (int rep = 0; rep < 8; rep++) {   // low k words zero, the rest random
    std::vector<uint32_t> v(8); for (int i = 0; i < 8; i++) v[i] = i < k ? 0u : rnd(); v[7] &= 0x0fffffffu; vals.push_back(v); }
  for (int k = 1; k < 8; k++) for (int rep = 0; rep < 8; rep++) {   // top 8 - k words equal p's
    std::vector<uint32_t> v(F::P, F::P + 8); for (int i = 0; i < k; i++) v[i] = rnd(); if (lt_p(v.data())) vals.push_back(v); }
  for (int rep = 0; rep < 64; rep++) { std::vector<uint32_t> v(8, 0); int top = rnd() % 8; for (int i = 0; i <= top; i++) v[i] = rnd(); if (top == 7) v[7] &= 0x0fffffffu; vals.push_back(v); }   // short values
  while (vals.size() & 1) vals.push_back(std::vector<uint32_t>(8, 0));
  for (auto &v : vals) push(v.data());
  const uint32_t n_struct = (uint32_t)vals.size() / 2;
  for (uint32_t j = 0; j < n_rand; j++) { uint32_t v[16]; for (int i = 0; i < 16; i++) v[i] = rnd(); v[7] &= 0x0fffffffu; v[15] &= 0x0fffffffu; for (int i = 0; i < 16; i++) in.push_back(v[i]); }
  const uint32_t n = n_struct + n_rand;
  uint32_t *d_in; int *d_out[3];
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  for (int w = 0; w < 3; w++) CK(hipMalloc(&d_out[w], 2 * (size_t)n * sizeof(int)));
  std::vector<int> out[3];
  double ms[3];
  for (int w = 0; w < 3; w++) {
    out[w].resize(2 * (size_t)n);
    for (int rep = 0; rep < 2; rep++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      if (w == 0) hipLaunchKernelGGL((k_jac<F, 0>), dim3((n + 127) / 128), dim3(128), 0, 0, d_in, n, d_out[w]);
      if (w == 1) hipLaunchKernelGGL((k_jac<F, 1>), dim3((n + 127) / 128), dim3(128), 0, 0, d_in, n, d_out[w]);
      if (w == 2) hipLaunchKernelGGL((k_jac<F, 2>), dim3((n + 127) / 128), dim3(128), 0, 0, d_in, n, d_out[w]);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[w] = t;
    }
    CK(hipMemcpy(out[w].data(), d_out[w], 2 * (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  }
  size_t bad = 0, plus = 0, minus = 0, zero = 0;
  for (size_t i = 0; i < 2 * (size_t)n; i++) {
    if (out[0][i] != out[2][i] || out[1][i] != out[2][i]) { if (bad < 5) fprintf(stderr, "%s: symbol %zu: macro %d single-bit %d euler %d\n", name, i, out[0][i], out[1][i], out[2][i]); bad++; }
    plus += out[2][i] == 1; minus += out[2][i] == -1; zero += out[2][i] == 0;
  }
  printf("%-28s %u pairs (%u structured): mismatches %zu  (+1: %zu, -1: %zu, 0: %zu)   macro-step pair %.3f ms, single-bit x2 %.3f ms, Euler x2 %.3f ms  => %.2f M symbols/s (single-bit %.2f)\n",
         name, n, n_struct, bad, plus, minus, zero, ms[0], ms[1], ms[2], 2e-3 * n / ms[0], 2e-3 * n / ms[1]);
  hipFree(d_in); for (int w = 0; w < 3; w++) hipFree(d_out[w]);
  return bad != 0;
}
int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u;
  int bad = 0;
  bad |= run<FqBandersnatch>("Bandersnatch Fq (BLS12-381 Fr)", n);
  bad |= run<FqEd25519>("Ed25519 Fq (2^255 - 19)", n / 8);
  printf(bad ? "FAIL\n" : "jacobi probe ok\n");
  return bad;
}
