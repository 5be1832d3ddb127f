// tools/issue_cost.hip -- what one vector instruction costs the SIMD that issues it on gfx950, per instruction class, measured with the
// shader-clock counter inside the kernel (s_memtime): 16 independent instances per loop iteration, hand-issued (asm volatile), at 1, 2, 4
// and 8 waves per SIMD.  Prints cycles per wave-instruction per SIMD (the reciprocal throughput the instruction budget of a kernel should
// be priced with: the guide gives 2 cycles for a plain 32-bit VALU instruction on the SIMD-32 and no integer-multiply figures).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/issue_cost tools/issue_cost.hip && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define OPS(X) \
  X(0, "v_mad_u64_u32 (vgpr x vgpr)", "v_mad_u64_u32 %0, vcc, %2, %3, %0", 1) \
  X(1, "v_mad_i64_i32", "v_mad_i64_i32 %0, vcc, %2, %3, %0", 1) \
  X(2, "v_mul_lo_u32", "v_mul_lo_u32 %1, %2, %1", 0) \
  X(3, "v_mul_hi_u32", "v_mul_hi_u32 %1, %2, %1", 0) \
  X(4, "v_mad_u32_u24", "v_mad_u32_u24 %1, %2, %3, %1", 0) \
  X(5, "v_add_u32", "v_add_u32 %1, %2, %1", 0) \
  X(6, "v_and_b32", "v_and_b32 %1, %2, %1", 0) \
  X(7, "v_lshrrev_b32", "v_lshrrev_b32 %1, 3, %1", 0) \
  X(8, "v_alignbit_b32", "v_alignbit_b32 %1, %2, %1, %3", 0) \
  X(9, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %1, %2, %1, vcc", 0) \
  X(10, "v_addc_co_u32 (carry chain through vcc)", "v_addc_co_u32 %1, vcc, %2, %1, vcc", 0) \
  X(11, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 0, %4", 1) \
  X(12, "v_ashrrev_i64", "v_ashrrev_i64 %0, 29, %0", 1) \
  X(13, "v_lshrrev_b64", "v_lshrrev_b64 %0, 3, %0", 1) \
  X(14, "v_bfe_u32", "v_bfe_u32 %1, %1, 3, 29", 0) \
  X(15, "v_and_or_b32", "v_and_or_b32 %1, %2, %3, %1", 0) \
  X(16, "v_add3_u32", "v_add3_u32 %1, %2, %3, %1", 0) \
  X(17, "v_or3_b32", "v_or3_b32 %1, %2, %3, %1", 0) \
  X(18, "v_lshl_or_b32", "v_lshl_or_b32 %1, %2, 3, %1", 0) \
  X(19, "v_mov_b32", "v_mov_b32 %1, %2", 0) \
  X(20, "v_cmp_lt_u64 (-> vcc)", "v_cmp_lt_u64 vcc, %0, %4", 1) \
  X(21, "v_ffbl_b32", "v_ffbl_b32 %1, %1", 0) \
  X(22, "v_bitop3_b32 (xor-and)", "v_bitop3_b32 %1, %2, %3, %1 bitop3:0x6a", 0) \
  X(23, "v_sub_co_u32 + v_subb_co_u32 pair (per instruction)", "v_sub_co_u32 %1, vcc, %2, %1\n\tv_subb_co_u32 %1, vcc, %3, %1, vcc", 2) \
  X(24, "v_lshlrev_b32", "v_lshlrev_b32 %1, 1, %1", 0) \
  X(25, "v_pk_add_u16", "v_pk_add_u16 %1, %2, %1", 0) \
  X(26, "v_mul_u32_u24", "v_mul_u32_u24 %1, %2, %1", 0) \
  X(27, "v_mad_u64_u32 (vgpr x sgpr)", "v_mad_u64_u32 %0, vcc, %2, %5, %0", 1) \
  X(28, "v_fma_f64", "v_fma_f64 %0, %0, %4, %4", 1) \
  X(29, "v_fma_f32", "v_fma_f32 %1, %1, %2, %3", 0) \
  X(30, "v_dot4_u32_u8", "v_dot4_u32_u8 %1, %2, %3, %1", 0)

template <int OP> __global__ void __launch_bounds__(256) k_issue(uint32_t *out, uint64_t *clk, int iters, uint32_t seed) {
  uint64_t a[16]; uint32_t b[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i] = (uint64_t)(seed + threadIdx.x) * (2 * i + 3); b[i] = seed * (2 * i + 5) + threadIdx.x; }
  uint32_t x = seed * 3 + blockIdx.x, y = (seed | 1) + (threadIdx.x & 7);
  uint64_t q = ((uint64_t)seed << 20) | 12345u;
  const uint32_t ys = seed | 1;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
#define X(ID, NAME, TXT, KIND) if constexpr (OP == ID) asm volatile(TXT : "+v"(a[i]), "+v"(b[i]) : "v"(x), "v"(y), "v"(q), "s"(ys) : "vcc");
      OPS(X)
#undef X
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32) ^ b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP> static double run(int waves_per_simd, int iters, int per_asm) {
  const int cus = 256, waves = cus * 4 * waves_per_simd, blocks = waves / 4;
  static uint32_t *d_out = nullptr; static uint64_t *d_clk = nullptr;
  if (!d_out) { CK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4)); CK(hipMalloc(&d_clk, 256 * 4 * 8 * 8)); }
  std::vector<uint64_t> h(waves);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, iters, 777u + rep);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_clk, waves * 8, hipMemcpyDeviceToHost));
    std::nth_element(h.begin(), h.begin() + waves / 2, h.end());
    // a SIMD issues for waves_per_simd waves during one wave's elapsed cycles
    const double cyc = (double)h[waves / 2] / ((double)iters * 16 * per_asm * waves_per_simd);
    if (rep) best = std::min(best, cyc);
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d; cycles per wave64 instruction per SIMD (shader-clock counter inside the kernel, median wave), 16 independent instances per iteration\n", prop.gcnArchName, prop.multiProcessorCount);
  printf("%-56s %8s %8s %8s %8s\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
#define X(ID, NAME, TXT, KIND) { const int per = KIND == 2 ? 2 : 1; printf("%-56s %8.2f %8.2f %8.2f %8.2f\n", NAME, run<ID>(1, 2048, per), run<ID>(2, 2048, per), run<ID>(4, 1024, per), run<ID>(8, 512, per)); }
  OPS(X)
#undef X
  return 0;
}
