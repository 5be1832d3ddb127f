// tools/lint_selftest.hip -- a code shape tools/lint_device_code.py must refuse (tests/test_device_lint.py compiles it and
// expects a violation): an out-of-line device function holding 1 536 multiply-adds whose carries leave through an SGPR pair,
// the signature of generic (per-statement) multiplications inlined en masse.  Never linked into the library, never run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define M1 asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "s20", "s21");
#define M8 M1 M1 M1 M1 M1 M1 M1 M1
#define M64 M8 M8 M8 M8 M8 M8 M8 M8
#define M512 M64 M64 M64 M64 M64 M64 M64 M64
__device__ __noinline__ static uint64_t sgpr_carry_multiply_adds_en_masse(uint64_t a, uint32_t x, uint32_t y) {
  M512 M512 M512
  return a;
}
__device__ __noinline__ static uint64_t a_few(uint64_t a, uint32_t x, uint32_t y) {
  M64
  return a;
}
__global__ void k_selftest(uint64_t *out) { out[threadIdx.x] = sgpr_carry_multiply_adds_en_masse(out[threadIdx.x], threadIdx.x, 7) ^ a_few(1, threadIdx.x, 9); }
int main() { return 0; }
