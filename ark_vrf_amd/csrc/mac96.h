// mac96.h -- the multiply-accumulate step of the product-scanning Montgomery multiplications (fp256.h, fpn.h).
// A column of the schoolbook product is summed in ONE 96-bit accumulator (lo64 | ex): each limb product is a single
// v_mad_u64_u32 accumulating in place, and its carry-out (VOP3B sdst) is folded into `ex` by one v_addc_co_u32 --
// 2 VALU ops per limb product.  (The operand-scanning CIOS form costs a mad plus a 64-bit add and zero-extension
// moves per product; measured on MI355X, tools/ubench.hip: 8 limbs 88 -> 129 Gmul/s, 12 limbs 42 -> 59 Gmul/s.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mont8_asm_gen.h"

namespace avrf {

__device__ __forceinline__ void mac96(uint64_t &lo, uint32_t &ex, uint32_t a, uint32_t b) {
  uint64_t cy;
  asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32 %2, %1, 0, %2, %1" : "+v"(lo), "=&s"(cy), "+v"(ex) : "v"(a), "v"(b));
}
// second factor is a compile-time constant (a modulus limb): read from an SGPR
__device__ __forceinline__ void mac96_k(uint64_t &lo, uint32_t &ex, uint32_t a, uint32_t k) {
  uint64_t cy;
  asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32 %2, %1, 0, %2, %1" : "+v"(lo), "=&s"(cy), "+v"(ex) : "v"(a), "s"(k));
}

// a * b / 2^(32 N) mod p for N-limb operands < p; result in t[0..N) plus the returned bit 32 N, < 2p before the caller's
// conditional subtraction.  The returned bit is zero unless the top bit of p is set (F::FULL: secp256r1's two fields, which
// have no asm block).  P: modulus limbs, NINV = -p^-1 mod 2^32.
template <int N, class F>
__device__ __forceinline__ uint32_t mont_mul_ps(uint32_t (&t)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
#ifndef AVRF_NO_MONT_ASM
  if constexpr (N == 8 && MontAsm8<F>::value) {      // the same algorithm as one asm block with a sliding accumulator (tools/gen_mont_asm.py)
    MontAsm8<F>::mul(t, a, b);
    return 0;
  }
#ifndef AVRF_NO_MONT_ASM12
  if constexpr (N == 12 && MontAsm12<F>::value) { MontAsm12<F>::mul(t, a, b); return 0; }
#endif
#endif
  uint32_t m[N];
  uint64_t lo = 0; uint32_t ex = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac96(lo, ex, a[i], b[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac96_k(lo, ex, m[i], F::P[k - i]);
    m[k] = (uint32_t)lo * F::NINV;
    mac96_k(lo, ex, m[k], F::P[0]);
    lo = (lo >> 32) | ((uint64_t)ex << 32); ex = 0;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) mac96(lo, ex, a[i], b[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; i++) mac96_k(lo, ex, m[i], F::P[k - i]);
    t[k - N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)ex << 32); ex = 0;
  }
  t[N - 1] = (uint32_t)lo;
  return (uint32_t)(lo >> 32);
}

// a * a / 2^(32 N) mod p: the asm blocks with N (N + 1) / 2 limb products where there is one (tools/gen_mont_asm.py)
template <int N, class F>
__device__ __forceinline__ uint32_t mont_sqr_ps(uint32_t (&t)[N], const uint32_t (&a)[N]) {
#if !defined(AVRF_NO_MONT_ASM) && !defined(AVRF_NO_MONT_SQR)
  if constexpr (N == 8 && MontAsm8<F>::value) { MontAsm8<F>::sqr(t, a); return 0; }
#ifndef AVRF_NO_MONT_ASM12
  if constexpr (N == 12 && MontAsm12<F>::value) { MontAsm12<F>::sqr(t, a); return 0; }
#endif
#endif
  return mont_mul_ps<N, F>(t, a, a);
}

}  // namespace avrf
