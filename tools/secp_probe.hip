// debugging probe: secp256r1 field / group primitives on the device, one step per launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../ark_vrf_amd/csrc/proto_dev.h"
using namespace avrf;
using S = SuiteSecp256r1; using Fq = S::Fq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_field(uint32_t *out) {
  fp a = fp_const<Fq>(S::G_X), b = fp_const<Fq>(S::G_Y);
  fp m = fp_mul<Fq>(a, b), s = fp_add<Fq>(a, b), d = fp_sub<Fq>(a, b), q = fp_sqr<Fq>(a);
  fp i = fp_inv<Fq>(a), one = fp_mul<Fq>(i, a);
  for (int k = 0; k < 8; k++) { out[k] = m.v[k]; out[8 + k] = s.v[k]; out[16 + k] = d.v[k]; out[24 + k] = q.v[k]; out[32 + k] = one.v[k]; }
  out[40] = te_on_curve<S>(a, b);
}
template <class S> AVRF_DI te_ext smul_inline(te_pre p, fp k, int nbits) {
  te_ext tab[16];
  tab[0] = te_identity<S>(); tab[1] = te_from_pre<S>(p);
  for (int i = 2; i < 16; i++) tab[i] = te_madd<S>(tab[i - 1], p);
  te_ext acc = te_identity<S>();
  for (int w = (nbits + 3) / 4 - 1; w >= 0; w--) {
    acc = te_dbl<S>(te_dbl<S>(te_dbl<S>(te_dbl<S>(acc))));
    uint32_t d = (k.v[w >> 3] >> (4 * (w & 7))) & 15u;
    if (d) acc = te_add<S>(acc, tab[d]);
  }
  return acc;
}
__global__ void k_group(uint32_t *out, int step) {
  te_pre g; g.x = fp_const<Fq>(S::G_X); g.y = fp_const<Fq>(S::G_Y); g.k = fp_zero();
  te_ext p = te_from_pre<S>(g);
  if (step >= 1) p = te_dbl<S>(p);
  if (step >= 2) p = te_madd<S>(p, g);
  if (step >= 3) p = te_add<S>(p, p);
  if (step == 4) { p = te_add<S>(te_identity<S>(), p); }                       // identity + 6G
  if (step == 5) { p = te_add<S>(p, te_identity<S>()); }
  if (step == 6) { p = te_madd<S>(te_from_pre<S>(g), g); }                    // G + G through madd: the doubling case
  if (step == 7) { p = te_dbl<S>(te_identity<S>()); p = te_madd<S>(p, g); }   // 2 * identity, + G
  if (step == 8) { te_ext tab[4]; tab[0] = te_identity<S>(); tab[1] = te_from_pre<S>(g); for (int i = 2; i < 4; i++) tab[i] = te_madd<S>(tab[i - 1], g); p = tab[(out[63] & 1) + 2]; }
  if (step == 9) { fp k = fp_zero(); k.v[0] = 5; p = te_smul<S>(g, k, 4); }
  if (step == 10) { fp k = fp_zero(); k.v[0] = 0x35; p = te_smul<S>(g, k, 8); }
  if (step == 12) { fp k = fp_zero(); k.v[0] = 5; p = smul_inline<S>(g, k, 4); }
  if (step == 13) { fp k = fp_zero(); k.v[0] = 12345; p = smul_inline<S>(g, k, 256); }
  if (step == 11) { fp k = fp_zero(); k.v[0] = 12345; p = te_smul<S>(g, k, 256); }
  te_aff a = te_to_aff<S>(p);
  fp x = fp_from_mont<Fq>(a.x), y = fp_from_mont<Fq>(a.y);
  for (int k = 0; k < 8; k++) { out[k] = x.v[k]; out[8 + k] = y.v[k]; }
}
int main() {
  uint32_t *d, h[64]; CK(hipMalloc(&d, 256));
  fprintf(stderr, "field...\n");
  hipLaunchKernelGGL(k_field, dim3(1), dim3(64), 0, 0, d); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
  const char *nm[5] = {"mul", "add", "sub", "sqr", "a*inv(a)"};
  for (int r = 0; r < 5; r++) { printf("%s ", nm[r]); for (int k = 7; k >= 0; k--) printf("%08x", h[8 * r + k]); printf("\n"); }
  printf("G on curve: %u\n", h[40]); fflush(stdout);
  CK(hipMemset(d, 0, 256));
  int first = getenv("STEP") ? atoi(getenv("STEP")) : 4;
  for (int step = first; step <= first; step++) {
    fprintf(stderr, "group step %d...\n", step);
    hipLaunchKernelGGL(k_group, dim3(1), dim3(64), 0, 0, d, step); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    printf("step %d x ", step); for (int k = 7; k >= 0; k--) printf("%08x", h[k]); printf(" y "); for (int k = 7; k >= 0; k--) printf("%08x", h[8 + k]); printf("\n"); fflush(stdout);
  }
  return 0;
}
