// tools/gather_probe.hip -- random 96-byte gathers (one G1 affine point of the 381-bit curve) from tables of 16 MB .. 160 GB: what a lane of
// the bucket accumulation sees when the fixed-base table holds EVERY multiple of every window row (no buckets, no sort: the table of a
// ring-1024 setup is ~164 GB of the 288 GB of HBM).  Prints latency (dependent chain per lane) and throughput (four independent gathers in
// flight per lane) at two waves per SIMD, the occupancy of k_accumulate<G1>.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/gather_probe tools/gather_probe.hip && /tmp/gather_probe [max_GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int DEP> __global__ void __launch_bounds__(256) k_gather(const uint4 *tab, uint64_t rows, int iters, uint32_t *out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = mix(t * 0x9e3779b97f4a7c15ULL + 1);
  uint4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; i++) {
    if (DEP) {
      const uint4 *p = tab + (s % rows) * 6;
      uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5];
      acc.x ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y; acc.y += a.y;
      s = mix(s + acc.x);                                  // the next index depends on the data
    } else {
      uint64_t s0 = mix(s + 1), s1 = mix(s + 2), s2 = mix(s + 3), s3 = mix(s + 4); s = s3;
      const uint4 *p0 = tab + (s0 % rows) * 6, *p1 = tab + (s1 % rows) * 6, *p2 = tab + (s2 % rows) * 6, *p3 = tab + (s3 % rows) * 6;
      uint4 v[24];
#pragma unroll
      for (int k = 0; k < 6; k++) { v[k] = p0[k]; v[6 + k] = p1[k]; v[12 + k] = p2[k]; v[18 + k] = p3[k]; }
#pragma unroll
      for (int k = 0; k < 24; k++) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    }
  }
  out[t] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
__global__ void k_fill(uint4 *tab, uint64_t n16) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { uint32_t v = (uint32_t)mix(i); tab[i] = uint4{v, v + 1, v + 2, v + 3}; }
}
int main(int argc, char **argv) {
  const double max_gb = argc > 1 ? atof(argv[1]) : 160.0;
  size_t free_b = 0, total_b = 0; CK(hipMemGetInfo(&free_b, &total_b));
  printf("HBM free %.1f GB of %.1f GB\n", free_b / 1e9, total_b / 1e9);
  const int lanes = 256 * 2 * 256;                          // two workgroups of 256 per CU
  uint32_t *out; CK(hipMalloc(&out, lanes * 4));
  const double sizes[] = {0.016, 1.0, 16.0, 64.0, max_gb};
  for (double gb : sizes) {
    if (gb > max_gb) continue;
    const uint64_t rows = (uint64_t)(gb * 1e9 / 96);
    uint4 *tab;
    if (hipMalloc(&tab, rows * 96) != hipSuccess) { printf("%.3f GB: allocation failed\n", gb); (void)hipGetLastError(); continue; }
    hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, 0, tab, rows * 6);
    CK(hipDeviceSynchronize());
    for (int dep = 1; dep >= 0; dep--) {
      const int iters = dep ? 64 : 32;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        if (dep) hipLaunchKernelGGL(k_gather<1>, dim3(lanes / 256), dim3(256), 0, 0, tab, rows, iters, out);
        else hipLaunchKernelGGL(k_gather<0>, dim3(lanes / 256), dim3(256), 0, 0, tab, rows, iters, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      const double g = (double)lanes * iters * (dep ? 1 : 4);
      if (dep) printf("%8.3f GB table: dependent gathers %.2f us each per lane (%.2f G gathers/s with one in flight per lane, %d lanes)\n", gb, best * 1e3 / iters, g / best * 1e-6, lanes);
      else printf("%8.3f GB table: four in flight per lane: %.2f G gathers/s = %.2f TB/s of 96-byte rows\n", gb, g / best * 1e-6, g * 96 / best * 1e-9);
    }
    CK(hipFree(tab));
  }
  return 0;
}
