// Dev probe: HBM read bandwidth of a plain linear stream on RANDOM data (the DVFS caveat: constant data reads faster).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void fill(uint32_t *p, size_t n, uint32_t seed, int random) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  for (; i < n; i += s) { uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = random ? x : 0x3c003c00u; }
}
template <bool NT> __global__ __launch_bounds__(256) void rd(const u32x4 *w, size_t nvec, uint32_t *out) {
  u32x4 a = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i + 7 * stride < nvec; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(w + i + j * stride) : w[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) a ^= v[j];
  }
  if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345678u) out[0] = 1;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  uint32_t *w[2], *out; hipMalloc(&w[0], bytes); hipMalloc(&w[1], bytes); hipMalloc(&out, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int random = 0; random < 2; ++random) {
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, w[k], bytes / 4, 17u + k, random);
    for (int nt = 0; nt < 2; ++nt)
      for (int blocks : {1024, 2048, 4096, 8192}) {
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(rd<false>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)w[it & 1], bytes / 16, out);
        hipEventRecord(a);
        for (int it = 0; it < 20; ++it) {
          if (nt) hipLaunchKernelGGL(rd<true>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)w[it & 1], bytes / 16, out);
          else hipLaunchKernelGGL(rd<false>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)w[it & 1], bytes / 16, out);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s data, %s loads, %5d blocks: %7.1f GB/s\n", random ? "random  " : "constant", nt ? "nt   " : "plain", blocks, bytes * 20 / (ms * 1e-3) * 1e-9);
      }
  }
  return 0;
}
