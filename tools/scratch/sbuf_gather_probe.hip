#include <hip/hip_runtime.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ void k(const int *m, int n, int *out) {
  const unsigned long long a_ = (unsigned long long)m;
  v4i rs = {(int)(unsigned)a_, (int)((a_ >> 32) & 0xffff), n * 4, 0x00020000};
  const int base = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 128 + blockIdx.x * 512;
  v8i s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("s_buffer_load_dwordx8 %0, %1, %2" : "=&s"(s[i]) : "s"(rs), "s"(base + i * 32));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]));
  const int rl = (threadIdx.x & 63) >> 3;
  int acc = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = s[i][0];
#pragma unroll
    for (int j = 1; j < 8; ++j) q = (rl == j) ? s[i][j] : q;
    acc += q * (i + 1);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int n = 1000, nb = 3;  // 3 blocks x 256 threads: 4 waves x 32 entries per block = 128 -> reads up to 384; n = 300 to test OOB -> 0
  int *m, *out; hipMalloc(&m, 4096 * 4); hipMalloc(&out, nb * 256 * 4);
  int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 1;
  hipMemcpy(m, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, m, 300, out);
  int r[nb * 256]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; ++b) for (int t = 0; t < 256; ++t) {
    int w = t >> 6, rl = (t & 63) >> 3, want = 0;
    for (int i = 0; i < 4; ++i) { int idx = b * 128 + w * 32 + i * 8 + rl; int v = idx < 300 ? idx + 1 : 0; want += v * (i + 1); }
    if (r[b * 256 + t] != want) { if (bad < 5) printf("b%d t%d got %d want %d\n", b, t, r[b*256+t], want); ++bad; }
  }
  printf("bad=%d\n", bad); return bad != 0;
}
