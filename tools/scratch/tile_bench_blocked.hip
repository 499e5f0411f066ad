// round 6 probe: the 128 x 256 ring tile (gemm_dev.h) under different wrappers -- one tile per workgroup vs a persistent loop, and the
// register budget (__launch_bounds__ second argument) -- at the headline shape.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off -I tutel_amd/csrc tools/scratch/tile_bench.hip -o tools/scratch/tile_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long g_probe[8];
#define GEMM_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 17) g_probe[i] = wall_clock64(); } while (0)
#define GEMM_PROBE_W_BLOCKED 1
#include "gemm_dev.h"
void tutel_set_error(const char *, ...) {}
int tutel_get_option(int) { return -1; }

template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_one(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = gridDim.x;
  int w;
  {
    const int b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, pos = b >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  gemm_big_tile<bf16_t, true, TUTEL_ACT_RELU, 4, 3, true, 128, false>(p, w / p.ntn, 0, w % p.ntn, smem);
  GEMM_PROBE(5);
}
// persistent, static assignment: workgroup b (XCD b & 7, slot b >> 3) runs items slot, slot + 32, ... of its XCD's list
template <int MINW, int ACT>
__global__ __launch_bounds__(256, MINW) void k_loop(GemmArgs p, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  for (int i = slot; i < per_xcd; i += nslot) {
    const int w = xcd * per_xcd + i;
    __syncthreads();
    gemm_big_tile<bf16_t, true, ACT, 4, 3, true, 128, false>(p, w / p.ntn, 0, w % p.ntn, smem);
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
  const int E = 64, R = 128, N = 2048, K = argc > 1 ? atoi(argv[1]) : 2048;
  uint16_t *A, *W[2], *D;
  CK(hipMalloc(&A, (size_t)E * R * K * 2)); CK(hipMalloc(&D, (size_t)E * R * N * 2));
  for (int i = 0; i < 2; ++i) CK(hipMalloc(&W[i], (size_t)E * N * K * 2));
  std::vector<uint16_t> h((size_t)E * N * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0xff);
  for (int i = 0; i < 2; ++i) CK(hipMemcpy(W[i], h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(A, h.data(), (size_t)E * R * K * 2, hipMemcpyHostToDevice));
  GemmArgs p = {};
  p.A = A; p.a_stride_e = (long long)R * K; p.a_stride_w = 0; p.a_rpw = R; p.lda = K;
  p.w_stride_e = (long long)N * K; p.ldw = K; p.bias = nullptr; p.bias_stride_e = 0;
  p.D = D; p.d_stride_e = (long long)R * N; p.d_stride_w = 0; p.d_rpw = R; p.ldd = N;
  p.E_loc = E; p.R = R; p.N = N; p.K = K; p.row_counts = nullptr; p.row_align = 1; p.a_rows = nullptr; p.a_span_bytes = R * K * 2;
  p.fits32 = true; p.rot_on = true; p.sgather = false; p.d_store = 1; p.ntm = 1; p.ntn = N / 256; p.act_rt = TUTEL_ACT_RELU;
  const size_t lds = (size_t)3 * 3 * GL_STAGE * 2 + 16;
  auto opt = [&](const void *k) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); };
  opt((const void *)k_one<2>); opt((const void *)k_one<1>); opt((const void *)k_loop<2, TUTEL_ACT_RELU>); opt((const void *)k_loop<1, TUTEL_ACT_RELU>);
  opt((const void *)k_loop<1, GEMM_ACT_RUNTIME>); opt((const void *)k_loop<2, GEMM_ACT_RUNTIME>);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = E * (N / 256);
  auto run = [&](int v, int wi) {
    p.W = W[wi];
    switch (v) {
      case 0: hipLaunchKernelGGL(k_one<2>, dim3(tiles), dim3(256), lds, 0, p); break;
      case 1: hipLaunchKernelGGL(k_one<1>, dim3(tiles), dim3(256), lds, 0, p); break;
      case 2: hipLaunchKernelGGL((k_loop<2, TUTEL_ACT_RELU>), dim3(256), dim3(256), lds, 0, p, tiles / 8); break;
      case 3: hipLaunchKernelGGL((k_loop<1, TUTEL_ACT_RELU>), dim3(256), dim3(256), lds, 0, p, tiles / 8); break;
      case 4: hipLaunchKernelGGL((k_loop<1, GEMM_ACT_RUNTIME>), dim3(256), dim3(256), lds, 0, p, tiles / 8); break;
      case 5: hipLaunchKernelGGL((k_loop<2, GEMM_ACT_RUNTIME>), dim3(256), dim3(256), lds, 0, p, tiles / 8); break;
    }
  };
  const char *names[] = {"one tile per WG, bounds(256,2) [production]", "one tile per WG, bounds(256,1)", "persistent static loop, bounds(256,2)",
                         "persistent static loop, bounds(256,1)", "persistent static loop, runtime act, bounds(256,1)", "persistent static loop, runtime act, bounds(256,2)"};
  for (int rep = 0; rep < 3; ++rep)
    for (int v = 0; v < 6; ++v) {
      for (int i = 0; i < 4; ++i) run(v, i & 1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) run(v, i & 1);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rep %d  %-55s %8.2f us per GEMM  %6.0f GB/s (weights + rows in + rows out)\n", rep, names[v], ms * 1000 / 20, ((double)E * N * K + (double)E * R * (K + N)) * 2 / (ms / 20 * 1e-3) * 1e-9);
    }
  // phase timeline of one workgroup (block 17) of the production wrapper
  for (int i = 0; i < 3; ++i) { run(0, i & 1); }
  CK(hipDeviceSynchronize());
  long long pr[8];
  CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr)));
  printf("tile phases of one workgroup (us from entry): prologue issued %.2f, first K-tile landed %.2f, last DMA issued %.2f, K loop done %.2f, epilogue done %.2f\n",
         (pr[1] - pr[0]) / 100.0, (pr[2] - pr[0]) / 100.0, (pr[3] - pr[0]) / 100.0, (pr[4] - pr[0]) / 100.0, (pr[5] - pr[0]) / 100.0);
  return 0;
}
