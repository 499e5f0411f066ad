// Dev probe (round 4): are WRITE-THROUGH stores (global_store ... sc0 sc1) of one kernel visible to PLAIN loads of the next
// kernel on the same stream, when the reading XCD already holds copies of the lines in its L2?  The question behind the reverted
// write-through peer stores of the IPC transport: there, rows stored with sc0 sc1 were read stale by the
// kernel behind the flag kernels.  Here, one process, one stream:
//     W_plain(v0) -> R(v0)            the readers' L2s now hold copies of v0 (reader block b reads what writer block b+1 wrote:
//                                     another XCD under the observed b % 8 placement)
//     W_flavour(v1) -> [delay] -> R   counts 16-byte vectors that still read v0
// flavours: plain | sc1 | sc0 sc1 | sc0 sc1 + s_waitcnt vmcnt(0) at the wave's end | nt.   delay: none, ~20 us, ~200 us of a
// one-wave spin kernel.  If the stale count falls with the delay, the stores were still in flight when the kernel "ended"; if it
// does not, the reading XCD's copies were never invalidated.
//     hipcc --offload-arch=gfx950 -O3 tools/scratch/wt_visibility.hip -o tools/scratch/wt_visibility && tools/scratch/wt_visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FL>
__global__ __launch_bounds__(256) void writer(u32x4 *buf, size_t per_block, uint32_t val) {
  u32x4 *p = buf + (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += 256) {
    u32x4 v = {val, (uint32_t)i, blockIdx.x, ~val};
    if (FL == 0) p[i] = v;
    else if (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
    else if (FL == 2 || FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
    else __builtin_nontemporal_store(v, p + i);
  }
  if (FL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void reader(const u32x4 *buf, size_t per_block, uint32_t expect, unsigned long long *stale, uint32_t *sink) {
  const int src = (blockIdx.x + 1) % gridDim.x;   // what the NEXT block id wrote: another XCD
  const u32x4 *p = buf + (size_t)src * per_block;
  unsigned bad = 0, acc = 0;
  for (size_t i = threadIdx.x; i < per_block; i += 256) {
    const u32x4 v = p[i];
    bad += (v[0] != expect) | (v[3] != ~expect);
    acc += v[1];
  }
  if (bad) atomicAdd(stale, (unsigned long long)bad);
  if (acc == 0xdeadbeefu) *sink = acc;
}

__global__ void spin(long long ticks) {   // one wave; 100 MHz wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
}

template <int FL> static void launch_writer(u32x4 *buf, int blocks, size_t per_block, uint32_t val, hipStream_t st) {
  hipLaunchKernelGGL(writer<FL>, dim3(blocks), dim3(256), 0, st, buf, per_block, val);
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 16;
  const int blocks = 2048, trials = 20;
  const size_t vecs = mb * 1024 * 1024 / 16, per_block = vecs / blocks;
  u32x4 *buf; unsigned long long *stale; uint32_t *sink;
  CK(hipMalloc(&buf, vecs * 16)); CK(hipMalloc(&stale, 8)); CK(hipMalloc(&sink, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  const char *names[5] = {"plain", "sc1", "sc0 sc1", "sc0 sc1 + vmcnt(0)", "nt"};
  const long long delays[3] = {0, 2000, 20000};
  printf("buffer %zu MB, %d blocks x 256, %d trials per cell; stale 16-byte vectors of %zu per trial\n", mb, blocks, trials, per_block * blocks);
  for (int fl = 0; fl < 5; ++fl)
    for (int d = 0; d < 3; ++d) {
      unsigned long long tot = 0, worst = 0;
      for (int t = 0; t < trials; ++t) {
        const uint32_t v0 = 1000u + 2 * t + 100000u * (fl * 3 + d), v1 = v0 + 1;
        launch_writer<0>(buf, blocks, per_block, v0, st);
        CK(hipMemsetAsync(stale, 0, 8, st));
        hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, st, buf, per_block, v0, stale, sink);
        unsigned long long warm = 0;
        CK(hipMemcpyAsync(&warm, stale, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (warm) { printf("warm-up pass read %llu stale vectors (plain stores!)\n", warm); }
        // the pass under test: no host synchronisation between the writer and the reader
        CK(hipMemsetAsync(stale, 0, 8, st));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 200LL);   // (the memset above is a blit kernel: keep it away from the writer)
        switch (fl) {
          case 0: launch_writer<0>(buf, blocks, per_block, v1, st); break;
          case 1: launch_writer<1>(buf, blocks, per_block, v1, st); break;
          case 2: launch_writer<2>(buf, blocks, per_block, v1, st); break;
          case 3: launch_writer<3>(buf, blocks, per_block, v1, st); break;
          default: launch_writer<4>(buf, blocks, per_block, v1, st); break;
        }
        if (delays[d]) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, delays[d]);
        hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, st, buf, per_block, v1, stale, sink);
        unsigned long long s = 0;
        CK(hipMemcpyAsync(&s, stale, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        tot += s; worst = s > worst ? s : worst;
      }
      printf("%-20s delay %6.0f us : stale total %10llu  worst trial %8llu\n", names[fl], delays[d] / 100.0, tot, worst);
    }
  return 0;
}
