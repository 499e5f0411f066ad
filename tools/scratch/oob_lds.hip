// Dev probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS into LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* a, uint32_t* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* l = (uint32_t*)smem;
  for (int i = threadIdx.x; i < 1024; i += 64) l[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  int voff = (threadIdx.x & 1) ? (int)0x80000000 : (int)(threadIdx.x * 16);   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = l[i];
}
int main() {
  uint32_t h[1024], *d, *o, ho[256];
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 1024);
  hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 8; ++lane) printf("lane %d: %08x %08x %08x %08x\n", lane, ho[lane*4], ho[lane*4+1], ho[lane*4+2], ho[lane*4+3]);
  return 0;
}
