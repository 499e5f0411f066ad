// cu_pin.hip -- dev probe: occupy N compute units for a fixed time on a given stream, so that a kernel on ANOTHER stream
// runs with N CUs taken away (what a co-running RCCL collective does to the stage GEMMs of the overlapped pipeline).
// Each block asks for 160 KB of LDS (the whole CU), so no other workgroup that needs LDS fits beside it, and sleeps.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/scratch/cu_pin.hip -o tools/scratch/libcu_pin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_pin_kernel(unsigned long long ticks, int *sink) {
  extern __shared__ unsigned char lds[];
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && sink != nullptr) sink[blockIdx.x] = lds[0];
}

extern "C" int cu_pin(int n_cus, double microseconds, int *sink, void *stream) {
  static bool optin = false;
  const int lds = 160 * 1024;
  if (!optin) {
    if (hipFuncSetAttribute((const void *)cu_pin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    optin = true;
  }
  hipLaunchKernelGGL(cu_pin_kernel, dim3(n_cus), dim3(64), lds, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0), sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
