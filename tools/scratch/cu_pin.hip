// cu_pin.hip -- dev probe: occupy N compute units for a fixed time on a given stream, so that a kernel on ANOTHER stream
// runs with N CUs taken away (what a co-running RCCL collective does to the stage GEMMs of the overlapped pipeline).
// Each block asks for 160 KB of LDS (the whole CU), so no other workgroup that needs LDS fits beside it, and sleeps.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/scratch/cu_pin.hip -o tools/scratch/libcu_pin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_pin_kernel(unsigned long long ticks, long long *sink) {
  extern __shared__ unsigned char lds[];
  const unsigned long long t0 = wall_clock64();  // 100 MHz, one clock for the whole device
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && sink != nullptr) {   // when this block really ran: [start, end] in device wall-clock ticks
    sink[2 * blockIdx.x] = (long long)t0 + (lds[0] & 0);
    sink[2 * blockIdx.x + 1] = (long long)wall_clock64();
  }
}

__global__ void cu_stamp_kernel(long long *out) { out[0] = (long long)wall_clock64(); }

extern "C" int cu_stamp(long long *out, void *stream) {
  hipLaunchKernelGGL(cu_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int cu_pin(int n_cus, double microseconds, long long *sink, void *stream) {
  static bool optin = false;
  const int lds = 160 * 1024;
  if (!optin) {
    if (hipFuncSetAttribute((const void *)cu_pin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    optin = true;
  }
  hipLaunchKernelGGL(cu_pin_kernel, dim3(n_cus), dim3(64), lds, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0), sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
