// routing_dev.h -- device functions of the location step that more than one translation unit runs (no relocatable device code:
// they are header templates).  routing.hip's location_kernel is their main user; dispatch.hip's decode kernel runs the one-block
// "finish" (dispatch_count, max load, gshard loss) in an extra block when the locations themselves were computed inside the first
// expert GEMM (the fused-location path, expert_gemm.hip / ep.hip round 5).
#pragma once
#include "common.h"

#define RT_THREADS 256
#define RT_WAVES 4

// COH = the fused kernel: data another block of the SAME launch wrote (or will overwrite) moves with device-scope relaxed atomics
// (sc1 accesses: coherent across the XCDs' L2s without a cache-wide write-back / invalidate -- a device-scope release + acquire
// fence pair around the barrier cost 25 us of a 40 us kernel, profiles/r03_routing_fused.txt).
template <bool COH> __device__ __forceinline__ int ld_i32(const int32_t *p) {
  return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH> __device__ __forceinline__ float ld_f32(const float *p) {
  return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH> __device__ __forceinline__ void st_i32(int32_t *p, int v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH> __device__ __forceinline__ void st_f32(float *p, float v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// phases 1 + 2: s_cur[j][e] = absolute location of the tile's first token that picks (choice j, expert e); s_tot[j][e] = totals
template <int NW, bool COH = false>
__device__ __forceinline__ void loc_prefix(int tid, int b, int E, int k, int ntiles, const int32_t *__restrict__ ws_hist,
                                           int32_t *s_cur, int32_t *s_tot, int32_t *__restrict__ dispatch_count) {
  const int lane = tid & 63, wid = tid >> 6, kE = k * E;
  // base[j][e] = sum over earlier tiles, tot[j][e] = sum over all tiles.  The tile axis is
  // split over the waves and unrolled so the (<=128) dependent-free L2 loads overlap.
  for (int i = tid; i < kE; i += NW * 64) { s_cur[i] = 0; s_tot[i] = 0; }
  __syncthreads();
  for (int i = lane; i < kE; i += 64) {
    int base = 0, tot = 0;
    for (int tl0 = wid; tl0 < ntiles; tl0 += NW * 16) {
      int h[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int tl = tl0 + u * NW;
        h[u] = (tl < ntiles) ? ld_i32<COH>(ws_hist + (size_t)tl * kE + i) : 0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int tl = tl0 + u * NW;
        tot += h[u];
        if (tl < b) base += h[u];
      }
    }
    if (wid < ntiles) {   // (waves past the tile count hold zeros)
      atomicAdd(&s_cur[i], base);
      atomicAdd(&s_tot[i], tot);
    }
  }
  __syncthreads();
  // choice j is queued after ALL tokens' choices < j (fast_dispatch.py:165-169)
  for (int e = tid; e < E; e += NW * 64) {
    int acc = 0;
    for (int j = 0; j < k; ++j) {
      s_cur[j * E + e] += acc;
      acc += s_tot[j * E + e];
    }
    if (b == 0) dispatch_count[e] = acc;
  }
  __syncthreads();
}

// phase 4 (one block): max count and gshard loss.  Column sums: `parts` threads per expert, each a contiguous tile range in
// fixed order, combined in fixed order.  The arithmetic is done by the block's first RT_THREADS threads in BOTH kernels, so the
// loss does not depend on which kernel computed it (deterministic, bit for bit).
template <int NW, bool COH = false>
__device__ __forceinline__ void loc_finish(int tid, int Tn, int E, int k, int ntiles, const float *__restrict__ ws_colsum,
                                           const int32_t *s_tot, float *s_parts, float *s_red, int *s_redi, const float *cs_first,
                                           bool cs_early, int32_t *__restrict__ stats, void *__restrict__ l_aux, int l_aux_dtype) {
  const int lane = tid & 63, wid = tid >> 6;
  const bool act = tid < RT_THREADS;
  const int cs_parts = (E >= RT_THREADS) ? 1 : (RT_THREADS / E);
  const int cs_per = (ntiles + cs_parts - 1) / cs_parts;
  int mx = 0;
  if (act)
    for (int e = tid; e < E; e += RT_THREADS) {
      int acc = 0;
      for (int j = 0; j < k; ++j) acc += s_tot[j * E + e];
      mx = max(mx, acc);
    }
  float part = 0.f;
  if (l_aux != nullptr) {
    __syncthreads();
    const int parts = cs_parts, per = cs_per;
    if (act)
      for (int w = tid; w < parts * E; w += RT_THREADS) {
        const int e = w % E, pt = w / E;
        const int a = pt * per, z = min(ntiles, a + per);
        float me = 0.f;
        for (int tl0 = a; tl0 < z; tl0 += 16) {
          float cs[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            cs[u] = (cs_early && tl0 == a) ? cs_first[u] : ((tl0 + u < z) ? ld_f32<COH>(ws_colsum + (size_t)(tl0 + u) * E + e) : 0.f);
#pragma unroll
          for (int u = 0; u < 16; ++u) me += cs[u];
        }
        s_parts[pt * E + e] = me;
      }
    __syncthreads();
    if (act)
      for (int e = tid; e < E; e += RT_THREADS) {
        float me = 0.f;
        for (int pt = 0; pt < parts; ++pt) me += s_parts[pt * E + e];
        float ce = (float)s_tot[e] * ((float)E / (float)Tn);
        part += me * ce;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, __shfl_xor(mx, o, 64));
    part += __shfl_xor(part, o, 64);
  }
  if (act && lane == 0) { s_red[wid] = part; s_redi[wid] = mx; }
  __syncthreads();
  if (tid == 0) {
    float p = 0.f;
    int m2 = 0;
    for (int w = 0; w < RT_WAVES; ++w) { p += s_red[w]; m2 = max(m2, s_redi[w]); }
    if (stats != nullptr) stats[0] = m2;
    if (l_aux != nullptr) {
      const float la = p / (float)Tn;
      if (l_aux_dtype == TUTEL_F32) reinterpret_cast<float *>(l_aux)[0] = la;
      else if (l_aux_dtype == TUTEL_BF16) reinterpret_cast<uint16_t *>(l_aux)[0] = f32_to_bf16_bits(la);
      else reinterpret_cast<_Float16 *>(l_aux)[0] = (_Float16)la;
    }
  }
}


// dispatch_count / max load / gshard loss from the per-tile histograms and score column sums the top-k kernel left in the routing
// workspace: exactly what block 0 of location_kernel computes (same functions, same thread count, same order: same bits), for
// callers that do not run location_kernel.  One block of RT_THREADS threads; `smem` >= route_finish_lds(E, k) bytes.
struct RouteFinish {
  int on, Tn, E, k, ntiles;
  const int32_t *ws_hist;
  const float *ws_colsum;
  int32_t *dispatch_count, *stats;
  void *l_aux;
  int l_aux_dtype;
};
static inline size_t route_finish_lds(int E, int k) { return ((size_t)2 * k * E + (size_t)(E > RT_THREADS ? E : RT_THREADS)) * 4; }
__device__ __forceinline__ void route_finish_block(const RouteFinish &f, unsigned char *smem) {
  int32_t *s_cur = reinterpret_cast<int32_t *>(smem);
  int32_t *s_tot = s_cur + (size_t)f.k * f.E;
  float *s_parts = reinterpret_cast<float *>(smem) + (size_t)2 * f.k * f.E;
  __shared__ float s_red[RT_WAVES];
  __shared__ int s_redi[RT_WAVES];
  const int tid = threadIdx.x;
  float cs_first[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) cs_first[u] = 0.f;
  loc_prefix<RT_WAVES>(tid, 0, f.E, f.k, f.ntiles, f.ws_hist, s_cur, s_tot, f.dispatch_count);
  loc_finish<RT_WAVES>(tid, f.Tn, f.E, f.k, f.ntiles, f.ws_colsum, s_tot, s_parts, s_red, s_redi, cs_first, false, f.stats, f.l_aux, f.l_aux_dtype);
}
