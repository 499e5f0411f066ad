// gate_proj.hip -- the gate projection of a 16-bit linear gate on MFMA for gfx950.
//
//   logits[t, e] = sum_m x[t, m] * wg[e, m]        x [T, M], wg [E, M] (nn.Linear weight), bf16 / fp16, fp32 accumulate
//
// Replaces `F.linear(x, self.wg.weight)` of LinearTopKGate.forward (tutel/gates/top.py:20-22) when the gate runs in the
// experts' 16-bit dtype (fp32_gate=False, the benchmarked configuration).  The shape is a skinny GEMM -- 4096 x 64 outputs
// over K = 2048 at the headline: 16.8 MB of tokens read once, 1 GFLOP -- so the only thing that matters is how many bytes
// are in flight: the library's kernel for it (hipBLASLt MT64x16x128, no split) takes 9.8 us = 1.7 TB/s.  Round 4's attempt
// to do the projection inside the top-k kernel (64 workgroups, MFMA fragments loaded straight from global memory, 32 bytes
// per row and instruction) was bound by the L1 request rate: 38.5 us.  This kernel is the other design:
//   * split-K across workgroups: block = (64-token tile, K slice of NP * 64), T/64 x S blocks >= 256 at the headline, the
//     S blocks of a token tile spread over the XCDs by `b % S` so that each XCD's L2 holds only its own wg slices;
//   * ONE shot of LDS-DMA per block: the whole [64][NP*64] token slab and the [E][NP*64] weight slab (128 KB at NP = 8) are
//     issued by `global_load_lds` (16 B per lane, 1 KB per wave instruction, rows of 128 contiguous bytes) before anything
//     waits -- 32 MB in flight chip-wide, no pipeline loop, no per-tile barrier;
//   * LDS image = the expert GEMM's: panels of [rows][64 k], 16-byte chunk c of row r at c ^ ((r >> 1) & 7) (the lane picks
//     the global chunk, the LDS side of the DMA is linear), conflict-free ds_read_b128 fragments;
//   * v_mfma_f32_32x32x16, weights as the A operand: a lane ends with 4 consecutive experts of one token per register group
//     -> 16-byte stores of the fp32 partial sums part[s][t][e];
//   * no atomics: the S partial sums are added in split order by the consumer (gate_topk_quad_kernel, routing.hip) and
//     rounded once to the logits dtype, so the logits -- and everything routed from them -- are reproducible bit for bit.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float gp_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 gp_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 gp_f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t gp_u32x4;
typedef __attribute__((ext_vector_type(4))) float gp_f32x4;

template <typename T> struct GpMma;
template <> struct GpMma<bf16_t> {
  __device__ static __forceinline__ gp_f32x16 run(gp_u32x4 a, gp_u32x4 b, gp_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gp_bf16x8, a), __builtin_bit_cast(gp_bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct GpMma<f16_t> {
  __device__ static __forceinline__ gp_f32x16 run(gp_u32x4 a, gp_u32x4 b, gp_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gp_f16x8, a), __builtin_bit_cast(gp_f16x8, b), c, 0, 0, 0);
  }
};

#define GP_TM 64        // tokens per block
#define GP_THREADS 256  // 4 waves: 2 token halves x 2 expert groups
#define GP_PANEL 4096   // elements of a [64][64] panel

__device__ __forceinline__ void gp_dma16(const uint16_t *g, uint16_t *l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// ER = expert rows staged per panel (64 | 128), NP = 64-wide K panels per block
template <typename T, int ER, int NP>
__global__ __launch_bounds__(GP_THREADS) void gate_proj_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ wg,
                                                              int Tn, int M, int E, int S, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *sX = reinterpret_cast<uint16_t *>(smem);  // [NP][64][64]
  uint16_t *sW = sX + NP * GP_PANEL;                  // [NP][ER][64]
  constexpr int WPW = ER / 32;                        // weight pieces (8 rows = 1 KB) per wave and panel
  constexpr int NPW = ER / 64;                        // 32-expert MFMA tiles per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, s = b % S, tile = b / S;
  const int t0 = tile * GP_TM;
  const int kp0 = s * NP;
  const int npan = min(NP, M / 64 - kp0);  // block-uniform; >= 1 by construction of S

  // ---- the one shot of DMA: panel-major, so the data of panel p has landed once the wave's first (p + 1) * (2 + WPW) ops have
  const int rl = lane >> 3;  // row of the piece
  {
    const uint16_t *xs[2], *ws[WPW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wid * 2 + i) + rl;
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      xs[i] = x + (size_t)min(t0 + r, Tn - 1) * M + (size_t)kp0 * 64 + c * 8;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int r = 8 * (wid * WPW + i) + rl;
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      ws[i] = wg + (size_t)min(r, E - 1) * M + (size_t)kp0 * 64 + c * 8;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p < npan) {
#pragma unroll
        for (int i = 0; i < 2; ++i) gp_dma16(xs[i] + p * 64, sX + p * GP_PANEL + (wid * 2 + i) * 512);
#pragma unroll
        for (int i = 0; i < WPW; ++i) gp_dma16(ws[i] + p * 64, sW + p * (ER * 64) + (wid * WPW + i) * 512);
      }
    }
  }

  const int tt = wid & 1, eg = wid >> 1;
  const int l31 = lane & 31, kg = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  int frag_k[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_k[kk] = (((kk * 2 + kg) ^ sw) << 3);
  const int x_row = (tt * 32 + l31) * 64;
  const int w_row = (eg * NPW * 32 + l31) * 64;

  gp_f32x16 acc[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // Consume panel by panel while the later panels are still landing: a wave's DMA ops retire in issue order (panel-major, OPS per
  // panel), so `s_waitcnt vmcnt((NP - 1 - p) * OPS)` says this wave's pieces of panel p are in LDS and the bare s_barrier after it
  // says every wave's are.  The fragment reads and MFMAs of panel p run under the
  // arrival of panels p + 1 ..; only the last panel's arithmetic is exposed.  A block whose K slice is short (npan < NP: the
  // counts above would be wrong) drains everything first.
  constexpr int OPS = 2 + WPW;
#define GP_FRAGS(P, FA, FW)                                                                          \
  do {                                                                                               \
    const uint16_t *cx_ = sX + (P) * GP_PANEL + x_row, *cw_ = sW + (P) * (ER * 64) + w_row;          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                               \
      FA[kk] = *reinterpret_cast<const gp_u32x4 *>(cx_ + frag_k[kk]);                                \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i)                                                \
        FW[kk][i] = *reinterpret_cast<const gp_u32x4 *>(cw_ + i * 32 * 64 + frag_k[kk]);             \
    }                                                                                                \
  } while (0)
#define GP_MMA(FA, FW)                                                                               \
  do {                                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                 \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i) acc[i] = GpMma<T>::run(FW[kk][i], FA[kk], acc[i]); \
  } while (0)
  if (npan == NP) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP - 1 - p) * OPS) : "memory");
      __builtin_amdgcn_s_barrier();
      gp_u32x4 fa[4], fw[4][NPW];
      GP_FRAGS(p, fa, fw);
      __builtin_amdgcn_sched_barrier(0);
      GP_MMA(fa, fw);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    __syncthreads();  // with DMA in flight: s_waitcnt vmcnt(0) + barrier -- every wave's pieces are in LDS
    for (int p = 0; p < npan; ++p) {
      gp_u32x4 fa[4], fw[4][NPW];
      GP_FRAGS(p, fa, fw);
      __builtin_amdgcn_sched_barrier(0);
      GP_MMA(fa, fw);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GP_FRAGS
#undef GP_MMA

  // ---- partial sums: lane = token t0 + 32 tt + l31; register group rg of tile i = experts (eg NPW + i) 32 + 8 rg + 4 kg + 0..3
  const int t = t0 + tt * 32 + l31;
  if (t < Tn) {
    float *row = part + ((size_t)s * Tn + t) * E;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = (eg * NPW + i) * 32 + rg * 8 + kg * 4;
        if (n < E) {
          gp_f32x4 v = {acc[i][rg * 4 + 0], acc[i][rg * 4 + 1], acc[i][rg * 4 + 2], acc[i][rg * 4 + 3]};
          *reinterpret_cast<gp_f32x4 *>(row + n) = v;
        }
      }
  }
}

// ---- split policy: a pure function of the shape (the consumer must agree on S) -----------------------------------------------
struct GpPlan {
  int er, np, splits;
};
static bool gp_plan(int T, int M, int E, int dtype, GpPlan *out) {
  if (!(dtype == TUTEL_BF16 || dtype == TUTEL_F16) || T < 1 || E < 1 || E > 128 || (E & 3) || M < 64 || (M & 63)) return false;
  const int P = M / 64, ntt = (T + GP_TM - 1) / GP_TM;
  const int er = E <= 64 ? 64 : 128;
  int np = er == 64 ? 8 : 4;              // 128 KB / 96 KB of LDS: one block per CU
  while (np > 2 && (long long)ntt * ((P + np - 1) / np) < 256) np >>= 1;  // few token tiles: more, thinner K slices
  const int S = (P + np - 1) / np;
  if (S > 64) return false;
  out->er = er;
  out->np = np;
  out->splits = S;
  return true;
}

extern "C" int tutel_amd_gate_proj_splits(int T, int M, int E, int dtype) {
  GpPlan pl;
  return gp_plan(T, M, E, dtype, &pl) ? pl.splits : 0;
}

template <typename T, int ER, int NP>
static int gp_launch(const void *x, const void *wg, int Tn, int M, int E, int S, float *part, hipStream_t st) {
  const size_t lds = (size_t)NP * (GP_PANEL + ER * 64) * 2;
  // idempotent and cheap; called per launch rather than remembered per (kernel, device) in a table shared between host threads
  if (lds > 65536) (void)hipFuncSetAttribute((const void *)gate_proj_kernel<T, ER, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntt = (Tn + GP_TM - 1) / GP_TM;
  hipLaunchKernelGGL((gate_proj_kernel<T, ER, NP>), dim3(ntt * S), dim3(GP_THREADS), lds, st, (const uint16_t *)x, (const uint16_t *)wg, Tn, M,
                     E, S, part);
  TUTEL_CHECK_LAUNCH("tutel_amd_gate_proj");
  return 0;
}

template <typename T>
static int gp_dispatch(const GpPlan &pl, const void *x, const void *wg, int Tn, int M, int E, float *part, hipStream_t st) {
  if (pl.er == 64) {
    if (pl.np == 8) return gp_launch<T, 64, 8>(x, wg, Tn, M, E, pl.splits, part, st);
    if (pl.np == 4) return gp_launch<T, 64, 4>(x, wg, Tn, M, E, pl.splits, part, st);
    return gp_launch<T, 64, 2>(x, wg, Tn, M, E, pl.splits, part, st);
  }
  if (pl.np == 4) return gp_launch<T, 128, 4>(x, wg, Tn, M, E, pl.splits, part, st);
  return gp_launch<T, 128, 2>(x, wg, Tn, M, E, pl.splits, part, st);
}

extern "C" int tutel_amd_gate_proj(const void *x, const void *wg, int dtype, int T, int M, int E, float *partials,
                                   size_t partial_bytes, tutel_stream_t stream) {
  GpPlan pl;
  if (T == 0) return 0;
  if (!gp_plan(T, M, E, dtype, &pl)) {
    tutel_set_error("tutel_amd_gate_proj: shape not covered (T=%d, M=%d, E=%d, dtype=%d): needs a 16-bit dtype, E <= 128, E %% 4 == 0, "
                    "M %% 64 == 0 -- use a library GEMM + tutel_amd_gate_topk", T, M, E, dtype);
    return TUTEL_AMD_ENOTSUP;
  }
  TUTEL_REQUIRE(x && wg && partials, "tutel_amd_gate_proj: null pointer");
  TUTEL_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wg & 15) == 0 && ((uintptr_t)partials & 15) == 0,
                "tutel_amd_gate_proj: x, wg and partials must be 16-byte aligned");
  TUTEL_REQUIRE(partial_bytes >= (size_t)pl.splits * T * E * sizeof(float), "tutel_amd_gate_proj: partials buffer too small (%zu < %zu)",
                partial_bytes, (size_t)pl.splits * T * E * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  StageScope stage(TUTEL_STAGE_GATE_PROJ, st);
  if (dtype == TUTEL_BF16) return gp_dispatch<bf16_t>(pl, x, wg, T, M, E, partials, st);
  return gp_dispatch<f16_t>(pl, x, wg, T, M, E, partials, st);
}

// ---- memory-side cache warm-up ---------------------------------------------------------------------------------------------------
// Plain loads of a byte range, nothing written: lines are allocated in the 256 MiB Infinity Cache (and pass through an L2).  The
// routing kernels (gate projection, top-k, locations) are latency chains on 64 workgroups that leave HBM idle for ~25 us while
// the first expert GEMM's weights -- which depend on nothing -- wait; a caller may run this on a second stream meanwhile.
#define CW_THREADS 512
#define CW_UNR 8   // 16-byte loads in flight per lane: 64 KB per block -- a stream needs ~10 MB in flight chip-wide (5 TB/s x 2 us)
__global__ __launch_bounds__(CW_THREADS) void cache_warm_kernel(const unsigned char *__restrict__ base, size_t chunk_vec, int n_chunks,
                                                                size_t stride_bytes, uint32_t *__restrict__ sink) {
  // blocks are dealt round-robin to the chunks: block b reads chunk b % n_chunks with the blocks (b / n_chunks) of that chunk
  const int c = blockIdx.x % n_chunks, bi = blockIdx.x / n_chunks, nbi = (gridDim.x - c + n_chunks - 1) / n_chunks;
  const gp_u32x4 *p = reinterpret_cast<const gp_u32x4 *>(base + (size_t)c * stride_bytes);
  const size_t stride = (size_t)nbi * CW_THREADS;
  size_t i = (size_t)bi * CW_THREADS + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (CW_UNR - 1) * stride < chunk_vec; i += CW_UNR * stride) {
    gp_u32x4 v[CW_UNR];
#pragma unroll
    for (int u = 0; u < CW_UNR; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < CW_UNR; ++u) acc ^= v[u][u & 3];
  }
  for (; i < chunk_vec; i += stride) acc ^= p[i][0];
  if (acc == 0x9e3779b9u && sink != nullptr) *sink = acc;  // keeps the loads alive; practically never true, harmless if it is
}

extern "C" int tutel_amd_cache_warm(const void *p, size_t chunk_bytes, int n_chunks, size_t stride_bytes, int blocks, void *sink4,
                                    tutel_stream_t stream) {
  if (chunk_bytes < 16 || n_chunks < 1) return 0;
  TUTEL_REQUIRE(p != nullptr && ((uintptr_t)p & 15) == 0 && (stride_bytes & 15) == 0, "tutel_amd_cache_warm: pointer and stride must be 16-byte aligned");
  if (blocks < n_chunks) blocks = n_chunks > 256 ? n_chunks : 256;
  hipLaunchKernelGGL(cache_warm_kernel, dim3(blocks), dim3(CW_THREADS), 0, (hipStream_t)stream, (const unsigned char *)p, chunk_bytes / 16, n_chunks,
                     stride_bytes, (uint32_t *)sink4);
  TUTEL_CHECK_LAUNCH("tutel_amd_cache_warm");
  return 0;
}
