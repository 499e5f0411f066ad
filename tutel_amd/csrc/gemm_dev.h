// gemm_dev.h -- device-side pieces of the grouped expert GEMMs that more than one translation unit runs (no relocatable device code:
// they are header templates): argument block, MFMA wrappers, the epilogues, LDS-DMA helpers and the 128 / 256-row ring tile as a
// device function.  expert_gemm.hip launches the tile one per workgroup; expert_ffn.hip (round 6) runs it item after item inside one
// persistent launch for fc1 -> activation -> fc2.
#pragma once
#include "common.h"

// probe points of the ring tile (tools/scratch/tile_bench.hip defines this to timestamp them; empty in the library)
#ifndef GEMM_PROBE
#define GEMM_PROBE(i)
#endif

struct GemmArgs;
// host side, expert_gemm.hip: argument checks + the GemmArgs block of one grouped GEMM (0: filled, 1: empty problem, < 0: error);
// the > 64 KB dynamic-LDS opt-in per (kernel, device)
int tutel_gemm_args(const void *A, int64_t a_stride_e, int64_t a_stride_w, int a_rows_per_w, int lda, const void *W, int w_kmajor,
                    int64_t w_stride_e, int ldw, const void *bias, int64_t bias_stride_e, void *D, int64_t d_stride_e, int64_t d_stride_w,
                    int d_rows_per_w, int ldd, int E_loc, int R, int N, int K, int dtype, int act, const int32_t *row_counts, int row_align,
                    const int32_t *a_rows, int a_rows_mod, const void *a_zero, const void *mul, const uint64_t *d_peer, int64_t d_peer_off,
                    const PeerCanary *d_can, const uint8_t *fl_idx8, int fl_n, int32_t *fl_loc, GemmArgs *out);
bool tutel_lds_optin(const void *kern, size_t lds);

#define GM_BM 128
#define GM_BN 128
#define GM_THREADS 256
#define GM_LDN (GM_BN + 32)   // elements per LDS row of the [k][n] weight tile (320 B)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef int sgv4 __attribute__((ext_vector_type(4)));
typedef int sgv8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                 __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

#define GEMM_ACT_RUNTIME 99   // tile instantiated once for every activation: the epilogue branches on GemmArgs::act_rt (block-uniform)
template <int ACT> __device__ __forceinline__ float activate(float v) {
  if (ACT == TUTEL_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == TUTEL_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  if (ACT == TUTEL_ACT_SILU) return v / (1.f + expf(-v));
  return v;
}

// The lane's 8 bias 4-vectors (2 N-subtiles x 4 register groups) are fetched BEFORE the K loop,
// branch-free with clamped addresses: in the epilogue the same loads sat inside divergent
// `continue` branches and hipcc serialised them -- 16 x (global_load_dwordx2 ; s_waitcnt vmcnt(0))
// per block, i.e. 16 exposed memory round trips shared by every variant of the kernel.
#define GM_PRELOAD_BIAS() GM_PRELOAD_BIAS_N(2)
#define GM_PRELOAD_BIAS_N(NI_)                                                                    \
  uint2 bias_r[NI_][4];                                                                           \
  {                                                                                               \
    const uint16_t *be_ = p.bias ? reinterpret_cast<const uint16_t *>(p.bias) + (size_t)e * p.bias_stride_e : nullptr; \
    _Pragma("unroll") for (int ni_ = 0; ni_ < NI_; ++ni_)                                         \
      _Pragma("unroll") for (int rg_ = 0; rg_ < 4; ++rg_) {                                       \
        int n_ = n0 + wn * (NI_ * 32) + ni_ * 32 + rg_ * 8 + kg * 4;                              \
        n_ = n_ < p.N ? n_ : p.N - 4;                                                             \
        bias_r[ni_][rg_] = be_ ? *reinterpret_cast<const uint2 *>(be_ + n_) : make_uint2(0u, 0u); \
      }                                                                                           \
  }

struct GemmArgs {
  const void *A; long long a_stride_e, a_stride_w; int a_rpw, lda;
  const void *W; long long w_stride_e; int ldw;
  const void *bias; long long bias_stride_e;
  void *D; long long d_stride_e, d_stride_w; int d_rpw, ldd;
  int E_loc, R, N, K;
  const int32_t *row_counts; int row_align;
  const int32_t *a_rows; int a_rows_mod; const void *a_zero;  // optional row gather for A (fused fast_encode)
  int a_span_bytes;                                           // gather: bytes of the token array (a_rows_mod rows)
  bool fits32;                                                // operands addressable with 32-bit byte offsets
  bool rot_on;                                                // K-tile rotation (see launch_gemm)
  bool sgather;                                               // ring kernels: slot-map entries through the scalar cache (TUTEL_OPT_GEMM_GATHER)
  int d_store;                                                // LDS epilogue: 1 write-through (sc0 sc1) stores, 0 plain (write-back), 2 non-temporal (TUTEL_OPT_GEMM_STORE)
  const uint8_t *fl_idx8; int fl_n; int32_t *fl_loc;          // fused location (FL kernels): byte copy of idx [k*T], its length, loc out
  const void *mul;                                            // optional epilogue multiplier, D's layout
  const uint64_t *d_peer; long long d_peer_off;               // optional: rows of source rank w go to d_peer[w] + d_peer_off (bytes)
  PeerCanary d_can;                                           // peer stores: epoch canaries written behind the rows (common.h)
  float *sk_ws; uint32_t *sk_flags;                           // split-K ping-pong kernel: partial accumulators + hand-over flags (see launch_pp_splitk)
  int ntm, ntn;
  int act_rt;                                                 // the activation when the tile is instantiated with GEMM_ACT_RUNTIME (expert_ffn.hip)
};

// address of output row m of expert e (elements of 2 bytes).  Plain: D + e*stride_e + (m / rpw)*stride_w + (m % rpw)*ldd.
// Peer stores (IPC transport of the expert-parallel pipeline, ep.hip): the rows that came from source rank w = m / rpw are
// written into THAT rank's return buffer, d_peer[w] + d_peer_off, where the second all-to-all would have delivered them.
__device__ __forceinline__ uint16_t *gemm_out_row(const GemmArgs &p, int e, int m) {
  const int w = m / p.d_rpw, l = m % p.d_rpw;
  if (p.d_peer != nullptr)
    return reinterpret_cast<uint16_t *>(p.d_peer[w] + p.d_peer_off) + (size_t)e * p.d_stride_e + (size_t)l * p.ldd;
  return reinterpret_cast<uint16_t *>(p.D) + (size_t)e * p.d_stride_e + (size_t)w * p.d_stride_w + (size_t)l * p.ldd;
}

// ---- epilogue shared by both kernels: lane holds, per accumulator, row m = l31, features
// 8*rg + 4*kg + 0..3.  D = act(acc + bias) [* mul], rounded once to T; `mul` (optional) has D's
// layout and is the gating operand of a GLU expert (llama_ffn.py:40).
template <typename T, int ACT, int NI = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &p, f32x16 (&acc)[NI][2], uint2 (&bias_r)[NI][4],
                                              int e, int m0, int n0, int wm, int wn, int l31, int kg,
                                              int row_limit) {
  uint16_t *De = reinterpret_cast<uint16_t *>(p.D) + (size_t)e * p.d_stride_e;
  const uint16_t *Me = p.mul ? reinterpret_cast<const uint16_t *>(p.mul) + (size_t)e * p.d_stride_e : nullptr;
  const bool has_bias = p.bias != nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + l31;
    if (m >= row_limit) continue;
    const size_t roff = (size_t)(m / p.d_rpw) * p.d_stride_w + (size_t)(m % p.d_rpw) * p.ldd;
    uint16_t *drow = p.d_peer != nullptr ? gemm_out_row(p, e, m) : De + roff;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * (NI * 32) + ni * 32 + rg * 8 + kg * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][rg * 4 + r];
        if (has_bias) {
          const uint2 bb = bias_r[ni][rg];
          uint16_t b4[4] = {(uint16_t)(bb.x & 0xffff), (uint16_t)(bb.x >> 16), (uint16_t)(bb.y & 0xffff), (uint16_t)(bb.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T tb;
            __builtin_memcpy(&tb, &b4[r], 2);
            v[r] += Elem<T>::to_f32(tb);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = activate<ACT>(v[r]);
        if (Me) {
          const uint2 mm = *reinterpret_cast<const uint2 *>(Me + roff + n);
          uint16_t m4[4] = {(uint16_t)(mm.x & 0xffff), (uint16_t)(mm.x >> 16), (uint16_t)(mm.y & 0xffff), (uint16_t)(mm.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T tm;
            __builtin_memcpy(&tm, &m4[r], 2);
            v[r] *= Elem<T>::to_f32(tm);
          }
        }
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T tv = Elem<T>::from_f32(v[r]);
          __builtin_memcpy(&o[r], &tv, 2);
        }
        uint2 ov;
        ov.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        ov.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2 *>(drow + n) = ov;
      }
    }
  }
  if (p.d_peer != nullptr) peer_canary_store(p.d_peer, p.d_can);
}

// fused fast_encode: the slot-map entries of the lane's 4 token-tile pieces (rows gr[0..3] of expert e), fetched TOGETHER.  Written
// per piece inside `if (p.a_rows)` branches, hipcc emitted one global load + s_waitcnt vmcnt(0) per piece (and again for the
// out-of-range test of the buffer-descriptor path): eight dependent L2 round trips in front of the first DMA of every block --
// about 2 us per block with one block per CU, the 4 us by which fc1 (gather) trailed fc2 at the headline shape.
__device__ __forceinline__ void gather_rows4(const GemmArgs &p, int e, const int (&gr)[4], int (&q)[4]) {
  q[0] = q[1] = q[2] = q[3] = 0;
  if (p.a_rows != nullptr) {
    const int32_t *m = p.a_rows + (size_t)e * p.R;
    const int q0 = m[gr[0]], q1 = m[gr[1]], q2 = m[gr[2]], q3 = m[gr[3]];
    q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
  }
}

// streamed-once weight loads may bypass cache allocation (each W byte is read by exactly one CU)
template <bool NT> __device__ __forceinline__ u32x4 ld16(const uint16_t *p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
  return *reinterpret_cast<const u32x4 *>(p);
}

#define GL_BK 64
#define GL_STAGE (GM_BM * GL_BK)  // elements per [128][64] tile = 8192 (16 KB); the [64][128] tile is the same size

__device__ __forceinline__ void glds16(const uint16_t *g, uint16_t *l, bool nt) {
  if (nt)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 2);
  else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// 16 bytes per lane, global -> LDS, through a buffer descriptor: address = base + voff (VGPR) + soff (SGPR)
template <bool NT>
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint16_t *l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, NT ? 2 : 0);
}

// ---- epilogue through LDS for the 8-wave 256 x 256 kernels: the MFMA layout gives a lane 4 consecutive
// features of ONE row per register group, i.e. 8-byte stores 32 rows apart (32 store instructions per wave,
// every 64-byte sector assembled from 4 instructions).  With one block per CU and all blocks finishing
// together that store tail is fully exposed: 13-14 us of 68 at 8 x 1024 x 2048 x 2048 (ablation, tools/pp_probe.py).
// Here each wave rounds its 64 x 128 sub-tile into a private LDS region (row pitch 272 B: the 8-byte writes of
// 16 lanes spread over 8 bank pairs, the 16-byte reads of a row are contiguous) and writes it out as whole
// 256-byte row segments, 16 bytes per lane, 4 rows per instruction: 16 store instructions per wave.
// Values are computed exactly as in gemm_epilogue (fp32 bias add, activation, optional gating product, one
// rounding) -- only the path to memory differs.
#define EP_PITCH 272   // NI = 4 (128 columns per wave); NI = 2: 144
template <typename T, int ACT, int NI = 4>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs &p, f32x16 (&acc)[NI][2], uint2 (&bias_r)[NI][4],
                                                  unsigned char *stage, int e, int m0, int n0, int wm, int wn,
                                                  int lane, int row_limit) {
  constexpr int PITCH = NI * 64 + 16;  // bytes per staged row (NI*32 features + 16 B: see EP_PITCH)
  const int l31 = lane & 31, kg = lane >> 5;
  uint16_t *De = reinterpret_cast<uint16_t *>(p.D) + (size_t)e * p.d_stride_e;
  const uint16_t *Me = p.mul ? reinterpret_cast<const uint16_t *>(p.mul) + (size_t)e * p.d_stride_e : nullptr;
  const bool has_bias = p.bias != nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + l31;
    size_t roff = 0;
    if (Me) {
      const int mc = min(m, p.R - 1);
      roff = (size_t)(mc / p.d_rpw) * p.d_stride_w + (size_t)(mc % p.d_rpw) * p.ldd;
    }
    unsigned char *srow = stage + (mi * 32 + l31) * PITCH + kg * 8;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][rg * 4 + r];
        if (has_bias) {
          const uint2 bb = bias_r[ni][rg];
          uint16_t b4[4] = {(uint16_t)(bb.x & 0xffff), (uint16_t)(bb.x >> 16), (uint16_t)(bb.y & 0xffff), (uint16_t)(bb.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T tb;
            __builtin_memcpy(&tb, &b4[r], 2);
            v[r] += Elem<T>::to_f32(tb);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = activate<ACT>(v[r]);
        if (Me) {
          const int n = min(n0 + wn * (NI * 32) + ni * 32 + rg * 8 + kg * 4, p.N - 4);
          const uint2 mm = *reinterpret_cast<const uint2 *>(Me + roff + n);
          uint16_t m4[4] = {(uint16_t)(mm.x & 0xffff), (uint16_t)(mm.x >> 16), (uint16_t)(mm.y & 0xffff), (uint16_t)(mm.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T tm;
            __builtin_memcpy(&tm, &m4[r], 2);
            v[r] *= Elem<T>::to_f32(tm);
          }
        }
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T tv = Elem<T>::from_f32(v[r]);
          __builtin_memcpy(&o[r], &tv, 2);
        }
        uint2 ov;
        ov.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        ov.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2 *>(srow + (ni * 32 + rg * 8) * 2) = ov;
      }
    }
  }
  // the region is private to the wave and LDS operations of one wave complete in order: no barrier
  constexpr int LPR = NI * 4, RPI = 64 / LPR;  // lanes per row (16 bytes each), rows per store instruction
  const int c16 = lane % LPR, r4 = lane / LPR;
  const int n = n0 + wn * (NI * 32) + c16 * 8;
  if (p.d_store != 0 && p.d_peer != nullptr) {
    // peer rows (IPC transport), write-through: the rows of ONE store instruction -- RPI consecutive rows starting at a multiple of RPI --
    // belong to one source rank (the host sets d_store for peer stores only when d_rpw % 8 == 0), so the descriptor over that rank's
    // return buffer is wave-uniform.  The payload leaves for the peer while the kernel runs instead of in its end-of-kernel write-back.
    const int mw = __builtin_amdgcn_readfirstlane(m0 + wm * 64);
    const int wmax = (p.R - 1) / p.d_rpw;
#pragma unroll 4
    for (int it = 0; it < 64 / RPI; ++it) {
      const int mf = mw + it * RPI;                       // first row of this instruction (uniform)
      const int w = min(mf / p.d_rpw, wmax), l0 = mf - (mf / p.d_rpw) * p.d_rpw;
      const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(p.d_peer[w] + p.d_peer_off), 0, -1, 0x00020000);
      const int row = it * RPI + r4;
      const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + row * PITCH + c16 * 16);
      if (mw + row < row_limit && n < p.N)
        __builtin_amdgcn_raw_buffer_store_b128(val, rs_w, (int)(((size_t)e * p.d_stride_e + (size_t)(l0 + r4) * p.ldd + n) * 2), 0, 17);
    }
    peer_canary_store(p.d_peer, p.d_can);
    return;
  }
  if (p.d_store != 0) {
    // TUTEL_OPT_GEMM_STORE (round 5): the output tile leaves with write-through (sc0 sc1; the default) or non-temporal stores -- buffer
    // stores through a descriptor over the expert's output, so the cache-policy bits come from the compiler (round 4's inline-assembly
    // stores lacked the hazard wait states).  Plain stores leave the tile dirty in the XCD's L2, and what is still
    // dirty when the last wave ends is written back THEN, with nothing left to hide it behind: measured (profiles/r05_store_ab.json)
    // fc1 110.2 -> 104.5 us and fc2 106.8 -> 103.3 us inside the headline forward, the MFMA-bound pair of an 8-way rank 130.4 -> 126.8 us,
    // its pipeline stage 76.4 -> 73.0 us.  Write-through keeps the lines valid in L2 for the next kernel (decode is unchanged at 12.1 us;
    // with non-temporal stores it reads the expert outputs from HBM: 14.2 us).  Same values, same addresses.
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(De, 0, -1, 0x00020000);
#define EP_STORE_LOOP(AUX)                                                                                        \
    _Pragma("unroll 4") for (int it = 0; it < 64 / RPI; ++it) {                                                   \
      const int row = it * RPI + r4;                                                                              \
      const int m = m0 + wm * 64 + row;                                                                           \
      const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + row * PITCH + c16 * 16);                         \
      if (m < row_limit && n < p.N) {                                                                             \
        const size_t roff = (size_t)(m / p.d_rpw) * p.d_stride_w + (size_t)(m % p.d_rpw) * p.ldd;                 \
        __builtin_amdgcn_raw_buffer_store_b128(val, rs_d, (int)((roff + n) * 2), 0, AUX);                         \
      }                                                                                                           \
    }
    if (p.d_store == 1) { EP_STORE_LOOP(17); } else { EP_STORE_LOOP(2); }
#undef EP_STORE_LOOP
    return;
  }
#pragma unroll 4
  for (int it = 0; it < 64 / RPI; ++it) {
    const int row = it * RPI + r4;
    const int m = m0 + wm * 64 + row;
    const u32x4 val = *reinterpret_cast<const u32x4 *>(stage + row * PITCH + c16 * 16);
    if (m < row_limit && n < p.N) {
      const size_t roff = (size_t)(m / p.d_rpw) * p.d_stride_w + (size_t)(m % p.d_rpw) * p.ldd;
      uint16_t *drow = p.d_peer != nullptr ? gemm_out_row(p, e, m) : De + roff;
      *reinterpret_cast<u32x4 *>(drow + n) = val;
    }
  }
  if (p.d_peer != nullptr) peer_canary_store(p.d_peer, p.d_can);
}

// -------------------------------------------------------------------------------------------
// 256 x 256 tile variant for R >= 256 rows per expert (expert-parallel ranks, large batches).
// There the GEMM is no longer bound by streaming the weights once from HBM but by the bytes that
// cross L2 -> CU per flop: a 128 x 128 tile moves (128+128)*2 B per 2*128*128 flop per unit of k
// = 64 flop/B, and at the ~36 GB/s per CU that path delivers (measured with the gate-projection
// probes) that is ~590 TFLOP/s chip-wide -- what the 128 x 128 kernels reach (650).
// 256 x 256 doubles the intensity.  Same LDS-DMA structure and swizzles as above:
//   8 waves = 4 (64-row groups) x 2 (128-column groups); wave tile 64 x 128 = 2 x 4 MFMA 32x32x16
//   tiles (128 accumulator registers); LDS stage = token tile [256][64] 32 KB + two [128n][64k]
//   (or [64k][128n]) weight sub-tiles 2 x 16 KB; 2 stages = 128 KB, one block per CU.
// -------------------------------------------------------------------------------------------
#define GB_BM 256
#define GB_THREADS 512

// NI = 32-column MFMA tiles per wave: 4 -> 256 x 256 block tile (two 128-column weight sub-tiles per
// stage), 2 -> 256 x 128 (one sub-tile; for launches whose 256 x 256 grid would leave CUs idle).
// BUF: LDS-DMA through buffer descriptors (no VALU on the issue path) and the epilogue through LDS -- the two levers of
// the ping-pong kernel below that carry over to this lockstep structure (k-major weights, 32-bit addressable operands).
// BM = 128 (4 waves, 2 x 2; round 4): the 128-row HBM-bound regime on a 128 x 256 tile -- every row of an expert and TWO of its
// 128-column weight tiles per block, so the token tile crosses L2 -> LDS once per 256 columns instead of once per 128, on a
// three-slot ring (3 x 48 KB, two K-tiles = 96 KB of DMA in flight per CU, never drained), one 4-wave block per CU.
// FL (round 5, the 128 x 256 ring only): FUSED LOCATION.  On the single-rank one-call path the stable rank of every (choice, token)
// entry inside its expert -- what location_kernel (routing.hip) computes between the top-k kernel and this GEMM: 7.3 us of dependent
// L2 round trips on 64 workgroups while 192 CUs and HBM idle -- is recomputed HERE by every block for ITS expert, instead of being
// waited for: the block scans the byte copy of idx (k*T bytes, 8 KB at the headline, L2-resident after the first block of an XCD
// touched it) for its expert id, ranks the matches in (choice, token) order with one block-wide prefix sum, and keeps the first C
// of them as its token-tile rows -- `slot_map[e][0..C)` without a launch in between, under the weight DMA of the first two K-tiles.
// The n-tile-0 block of each expert also stores loc[] of its entries and its row of the slot map (decode and the API want them).
// Recomputing beats synchronising: the scan is ~100 VALU operations per thread and one load round trip that was there anyway (the
// slot-map lookup).  dispatch_count / max load / gshard loss move into an extra block of the decode launch (dispatch.hip).
// The tile as a DEVICE FUNCTION (round 6): one (expert e, M-tile mt, N-tile nt) work item on the LDS region `smem`.  Its two callers are
// the one-tile-per-workgroup kernel expert_gemm_big_kernel (expert_gemm.hip) and the persistent fc1 -> fc2 kernel (expert_ffn.hip),
// which runs it item after item -- the same instructions in the same order, so the same bits whichever launches it.
struct GemmNoHook { __device__ __forceinline__ void operator()() const {} };
// `tail`: called by every thread between the K loop and the epilogue (the persistent kernel asks for its next ticket there, so that the
// atomic's round trip hides behind the epilogue's LDS staging and stores); a tile without rows calls it too.
template <typename T, bool W_KMAJOR, int ACT, int NI, int NS, bool BUF = false, int BM = GB_BM, bool FL = false, typename Tail = GemmNoHook>
__device__ __forceinline__ void gemm_big_tile(const GemmArgs &p, const int e, const int mt, const int nt, unsigned char *smem,
                                              Tail tail = Tail()) {
  static_assert(!FL || (BUF && NS == 3 && BM == 128 && W_KMAJOR), "FL: the 128 x 256 ring kernel only");
  constexpr int NW = BM / 32;               // waves: (BM / 64) row groups x 2 column groups
  constexpr int NSUB = NI / 2;              // 128-column weight sub-tiles per stage
  constexpr int WPW = 16 * NSUB / NW;       // weight DMA pieces per wave and stage
  constexpr int BN = NI * 64;               // block tile columns
  constexpr int A_STAGE = (BM / 128) * GL_STAGE;  // elements of the [BM][64] token tile
  uint16_t *sA = reinterpret_cast<uint16_t *>(smem);  // [NS][A_STAGE]           (BM rows x 64 k)
  uint16_t *sW = sA + NS * A_STAGE;                   // [NS][NSUB][GL_STAGE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;                 // compute roles: 64-row group, (NI*32)-column group
  const int wsub = (wn * NI * 32) / GM_BN, wcol = (wn * NI * 32) % GM_BN;  // sub-tile and column offset of the wave's columns

  const int m0 = mt * BM, n0 = nt * BN;
  GEMM_PROBE(0);

  int row_limit = p.R;
  if (p.row_counts != nullptr) {
    int c = p.row_counts[e];
    c = (c + p.row_align - 1) / p.row_align * p.row_align;
    row_limit = min(row_limit, c);
  }
  if (m0 >= row_limit) {
    tail();
    return;
  }

  const uint16_t *Ae = reinterpret_cast<const uint16_t *>(p.A) + (size_t)e * p.a_stride_e;
  const uint16_t *We = reinterpret_cast<const uint16_t *>(p.W) + (size_t)e * p.w_stride_e;

  // DMA sources: token tile pieces j = wid*4 + i (rows 8j..8j+7 of 256); weight pieces g = wid*WPW + i over the
  // NSUB sub-tiles of 16 pieces each
  const uint16_t *a_src[4], *w_src[WPW];
  int gr4[4], slot4[4];
  // SG (round 5; the ring kernels with buffer-descriptor DMA): the slot-map entries of the fused fast_encode come through the SCALAR
  // cache -- the wave's 32 token-tile rows are 32 consecutive map entries, 4 x s_load_dwordx8 -- and the token-tile addresses are
  // worked out only AFTER the weight pieces of the first tiles have been issued.  As vector loads the four lookups sat in front of
  // the first DMA (s_waitcnt vmcnt(0) before any weight byte was requested: one dependent L2 round trip at the head of every block),
  // and moving the weight issue above them would not have helped: vmcnt retires in order, so waiting for the lookups would have meant
  // waiting for the weight data.  Scalar loads count on lgkmcnt.  Round 6: they are ordinary loads from the constant
  // address space at a wave-uniform address, so the COMPILER places -- and tracks -- the lgkmcnt wait in front of their first use
  // (round 5 issued them from inline assembly with the wait in a second asm statement: nothing stopped hipcc from copying or
  // spilling the destination registers in between).  A wave whose 32 entries would reach past the map takes the vector lookups.
  constexpr bool SG = BUF && NS == 3;
  typedef int sgv8u __attribute__((ext_vector_type(8), aligned(4)));
  sgv8u sg_q[4];
  const bool fl_on = FL && p.fl_idx8 != nullptr;             // block-uniform
  const int mo_ = e * p.R + m0 + 32 * wid;                   // first map entry of this wave's rows (wave-uniform)
  const bool sg_on = SG && p.a_rows != nullptr && p.sgather && !fl_on && mo_ + 32 <= p.E_loc * p.R;  // wave-uniform
#pragma unroll
  for (int i = 0; i < 4; ++i) gr4[i] = min(m0 + 8 * (wid * 4 + i) + (lane >> 3), p.R - 1);
  if (!sg_on && !fl_on) gather_rows4(p, e, gr4, slot4);
#define GB_A_ADDR()                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
    const int r = 8 * (wid * 4 + i) + (lane >> 3);                                                             \
    const int c = (lane & 7) ^ ((r >> 1) & 7);                                                                 \
    const int gr = gr4[i];                                                                                     \
    a_src[i] = Ae + (size_t)(gr / p.a_rpw) * p.a_stride_w + (size_t)(gr % p.a_rpw) * p.lda + c * 8;            \
    if (p.a_rows != nullptr) {                                                                                 \
      const int q = slot4[i];                                                                                  \
      a_src[i] = (q >= 0 ? reinterpret_cast<const uint16_t *>(p.A) + (size_t)(q % p.a_rows_mod) * p.lda        \
                         : reinterpret_cast<const uint16_t *>(p.a_zero)) + c * 8;                              \
    }                                                                                                          \
  }
  if (!sg_on && !fl_on) { GB_A_ADDR(); }
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int g = wid * WPW + i, dg = g >> 4, j = g & 15;
    if (W_KMAJOR) {
      const int r = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int gn = min(n0 + dg * GM_BN + r, p.N - 1);
      w_src[i] = We + (size_t)gn * p.ldw + c * 8;
    } else {
      const int kr = 4 * j + (lane >> 4);
      const int cn = (lane & 15) ^ ((kr & 3) << 2);
      const int gn = min(n0 + dg * GM_BN + cn * 8, p.N - 8);
      w_src[i] = We + (size_t)kr * p.ldw + gn;
    }
  }
  const size_t w_step = W_KMAJOR ? (size_t)GL_BK : (size_t)GL_BK * p.ldw;
  const int piece_a = wid * 4 * 512, piece_w = wid * WPW * 512;  // weight pieces are consecutive across the sub-tiles
  int a_off[4], w_off[WPW];
  const uint16_t *abase = p.a_rows != nullptr ? reinterpret_cast<const uint16_t *>(p.A) : Ae;
#define GB_A_OFF()                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
    const int r = 8 * (wid * 4 + i) + (lane >> 3);                                                             \
    a_off[i] = (int)(unsigned)((const char *)a_src[i] - (const char *)abase);                                  \
    if ((p.a_rows != nullptr && slot4[i] < 0) || m0 + r >= row_limit)                                          \
      a_off[i] = (int)0x7ffff000u + (((lane & 7) ^ ((r >> 1) & 7)) << 4);  /* empty slot / past the row count: out of range -> zeros */ \
  }
  if (BUF) {
    if (!sg_on && !fl_on) { GB_A_OFF(); }
#pragma unroll
    for (int i = 0; i < WPW; ++i) w_off[i] = (int)(unsigned)((const char *)w_src[i] - (const char *)We);
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(p.a_rows != nullptr ? reinterpret_cast<const uint16_t *>(p.A) : Ae), 0, p.a_span_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(We), 0, -1, 0x00020000);
  // one M-tile per expert: every weight byte is fetched by exactly one block -> no-allocate loads pay on the
  // 256 x 256 tile with the chip covered (32 x 256 rows: 720 -> 758 TFLOP/s, dropless 64 x 157: 171 -> 156 us).
  // With several M-tiles the blocks re-read each other's weight tiles from L2 and the hint costs 6-15 %; it also
  // costs on the 256 x 128 tile and on half-empty grids (measured), so it is limited to the case that gains.
  const bool w_once = NI == 4 && p.ntm == 1 && gridDim.x >= 256;

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, kg = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  int frag_k[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_k[kk] = (((kk * 2 + kg) ^ sw) << 3);
  const int a_row = (wm * 64 + l31) * GL_BK;  // + mi*32*64
  const int wk_row = (wcol + l31) * GL_BK;    // + ni*32*64, inside sub-tile wsub
  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2;
  const int c_lo = wcol / 8 + (g16 & 1) * 2 + ((i16 & 3) >> 1);
  const int wt_row = ((g16 >> 1) * 8 + q4) * GM_BN;
  int wt_c[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) wt_c[ni] = (((c_lo + 4 * ni) ^ (q4 << 2)) << 3) + (i16 & 1) * 4;

  const int nk = p.K / GL_BK;
  // the same k order as the 128-tile kernels (rotation per PAIR of 128-column tiles)
  const int npair = NI == 4 ? p.ntn : (p.ntn + 1) >> 1, pair = NI == 4 ? nt : nt >> 1;
  const int rot = p.rot_on ? (int)(((long long)(pair + 3 * e) * nk / npair) % nk) : 0;

#define GB_ISSUE_A(KT, STG)                                                            \
  do {                                                                                 \
    int kr_ = (KT) + rot; kr_ = kr_ >= nk ? kr_ - nk : kr_;                            \
    uint16_t *da_ = sA + (STG) * A_STAGE + piece_a;                                    \
    if (BUF) {                                                                         \
      const int ao_ = kr_ * (GL_BK * 2);                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) bdma16<false>(rs_a, a_off[i_], ao_, da_ + i_ * 512); \
    } else {                                                                           \
      const size_t ao_ = (size_t)kr_ * GL_BK;                                          \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) glds16(a_src[i_] + ao_, da_ + i_ * 512, false); \
    }                                                                                  \
  } while (0)
#define GB_ISSUE_W(KT, STG)                                                            \
  do {                                                                                 \
    int kr_ = (KT) + rot; kr_ = kr_ >= nk ? kr_ - nk : kr_;                            \
    uint16_t *dw_ = sW + (STG) * NSUB * GL_STAGE + piece_w;                            \
    if (BUF) {                                                                         \
      const int wo_ = (int)(kr_ * (w_step * 2));                                       \
      if (w_once) { _Pragma("unroll") for (int i_ = 0; i_ < WPW; ++i_) bdma16<true>(rs_w, w_off[i_], wo_, dw_ + i_ * 512); } \
      else { _Pragma("unroll") for (int i_ = 0; i_ < WPW; ++i_) bdma16<false>(rs_w, w_off[i_], wo_, dw_ + i_ * 512); } \
    } else {                                                                           \
      const size_t wo_ = (size_t)kr_ * w_step;                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < WPW; ++i_) glds16(w_src[i_] + wo_, dw_ + i_ * 512, w_once); \
    }                                                                                  \
  } while (0)
#define GB_ISSUE(KT, STG) do { GB_ISSUE_A(KT, STG); GB_ISSUE_W(KT, STG); } while (0)
#define GB_LOAD_FRAGS(FA, FW, KK)                                                      \
  do {                                                                                 \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                   \
      FA[mi] = *reinterpret_cast<const u32x4 *>(ca + a_row + mi * 32 * GL_BK + frag_k[KK]); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                \
      if (W_KMAJOR) {                                                                  \
        FW[ni] = *reinterpret_cast<const u32x4 *>(cw + wk_row + ni * 32 * GL_BK + frag_k[KK]); \
      } else {                                                                         \
        const uint16_t *ptr = cw + wt_row + (KK) * 16 * GM_BN + wt_c[ni];              \
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
            (__attribute__((address_space(3))) s16x4_t *)(ptr));                       \
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
            (__attribute__((address_space(3))) s16x4_t *)(ptr + 4 * GM_BN));           \
        u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi); \
        u32x4 f = {lo2[0], lo2[1], hi2[0], hi2[1]};                                    \
        FW[ni] = f;                                                                    \
      }                                                                                \
    }                                                                                  \
  } while (0)
#define GB_MMA(FA, FW)                                                                 \
  do {                                                                                 \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                  \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                 \
        acc[ni][mi] = Mma<T>::run(FW[ni], FA[mi], acc[ni][mi]);                        \
  } while (0)

#define GB_TILE(BUF)                                                                   \
  do {                                                                                 \
    const uint16_t *ca = sA + (BUF) * A_STAGE, *cw = sW + ((BUF) * NSUB + wsub) * GL_STAGE; \
    /* two half-tiles: fragments of two k-steps, then their MFMAs; the partner wave on the SIMD runs its MFMAs \
       while this one waits for LDS */                                                 \
    u32x4 fa[2][2], fw[2][NI];                                                         \
    GB_LOAD_FRAGS(fa[0], fw[0], 0);                                                    \
    GB_LOAD_FRAGS(fa[1], fw[1], 1);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    GB_MMA(fa[0], fw[0]);                                                              \
    GB_MMA(fa[1], fw[1]);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    GB_LOAD_FRAGS(fa[0], fw[0], 2);                                                    \
    GB_LOAD_FRAGS(fa[1], fw[1], 3);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    GB_MMA(fa[0], fw[0]);                                                              \
    GB_MMA(fa[1], fw[1]);                                                              \
  } while (0)

  if (NS == 2) {
    GB_ISSUE(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) GB_ISSUE(kt + 1, buf ^ 1);
      GB_TILE(buf);
      __syncthreads();  // all waves done with stage `buf`; next tile's DMA has landed (vmcnt(0))
    }
  } else {
    // NS-slot ring (the 256 x 128 tile leaves room for three 48 KB slots): NS - 1 tiles in flight while one is
    // consumed.  A __syncthreads would drain every DMA, so the wait is a hand-counted vmcnt (each tile = 4 + WPW
    // DMA ops per wave, landing in order) and a bare s_barrier, which also says every wave is done reading the
    // slot the next issue overwrites.
    constexpr int OPS = 4 + WPW;
    // prologue: the WEIGHT pieces of the first NS - 1 tiles go out before the token pieces -- with the fused fast_encode the token
    // addresses come from a slot-map lookup (one more dependent L2 round trip per block), the weight addresses do not, so the HBM
    // stream starts without waiting for it.  In flight, in issue order: [W(0) .. W(NS-2)] [A(0) .. A(NS-2)]; the first K-tile
    // needs everything up to A(0), i.e. all but the 4 * (NS - 2) token ops after it (NS = 3: equal to the steady-state count from
    // the second tile on, where each iteration issues [A, W] of one tile).
    static_assert(NS <= 3, "the prologue order below is worked out for rings of at most three slots");
    // FL: the byte copy of idx goes out FIRST, as LDS-DMA into the 16 KB past the ring -- vmcnt retires in order, so being older than
    // the weight DMA is what lets `s_waitcnt vmcnt(16)` below wait for it WITHOUT waiting for the weights (16 = the 2 x WPW weight
    // ops every wave issues after it; as register loads hipcc placed a vmcnt(0) at their first use).  1 KB per wave instruction
    // through a descriptor whose range is the buffer: the tail reads zeros.
    constexpr int FL_CH = 4;  // 16-byte chunks per thread: k*T <= 15 KB of entries
    unsigned char *s_fl = smem + (size_t)NS * (A_STAGE + NSUB * GL_STAGE) * 2;  // [15360] idx bytes, [128] slots, [4] wave totals
    if (FL && fl_on) {
      static_assert(!FL || WPW * (NS - 1) == 16, "the vmcnt(16) of the fused-location prologue counts the weight ops of the first two tiles");
      const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.fl_idx8), 0, (p.fl_n + 15) & ~15, 0x00020000);
      // wave w fetches exactly the bytes ITS threads scan (thread t owns entries [t * per, (t + 1) * per), per a multiple of 16: a
      // wave's share is per / 16 KB): its own `s_waitcnt vmcnt(16)` is then all the ordering the reads below need -- no barrier
      const int per_ = (((p.fl_n + 255) >> 8) + 15) & ~15, ipw = per_ >> 4;
      for (int i = wid * ipw; i < (wid + 1) * ipw; ++i)
        if (i < 15) bdma16<false>(rs_i, i * 1024 + lane * 16, 0, reinterpret_cast<uint16_t *>(s_fl + i * 1024));
    }
    // (the scalar loads go out here, not at the top of the kernel: hipcc fetches kernel arguments lazily and every
    // `s_waitcnt lgkmcnt(0)` it places for them would wait for these too)
    if (sg_on) {
      // everything the weight issue below consumes is forced into registers FIRST (empty volatile asm statements keep their order):
      // SMEM returns out of order, so the one wait hipcc can place for a late kernel-argument fetch is lgkmcnt(0) -- which would
      // also wait for the map entries
      asm volatile("" ::"s"(reinterpret_cast<uintptr_t>(We)), "s"(rot), "s"(nk), "s"((int)w_once), "s"((int)(w_step * 2)));
#pragma unroll
      for (int i = 0; i < WPW; ++i) asm volatile("" ::"v"(w_off[i]));
      const __attribute__((address_space(4))) sgv8u *mp =
          reinterpret_cast<const __attribute__((address_space(4))) sgv8u *>(reinterpret_cast<uintptr_t>(p.a_rows) + (size_t)mo_ * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) sg_q[i] = mp[i];
    }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < nk) GB_ISSUE_W(t, t);
    if (sg_on) {  // the weight stream is on its way: now the slot-map entries (first use: hipcc waits here) and the token addresses
      const int rl_ = lane >> 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q = sg_q[i][0];
#pragma unroll
        for (int j = 1; j < 8; ++j) q = (rl_ == j) ? sg_q[i][j] : q;
        slot4[i] = q;
      }
      GB_A_ADDR();
      GB_A_OFF();
    }
    if (FL && fl_on) {
      int *s_slot = reinterpret_cast<int *>(s_fl + 15360);
      int *s_wt = s_slot + 128;
      if (tid < 128) s_slot[tid] = -1;
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // this wave's share of the idx bytes is in LDS; the weight pieces are still in flight
      // The scan sits on the block's critical path (the token tile cannot be requested before it ends), so it is kept to a few
      // hundred cycles: dead chunks skipped (block-uniform), the tail mask only in the one thread that crosses k*T, the wave prefix
      // from 7 ballots over the bits of the per-thread count (<= 64) instead of 6 ds_bpermute round trips, and only threads that
      // own a match enter the assignment loop.
      const int per = (((p.fl_n + 255) >> 8) + 15) & ~15;  // consecutive entries per thread, a multiple of 16 (<= 64)
      const int nch = per >> 4;                            // block-uniform
      const int fl_e0 = tid * per;
      const uint32_t eb = (uint32_t)e * 0x01010101u;
      uint32_t fl_t[FL_CH][4];                             // bit 7 of every byte that equals this block's expert id
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < FL_CH; ++c) {
        if (c < nch) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(s_fl + min(fl_e0 + c * 16, 15360 - 16));
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t x = v[d] ^ eb;
            fl_t[c][d] = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
          }
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d) fl_t[c][d] = 0u;
        }
      }
      if (fl_e0 + per > p.fl_n) {  // entries past k*T (the last threads only)
#pragma unroll
        for (int c = 0; c < FL_CH; ++c)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int nv = p.fl_n - (fl_e0 + c * 16 + d * 4);
            fl_t[c][d] = nv >= 4 ? fl_t[c][d] : (nv <= 0 ? 0u : (fl_t[c][d] & ((1u << (8 * nv)) - 1u)));
          }
      }
#pragma unroll
      for (int c = 0; c < FL_CH; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) cnt += __popc(fl_t[c][d]);
      int below = 0;  // matches in the lower lanes of the wave
#pragma unroll
      for (int bit = 0; bit < 7; ++bit) {
        const unsigned long long bb = __ballot((cnt >> bit) & 1);
        below += __popcll(bb & ((1ull << lane) - 1ull)) << bit;
      }
      const int wtot = __shfl(below + cnt, 63, 64);
      if (lane == 0) s_wt[wid] = wtot;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int r = below;
      for (int w2 = 0; w2 < wid; ++w2) r += s_wt[w2];
      const bool first_tile = nt == 0;
      if (cnt != 0) {
#pragma unroll
        for (int c = 0; c < FL_CH; ++c)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            uint32_t t = fl_t[c][d];
            while (t) {
              const int q = fl_e0 + c * 16 + d * 4 + ((__ffs(t) - 1) >> 3);
              if (first_tile) p.fl_loc[q] = r;
              if (r < p.R) s_slot[r] = q;
              ++r;
              t &= t - 1;
            }
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) slot4[i] = s_slot[8 * (wid * 4 + i) + (lane >> 3)];
      if (first_tile && tid < p.R) const_cast<int32_t *>(p.a_rows)[(size_t)e * p.R + tid] = s_slot[tid];
      GB_A_ADDR();
      GB_A_OFF();
    }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < nk) GB_ISSUE_A(t, t);
    int kt = 0, slot = 0;
    GEMM_PROBE(1);  // prologue issued
    for (; kt + NS - 1 < nk; ++kt) {
      if (kt == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS * (NS - 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      if (kt == 0) { GEMM_PROBE(2); }  // first K-tile landed
      const int nslot = slot == 0 ? NS - 1 : slot - 1;  // (kt + NS - 1) % NS
      GB_ISSUE(kt + NS - 1, nslot);
      GB_TILE(slot);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    GEMM_PROBE(3);  // last DMA issued: the ring drains
    for (; kt < nk; ++kt) {  // drain
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      GB_TILE(slot);
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
    GEMM_PROBE(4);  // K loop done
  }
#undef GB_TILE
#undef GB_A_ADDR
#undef GB_A_OFF
#undef GB_ISSUE
#undef GB_ISSUE_A
#undef GB_ISSUE_W
#undef GB_LOAD_FRAGS
#undef GB_MMA

  tail();
  GM_PRELOAD_BIAS_N(NI);
  if (BUF && !((p.ldd & 7) || (p.d_stride_e & 7) || (p.d_stride_w & 7) || (reinterpret_cast<uintptr_t>(p.D) & 15))) {
    __syncthreads();  // every wave is done with the K-tile stages: LDS is free for the staging regions
    unsigned char *stage = smem + wid * (64 * (NI * 64 + 16));
    if (ACT == GEMM_ACT_RUNTIME) {  // one copy of everything above for all activations; the same epilogue code (and bits) per activation
      if (p.act_rt == TUTEL_ACT_NONE) gemm_epilogue_lds<T, TUTEL_ACT_NONE, NI>(p, acc, bias_r, stage, e, m0, n0, wm, wn, lane, row_limit);
      else if (p.act_rt == TUTEL_ACT_RELU) gemm_epilogue_lds<T, TUTEL_ACT_RELU, NI>(p, acc, bias_r, stage, e, m0, n0, wm, wn, lane, row_limit);
      else if (p.act_rt == TUTEL_ACT_GELU) gemm_epilogue_lds<T, TUTEL_ACT_GELU, NI>(p, acc, bias_r, stage, e, m0, n0, wm, wn, lane, row_limit);
      else gemm_epilogue_lds<T, TUTEL_ACT_SILU, NI>(p, acc, bias_r, stage, e, m0, n0, wm, wn, lane, row_limit);
      return;
    }
    gemm_epilogue_lds<T, ACT == GEMM_ACT_RUNTIME ? TUTEL_ACT_NONE : ACT, NI>(p, acc, bias_r, stage, e, m0, n0, wm, wn, lane, row_limit);
    return;
  }
  static_assert(ACT != GEMM_ACT_RUNTIME || BUF, "the run-time activation switch lives in the LDS epilogue");
  gemm_epilogue<T, ACT == GEMM_ACT_RUNTIME ? TUTEL_ACT_NONE : ACT, NI>(p, acc, bias_r, e, m0, n0, wm, wn, l31, kg, row_limit);
}

