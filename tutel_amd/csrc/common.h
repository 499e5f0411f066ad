// common.h -- shared device helpers for libtutel_amd.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tutel_amd.h"

#define WAVE 64

// ---- error reporting ---------------------------------------------------------------------
void tutel_set_error(const char *fmt, ...);
int tutel_get_option(int key);  // TUTEL_OPT_*: -1 automatic, 0 / 1 forced (api.hip)

// ---- per-stage timing (api.hip): HIP events around the launches of one C-ABI call, when enabled -------------
int tutel_stage_begin(int stage, hipStream_t st);  // returns a token (-1: timing off)
void tutel_stage_end(int token, hipStream_t st);
void tutel_stage_hint(int stage);                  // the next launches of this thread belong to `stage` (-1: clear)
void tutel_gemm_corun_hint(int on);                // the next GEMM launches of this thread run beside a collective on another stream
int tutel_gemm_corun();
void tutel_stage_range_push(int stage);            // roctx range named after the stage (no-op without libroctx64)
void tutel_stage_range_pop();
struct StageScope {
  int token;
  hipStream_t st;
  StageScope(int stage, hipStream_t s) : token(tutel_stage_begin(stage, s)), st(s) { tutel_stage_range_push(stage); }
  ~StageScope() {
    tutel_stage_end(token, st);
    tutel_stage_range_pop();
  }
};

// ---- internal entry points shared between translation units (not part of the C ABI) -----------------------------------
// peer stores of the IPC transport (ep.hip): tab[w] = base address, in this process, of rank w's exchange segment
// In-band epoch canaries (round 5): EP_NCAN 32-bit words per (direction, stage, source rank) at the end of every exchange
// segment.  The LAST blocks of a producing kernel store the epoch the following signal kernel will publish into the canary words
// of every destination rank -- after their row stores, through the same store path -- and the wait kernel, once the flag has
// arrived, reads them back with system-scope loads: a flag that overtook the rows of its own kernel (the failure the peer-store
// protocol must never show, and the one a one-GPU lease cannot provoke) leaves canaries of the previous epoch behind and is
// reported instead of being consumed silently.
// What a matching canary does NOT prove: the words are written by the LAST min(grid, EP_NCAN) blocks after a block-local
// barrier -- their own row stores have been ISSUED in front of them on the same path, the stores of the other blocks (dispatched
// earlier, normally finished earlier) are not waited for.  A stale canary is therefore proof that a flag overtook rows; a fresh
// one is strong evidence, not proof, that it did not.  It is a detector that turns the one failure this protocol must never show
// into an error on the NEXT call -- not a fence; the ordering itself comes from the signal kernel running after the producer on
// the same stream.  TUTEL_OPT_EP_CANARY must be set alike on every rank (a rank that checks words its peer never writes misfires).
#define EP_NCAN 16
struct PeerCanary {
  const uint32_t *epoch;  // device: epochs signalled so far for this (direction, stage); the producer publishes *epoch + 1 (nullptr: off)
  long long off;          // byte offset, inside every rank's segment, of THIS producer's EP_NCAN words for that (direction, stage)
  int world;              // destination ranks
  int stale;              // test injection (TUTEL_OPT_EP_INJECT): publish the OLD epoch, as if the stores had not landed
};
struct EncodePeer {
  const uint64_t *tab;  // device array of `world` segment bases (nullptr: plain local output)
  long long off;        // byte offset of the receive array inside every rank's segment
  int rank, rows;       // my rank; bucket rows per (stage, destination rank) block
  int slot0;            // first bucket row of this launch (stages are encoded by separate launches)
  PeerCanary can;
};
int tutel_encode_launch(const void *x, int dtype, const int32_t *slot_map, const void *gates, int gate_dtype, int T, int M,
                        int n_slots, int capacity, int num_experts, int chunk_rows, int expert_slice, int ep_world, void *out,
                        const EncodePeer &peer, hipStream_t st);
// tutel_amd_expert_gemm with the output rows of source rank w written to d_peer[w] + d_peer_off (bytes) instead of
// D + w * d_stride_w: the second expert GEMM stores straight into the return buffers of the ranks the rows came from
int tutel_expert_gemm_peer(const void *A, int64_t a_stride_e, int64_t a_stride_w, int a_rows_per_w, int lda, const void *W,
                           int w_kmajor, int64_t w_stride_e, int ldw, const void *bias, int64_t bias_stride_e,
                           const uint64_t *d_peer, int64_t d_peer_off, int64_t d_stride_e, int d_rows_per_w, int ldd, int E_loc,
                           int R, int N, int K, int dtype, int act, const PeerCanary &can, hipStream_t st);

// expert_gemm.hip: the gather GEMM with the locations computed inside the launch (fused location); loc == NULL: eligibility query
int tutel_expert_gemm_gather_fl(const void *X, int ldx, int32_t *slot_map, int T, const void *zero_row, const void *W, int64_t w_stride_e,
                                int ldw, const void *bias, int64_t bias_stride_e, void *D, int64_t d_stride_e, int ldd, int E_loc, int R,
                                int N, int K, int dtype, int act, const uint8_t *idx8, int n, int32_t *loc, hipStream_t st);
// expert_ffn.hip: fc1 -> activation -> fc2 in one persistent launch.  idx8 != NULL: fused location (loc out).  query != 0: answers only.
// TUTEL_AMD_ENOTSUP: this shape / layout takes the two-launch path (nothing was launched)
int tutel_expert_ffn(const void *X, int64_t x_stride_e, int ldx, const int32_t *slot_map, int T, const void *zero_row, const void *W1,
                     int64_t w1_stride_e, int ldw1, const void *b1, int64_t b1_stride_e, void *hid, int64_t hid_stride_e, int ldh,
                     const void *W2, int64_t w2_stride_e, int ldw2, const void *b2, int64_t b2_stride_e, void *D, int64_t d_stride_e, int ldd,
                     int E_loc, int R, int M, int H, int M_out, int dtype, int act, const uint8_t *idx8, int n, int32_t *loc, int query,
                     hipStream_t st);
// dispatch.hip: fast_decode + the routing finish (routing_dev.h) in one launch
struct RouteFinish;
void tutel_route_finish_args(int T, int E, int k, void *ws, RouteFinish *out);  // routing.hip
int tutel_decode_finish_launch(const void *buf, int dtype, const int32_t *idx, const int32_t *loc, const void *gates, int gate_dtype, int T, int M,
                               int k, int capacity, int num_experts, void *out, const RouteFinish &fin, hipStream_t st);
// routing.hip: top-k on logits (`in`) or on the gate projection's split-K partial sums, optionally leaving a byte copy of idx
// (idx8 [k * T], E <= 128) for the fused-location expert GEMM
int tutel_gate_topk_launch(const void *in, const float *partials, int splits, int dtype, int T, int E, int k, int normalize_gate,
                           void *logits_out, int32_t *idx, void *gates, void *ws, int32_t *clear_map, int clear_n, uint8_t *idx8,
                           hipStream_t st);

#define TUTEL_REQUIRE(cond, ...)           \
  do {                                     \
    if (!(cond)) {                         \
      tutel_set_error(__VA_ARGS__);        \
      return -1;                           \
    }                                      \
  } while (0)

#define TUTEL_CHECK_LAUNCH(what)                                                    \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) {                                                         \
      tutel_set_error("%s: launch failed: %s", what, hipGetErrorString(_e));        \
      return (int)_e;                                                               \
    }                                                                               \
  } while (0)

// ---- element types -----------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

// fp32 -> bf16 round-to-nearest-even, NaN -> 0x7FC0 (c10::BFloat16 round_to_nearest_even).
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}

template <typename T> struct Elem;
// fp64: routing only (scores / gates of an fp64 gate); compute type double
template <> struct Elem<double> {
  using ct = double;
  static constexpr int dtype = 3;
  __device__ static __forceinline__ double to_f32(double x) { return x; }
  __device__ static __forceinline__ double from_f32(double x) { return x; }
  __device__ static __forceinline__ double eps() { return 2.220446049250313e-16; }
};
template <> struct Elem<float> {
  using ct = float;
  static constexpr int dtype = TUTEL_F32;
  __device__ static __forceinline__ float to_f32(float x) { return x; }
  __device__ static __forceinline__ float from_f32(float x) { return x; }
  __device__ static __forceinline__ float eps() { return 1.1920928955078125e-07f; }
};
template <> struct Elem<bf16_t> {
  using ct = float;
  static constexpr int dtype = TUTEL_BF16;
  __device__ static __forceinline__ float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
  __device__ static __forceinline__ bf16_t from_f32(float x) { return bf16_t{f32_to_bf16_bits(x)}; }
  __device__ static __forceinline__ float eps() { return 0.0078125f; }
};
template <> struct Elem<f16_t> {
  using ct = float;
  static constexpr int dtype = TUTEL_F16;
  __device__ static __forceinline__ float to_f32(f16_t x) { return (float)x.v; }
  __device__ static __forceinline__ f16_t from_f32(float x) { return f16_t{(_Float16)x}; }
  __device__ static __forceinline__ float eps() { return 0.0009765625f; }
};

// round x to T and back (what a `T`-typed torch op does to an fp32 intermediate)
template <typename T> __device__ __forceinline__ typename Elem<T>::ct round_to(typename Elem<T>::ct x) {
  return Elem<T>::to_f32(Elem<T>::from_f32(x));
}
__device__ __forceinline__ float ct_exp(float x) { return expf(x); }
__device__ __forceinline__ double ct_exp(double x) { return exp(x); }
__device__ __forceinline__ float ct_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double ct_max(double a, double b) { return fmax(a, b); }

static inline int dtype_size(int dtype) { return dtype == TUTEL_F32 ? 4 : 2; }
static inline bool dtype_ok(int dtype) {
  return dtype == TUTEL_F32 || dtype == TUTEL_F16 || dtype == TUTEL_BF16;
}

// 16-byte vector of raw bits
struct __attribute__((aligned(16))) vec16 { uint32_t w[4]; };

// unpack / pack 16 bytes <-> fp32 lanes
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void unpack(const vec16 &v, float *f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
  }
  __device__ static __forceinline__ void pack(const float *f, vec16 &v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(const vec16 &v, float *f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v.w[i] << 16);
      f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void pack(const float *f, vec16 &v) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v.w[i] = (uint32_t)f32_to_bf16_bits(f[2 * i]) | ((uint32_t)f32_to_bf16_bits(f[2 * i + 1]) << 16);
  }
};
template <> struct Vec<f16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(const vec16 &v, float *f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c;
      c.u = v.w[i];
      f[2 * i] = (float)c.h[0];
      f[2 * i + 1] = (float)c.h[1];
    }
  }
  __device__ static __forceinline__ void pack(const float *f, vec16 &v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c;
      c.h[0] = (_Float16)f[2 * i];
      c.h[1] = (_Float16)f[2 * i + 1];
      v.w[i] = c.u;
    }
  }
};

// exact (non-contracted) fp32 multiply / add: the reference's CPU loops and its k-temps-then-sum
// structure round every product and every sum separately (fast_dispatch.py:61-66).
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

// ---- wave helpers ------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- epoch canaries of the IPC transport (see PeerCanary) --------------------------------------------------------------
// Called by EVERY thread of EVERY block (1-D grid) as the last thing a producing kernel does.  The last min(grid, EP_NCAN)
// blocks -- dispatched last, finishing last -- publish: after a block-wide barrier (every wave of the block has issued its
// row stores), thread (w, j) stores the epoch into word j of rank w's canary words when j is this block's share.
__device__ __forceinline__ void peer_canary_store(const uint64_t *tab, const PeerCanary &c) {
  if (c.epoch == nullptr) return;
  const int nb = (int)gridDim.x, nlast = nb < EP_NCAN ? nb : EP_NCAN, back = nb - 1 - (int)blockIdx.x;
  if (back >= nlast) return;  // block-uniform
  __syncthreads();
  const int t = (int)threadIdx.x;
  if (t >= c.world * EP_NCAN) return;
  const int w = t / EP_NCAN, j = t % EP_NCAN;
  if (j % nlast != back) return;
  const uint32_t e = *c.epoch + (c.stale ? 0u : 1u);
  reinterpret_cast<uint32_t *>(tab[w] + c.off)[j] = e;
}
