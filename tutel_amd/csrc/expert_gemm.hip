// expert_gemm.hip -- per-expert FFN grouped GEMM on MFMA for gfx950.
//
//   D[e, r, :] = act( A[e, r, :] @ op(W[e]) + bias[e, :] ),  bf16/fp16 in, fp32 accumulate.
//
// Replaces torch.matmul (+bias add +activation) of FusedExpertsNetwork.forward
// (tutel/experts/ffn.py:114-120) and torch.ops.tutel_ops.sparse_bmm_infer
// (custom_kernel.cpp:874-889) -- one launch for all experts, device-side row counts.
//
// Regime: at the headline shape each expert has only C = 128 rows, so every weight
// byte is used for 128 MACs per column -> 1.07 GB of weights vs 137 GFLOP: the launch is bound by
// streaming W from HBM once, not by MFMA.  Hence:
//   * block tile 128 rows x 128 features: ALL rows of an expert in one M-tile, so W is read from
//     HBM exactly once; the activation tile is re-read by the N-tiles of the same expert, which
//     the XCD-aware block order keeps on one XCD (its private 4 MiB L2);
//   * 4 waves (2x2), each 64x64 = 2x2 v_mfma_f32_32x32x16 accumulators (64 acc VGPRs);
//   * operands are swapped (weights = MFMA "A", activations = MFMA "B") so each lane ends up
//     with 4 consecutive output features of one row -> 8-byte epilogue stores, bias is a 4-vector;
//   * global -> registers -> LDS double buffer; next K-tile's global loads are issued before the
//     MFMAs of the current one; one __syncthreads per K-tile; 2 blocks per CU;
//   * LDS rows padded (+16 B for [.,k]-major tiles, +64 B for the [k][n] weight tile) so the
//     ds_read_b128 fragment reads and the ds_read_b64_tr_b16 transposing reads are conflict-free;
//   * [K,N]-major weights (batched_fc2_w) are consumed as stored: the k-contiguous fragment the
//     MFMA wants is produced by gfx950's transposing LDS read, no transposed weight copy in HBM;
//   * row addressing folds the expert-parallel [W,E_loc,C,M] <-> [E_loc,W*C,M] permutes
//     (communicate.py:606-622) into the loads/stores.
#include <stdlib.h>

#include "common.h"

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

template <int ACT> __device__ __forceinline__ float activate(float v) {
  if (ACT == TUTEL_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == TUTEL_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  if (ACT == TUTEL_ACT_SILU) return v / (1.f + expf(-v));
  return v;
}

struct GemmArgs {
  const void *A; long long a_stride_e, a_stride_w; int a_rpw, lda;
  const void *W; long long w_stride_e; int ldw;
  const void *bias; long long bias_stride_e;
  void *D; long long d_stride_e, d_stride_w; int d_rpw, ldd;
  int E_loc, R, N, K;
  const int32_t *row_counts; int row_align;
  int ntm, ntn;
};

// streamed-once weight loads may bypass cache allocation (each W byte is read by exactly one CU)
template <bool NT> __device__ __forceinline__ u32x4 ld16(const uint16_t *p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
  return *reinterpret_cast<const u32x4 *>(p);
}

// Tile configuration: BK = K-tile depth (64 | 128); NBUF = LDS buffers (2: one barrier per K-tile;
// 1: two barriers, half the LDS -> more blocks per CU); OCC = blocks per CU the register
// allocator is asked to allow; NT = non-temporal weight loads.
template <typename T, bool W_KMAJOR, int ACT, int BK, int NBUF, int OCC, bool NT, bool ROT>
__global__ __launch_bounds__(GM_THREADS, OCC) void expert_gemm_kernel(GemmArgs p) {
  constexpr int LDK = BK + 8;                                       // padded [rows][k] LDS row (elements)
  constexpr int A_TILE = GM_BM * LDK;                               // elements
  constexpr int W_TILE = W_KMAJOR ? GM_BN * LDK : BK * GM_LDN;      // elements
  constexpr int CPR = BK / 8;                                       // 16-byte chunks per [rows][k] row
  constexpr int RPP = GM_THREADS / CPR;                             // rows per load pass
  constexpr int NLA = GM_BM / RPP;                                  // A loads per thread per K-tile
  constexpr int NLW = W_KMAJOR ? GM_BN / RPP : BK / 16;             // W loads per thread per K-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *sA = reinterpret_cast<uint16_t *>(smem);
  uint16_t *sW = sA + NBUF * A_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- XCD-aware work order: consecutive work items (same expert, neighbouring tiles) go to
  // the same XCD (hardware places block b on XCD b % 8; speed only, never correctness).
  const int nb = gridDim.x;
  int w;
  {
    const int b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, pos = b >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int mt = w % p.ntm;
  const int nt = (w / p.ntm) % p.ntn;
  const int e = w / (p.ntm * p.ntn);
  const int m0 = mt * GM_BM, n0 = nt * GM_BN;

  int row_limit = p.R;
  if (p.row_counts != nullptr) {
    int c = p.row_counts[e];
    c = (c + p.row_align - 1) / p.row_align * p.row_align;
    row_limit = min(row_limit, c);
  }
  if (m0 >= row_limit) return;

  const uint16_t *Ae = reinterpret_cast<const uint16_t *>(p.A) + (size_t)e * p.a_stride_e;
  const uint16_t *We = reinterpret_cast<const uint16_t *>(p.W) + (size_t)e * p.w_stride_e;

  // ---- per-thread global source pointers (advance by BK along k each tile)
  const uint16_t *a_src[NLA];
  const uint16_t *w_src[NLW];
  int a_dst[NLA], w_dst[NLW];  // LDS element offsets inside a tile
  {
    const int kc = tid % CPR, rbase = tid / CPR;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      int r = rbase + RPP * i;
      int gr = min(m0 + r, p.R - 1);
      a_src[i] = Ae + (size_t)(gr / p.a_rpw) * p.a_stride_w + (size_t)(gr % p.a_rpw) * p.lda + kc * 8;
      a_dst[i] = r * LDK + kc * 8;
    }
    if (W_KMAJOR) {
#pragma unroll
      for (int i = 0; i < NLW; ++i) {
        int r = rbase + RPP * i;
        int gn = min(n0 + r, p.N - 1);
        w_src[i] = We + (size_t)gn * p.ldw + kc * 8;
        w_dst[i] = r * LDK + kc * 8;
      }
    } else {
      const int nc = tid & 15, kbase = tid >> 4;  // [k][n] tile: 16 x 16B chunks per row
      int gn = min(n0 + nc * 8, p.N - 8);
#pragma unroll
      for (int i = 0; i < NLW; ++i) {
        int kr = kbase + 16 * i;
        w_src[i] = We + (size_t)kr * p.ldw + gn;
        w_dst[i] = kr * GM_LDN + nc * 8;
      }
    }
  }
  const size_t w_step = W_KMAJOR ? (size_t)BK : (size_t)BK * p.ldw;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (elements), constant over the K loop
  const int l31 = lane & 31, kg = lane >> 5;
  const int a_frag_off = (wm * 64 + l31) * LDK + kg * 8;                // + mi*32*LDK + kk*16
  const int wk_frag_off = (wn * 64 + l31) * LDK + kg * 8;               // k-major W
  // n-major W via ds_read_b64_tr_b16: 16-lane group g reads the 4(k) x 16(n) block at
  // rows kk*16 + (g>>1)*8 + h*4, cols wn*64 + ni*32 + (g&1)*16; lane i of the group supplies the
  // address of row (i>>2), cols 4*(i&3)..+3 and receives column i, rows 0..3.
  const int g16 = lane >> 4, i16 = lane & 15;
  const int wt_frag_off = ((g16 >> 1) * 8 + (i16 >> 2)) * GM_LDN + wn * 64 + (g16 & 1) * 16 + 4 * (i16 & 3);

  const int nk = p.K / BK;
  // ROT: stagger the K-tile order per block so concurrently running blocks (same expert, other
  // N-tiles; other experts) are not all at the same k offset of 4 KB-strided rows at once.
  const int rot = ROT ? (int)(((long long)(nt + 3 * e) * nk / p.ntn) % nk) : 0;

  // Prefetch registers: straight-line unrolled code over fixed-size arrays (no lambdas, no
  // conditionals around the loads -- hipcc otherwise demotes them to scratch / waits vmcnt(0)).
  u32x4 ra[NLA], rw[NLW];
#define GM_GLOAD(KT)                                                                   \
  do {                                                                                 \
    int kr_ = (KT) + rot; kr_ = kr_ >= nk ? kr_ - nk : kr_;                            \
    const size_t ao_ = (size_t)kr_ * BK, wo_ = (size_t)kr_ * w_step;                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NLA; ++i_) ra[i_] = ld16<false>(a_src[i_] + ao_); \
    _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) rw[i_] = ld16<NT>(w_src[i_] + wo_);    \
  } while (0)
#define GM_LSTORE(BUF)                                                                 \
  do {                                                                                 \
    uint16_t *da_ = sA + (BUF) * A_TILE, *dw_ = sW + (BUF) * W_TILE;                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NLA; ++i_) *reinterpret_cast<u32x4 *>(da_ + a_dst[i_]) = ra[i_]; \
    _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) *reinterpret_cast<u32x4 *>(dw_ + w_dst[i_]) = rw[i_]; \
  } while (0)

  GM_GLOAD(0);
  GM_LSTORE(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (NBUF == 2) ? (kt & 1) : 0;
    // prefetch the next K-tile (the last iteration re-reads its own tile: harmless, keeps the
    // loop body branch-free so the loads stay in flight across the MFMA block)
    const int kn = (kt + 1 < nk) ? kt + 1 : kt;
    GM_GLOAD(kn);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ABOVE the MFMA block (hipcc sinks them)

    const uint16_t *ca = sA + buf * A_TILE, *cw = sW + buf * W_TILE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      u32x4 fa[2], fw[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        fa[mi] = *reinterpret_cast<const u32x4 *>(ca + a_frag_off + mi * 32 * LDK + kk * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (W_KMAJOR) {
          fw[ni] = *reinterpret_cast<const u32x4 *>(cw + wk_frag_off + ni * 32 * LDK + kk * 16);
        } else {
          const uint16_t *ptr = cw + wt_frag_off + kk * 16 * GM_LDN + ni * 32;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t *)(ptr));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t *)(ptr + 4 * GM_LDN));
          u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
          u32x4 f = {lo2[0], lo2[1], hi2[0], hi2[1]};
          fw[ni] = f;
        }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mma<T>::run(fw[ni], fa[mi], acc[ni][mi]);
    }

    __builtin_amdgcn_sched_barrier(0);
    if (NBUF == 2) {
      GM_LSTORE(buf ^ 1);
      __syncthreads();
    } else {
      __syncthreads();  // every wave is done reading the single buffer
      GM_LSTORE(0);
      __syncthreads();
    }
  }
#undef GM_GLOAD
#undef GM_LSTORE

  // ---- epilogue: lane holds, per accumulator, row m = l31, features 8*rg + 4*kg + 0..3
  uint16_t *De = reinterpret_cast<uint16_t *>(p.D) + (size_t)e * p.d_stride_e;
  const uint16_t *be = p.bias ? reinterpret_cast<const uint16_t *>(p.bias) + (size_t)e * p.bias_stride_e : nullptr;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + l31;
    if (m >= row_limit) continue;
    uint16_t *drow = De + (size_t)(m / p.d_rpw) * p.d_stride_w + (size_t)(m % p.d_rpw) * p.ldd;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * 64 + ni * 32 + rg * 8 + kg * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][rg * 4 + r];
        if (be) {
          uint2 bb = *reinterpret_cast<const uint2 *>(be + n);
          uint16_t b4[4] = {(uint16_t)(bb.x & 0xffff), (uint16_t)(bb.x >> 16), (uint16_t)(bb.y & 0xffff), (uint16_t)(bb.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T tb;
            __builtin_memcpy(&tb, &b4[r], 2);
            v[r] += Elem<T>::to_f32(tb);
          }
        }
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T tv = Elem<T>::from_f32(activate<ACT>(v[r]));
          __builtin_memcpy(&o[r], &tv, 2);
        }
        uint2 ov;
        ov.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        ov.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2 *>(drow + n) = ov;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// ds_read_b64_tr_b16 permutation probe (self-test)
// -------------------------------------------------------------------------------------------
__global__ void probe_tr16_kernel(uint16_t *out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(lds + l * 4));
#pragma unroll
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)t[j];
}

extern "C" int tutel_amd_probe_tr16(uint16_t *out, tutel_stream_t stream) {
  TUTEL_REQUIRE(out != nullptr, "tutel_amd_probe_tr16: null pointer");
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  TUTEL_CHECK_LAUNCH("tutel_amd_probe_tr16");
  return 0;
}

// -------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------
template <int BK, int NBUF>
static constexpr size_t gemm_lds_bytes(bool kmajor) {
  return (size_t)(NBUF * GM_BM * (BK + 8) + NBUF * (kmajor ? GM_BN * (BK + 8) : BK * GM_LDN)) * 2;
}

template <typename T, bool KM, int ACT, int BK, int NBUF, int OCC, bool NT, bool ROT>
static int launch_cfg(const GemmArgs &a, int grid, hipStream_t st) {
  const size_t lds = gemm_lds_bytes<BK, NBUF>(KM);
  auto kern = expert_gemm_kernel<T, KM, ACT, BK, NBUF, OCC, NT, ROT>;
  static bool optin = false;  // one flag per instantiation: > 64 KiB of dynamic LDS needs the opt-in
  if (!optin) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    optin = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(GM_THREADS), lds, st, a);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
}

// Tile configuration per weight layout, from on-hardware A/B at the headline shape
// (tools/gemm_sweep.py; fc1 158.6 -> 136.2 us, fc2 129.0 -> 126.9 us):
//   k-major W (fc1): non-temporal weight loads + rotated K order.  Rows are 4 KB apart and every
//     concurrently running block would otherwise sit at the same k offset of its 128 rows --
//     HBM channel hot-spotting; staggering the start tile per (expert, N-tile) removes it.
//   n-major W (fc2): non-temporal weight loads, natural K order (256 B row chunks, 64 rows).
// TUTEL_AMD_GEMM_PLAIN=1 selects the plain variant (cached loads, natural order) for A/B runs.
static bool gemm_plain() {
  static int v = -1;
  if (v < 0) {
    const char *s = getenv("TUTEL_AMD_GEMM_PLAIN");
    v = (s && atoi(s) != 0) ? 1 : 0;
  }
  return v == 1;
}

template <typename T, bool KM, int ACT>
static int launch_gemm(const GemmArgs &a, int grid, hipStream_t st) {
  if (gemm_plain()) return launch_cfg<T, KM, ACT, 64, 2, 2, false, false>(a, grid, st);
  return launch_cfg<T, KM, ACT, 64, 2, 2, true, KM>(a, grid, st);
}

template <typename T, bool KM>
static int launch_gemm_act(const GemmArgs &a, int act, int grid, hipStream_t st) {
  switch (act) {
    case TUTEL_ACT_NONE: return launch_gemm<T, KM, TUTEL_ACT_NONE>(a, grid, st);
    case TUTEL_ACT_RELU: return launch_gemm<T, KM, TUTEL_ACT_RELU>(a, grid, st);
    case TUTEL_ACT_GELU: return launch_gemm<T, KM, TUTEL_ACT_GELU>(a, grid, st);
    case TUTEL_ACT_SILU: return launch_gemm<T, KM, TUTEL_ACT_SILU>(a, grid, st);
    default: tutel_set_error("tutel_amd_expert_gemm: unknown activation %d", act); return -1;
  }
}

extern "C" int tutel_amd_expert_gemm(const void *A, int64_t a_stride_e, int64_t a_stride_w,
                                     int a_rows_per_w, int lda, const void *W, int w_kmajor,
                                     int64_t w_stride_e, int ldw, const void *bias,
                                     int64_t bias_stride_e, void *D, int64_t d_stride_e,
                                     int64_t d_stride_w, int d_rows_per_w, int ldd, int E_loc,
                                     int R, int N, int K, int dtype, int act,
                                     const int32_t *row_counts, int row_align,
                                     tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype == TUTEL_BF16 || dtype == TUTEL_F16, "tutel_amd_expert_gemm: dtype must be bf16 or fp16 (got %d)", dtype);
  TUTEL_REQUIRE(E_loc >= 0 && R >= 0 && N >= 1 && K >= 1, "tutel_amd_expert_gemm: bad sizes E_loc=%d R=%d N=%d K=%d", E_loc, R, N, K);
  TUTEL_REQUIRE(K % 64 == 0, "tutel_amd_expert_gemm: K=%d must be a multiple of 64", K);
  TUTEL_REQUIRE(N % 8 == 0, "tutel_amd_expert_gemm: N=%d must be a multiple of 8", N);
  if (E_loc == 0 || R == 0) return 0;
  TUTEL_REQUIRE(A && W && D, "tutel_amd_expert_gemm: null pointer");
  TUTEL_REQUIRE(a_rows_per_w >= 1 && d_rows_per_w >= 1, "tutel_amd_expert_gemm: rows_per_w must be >= 1");
  TUTEL_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldd % 4 == 0 && a_stride_e % 8 == 0 && a_stride_w % 8 == 0 &&
                    w_stride_e % 8 == 0 && d_stride_e % 4 == 0 && d_stride_w % 4 == 0 && bias_stride_e % 4 == 0,
                "tutel_amd_expert_gemm: leading dimensions / strides must keep rows 16-byte aligned");
  TUTEL_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)D % 8) == 0 && ((uintptr_t)bias % 8) == 0,
                "tutel_amd_expert_gemm: pointers must be 16-byte aligned");
  TUTEL_REQUIRE(row_counts == nullptr || row_align >= 1, "tutel_amd_expert_gemm: row_align must be >= 1");

  GemmArgs a;
  a.A = A; a.a_stride_e = a_stride_e; a.a_stride_w = a_stride_w; a.a_rpw = a_rows_per_w; a.lda = lda;
  a.W = W; a.w_stride_e = w_stride_e; a.ldw = ldw;
  a.bias = bias; a.bias_stride_e = bias_stride_e;
  a.D = D; a.d_stride_e = d_stride_e; a.d_stride_w = d_stride_w; a.d_rpw = d_rows_per_w; a.ldd = ldd;
  a.E_loc = E_loc; a.R = R; a.N = N; a.K = K;
  a.row_counts = row_counts; a.row_align = row_align < 1 ? 1 : row_align;
  a.ntm = (R + GM_BM - 1) / GM_BM;
  a.ntn = (N + GM_BN - 1) / GM_BN;
  long long grid_ll = (long long)E_loc * a.ntm * a.ntn;
  TUTEL_REQUIRE(grid_ll < 0x7fffffffLL, "tutel_amd_expert_gemm: grid too large");
  const int grid = (int)grid_ll;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TUTEL_BF16)
    return w_kmajor ? launch_gemm_act<bf16_t, true>(a, act, grid, st) : launch_gemm_act<bf16_t, false>(a, act, grid, st);
  return w_kmajor ? launch_gemm_act<f16_t, true>(a, act, grid, st) : launch_gemm_act<f16_t, false>(a, act, grid, st);
}
