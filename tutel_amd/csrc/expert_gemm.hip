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
#include "common.h"

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64
#define GM_THREADS 256
#define GM_LDK (GM_BK + 8)    // elements per LDS row of a [rows][k] tile   (144 B)
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

template <typename T, bool W_KMAJOR, int ACT>
__global__ __launch_bounds__(GM_THREADS, 2) void expert_gemm_kernel(GemmArgs p) {
  // LDS: activations [2][BM][LDK], weights [2][BN][LDK] (k-major) or [2][BK][LDN] (n-major)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_TILE = GM_BM * GM_LDK;                               // elements
  constexpr int W_TILE = W_KMAJOR ? GM_BN * GM_LDK : GM_BK * GM_LDN;   // elements
  uint16_t *sA = reinterpret_cast<uint16_t *>(smem);
  uint16_t *sW = sA + 2 * A_TILE;

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
  const uint16_t *a_src[4];
  const uint16_t *w_src[4];
  int a_dst[4], w_dst[4];  // LDS element offsets inside a tile
  {
    const int kc = tid & 7, rbase = tid >> 3;  // [rows][k] tiles: 8 x 16B chunks per row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = rbase + 32 * i;
      int gr = min(m0 + r, p.R - 1);
      a_src[i] = Ae + (size_t)(gr / p.a_rpw) * p.a_stride_w + (size_t)(gr % p.a_rpw) * p.lda + kc * 8;
      a_dst[i] = r * GM_LDK + kc * 8;
      if (W_KMAJOR) {
        int gn = min(n0 + r, p.N - 1);
        w_src[i] = We + (size_t)gn * p.ldw + kc * 8;
        w_dst[i] = r * GM_LDK + kc * 8;
      }
    }
    if (!W_KMAJOR) {
      const int nc = tid & 15, kbase = tid >> 4;  // [k][n] tile: 16 x 16B chunks per row
      int gn = min(n0 + nc * 8, p.N - 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kr = kbase + 16 * i;
        w_src[i] = We + (size_t)kr * p.ldw + gn;
        w_dst[i] = kr * GM_LDN + nc * 8;
      }
    }
  }
  const size_t w_step = W_KMAJOR ? (size_t)GM_BK : (size_t)GM_BK * p.ldw;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (elements), constant over the K loop
  const int l31 = lane & 31, kg = lane >> 5;
  const int a_frag_off = (wm * 64 + l31) * GM_LDK + kg * 8;             // + mi*32*LDK + kk*16
  const int wk_frag_off = (wn * 64 + l31) * GM_LDK + kg * 8;            // k-major W
  // n-major W via ds_read_b64_tr_b16: 16-lane group g reads the 4(k) x 16(n) block at
  // rows kk*16 + (g>>1)*8 + h*4, cols wn*64 + ni*32 + (g&1)*16; lane i of the group supplies the
  // address of row (i>>2), cols 4*(i&3)..+3 and receives column i, rows 0..3.
  const int g16 = lane >> 4, i16 = lane & 15;
  const int wt_frag_off = ((g16 >> 1) * 8 + (i16 >> 2)) * GM_LDN + wn * 64 + (g16 & 1) * 16 + 4 * (i16 & 3);

  const int nk = p.K / GM_BK;

  // Prefetch registers: plain named vectors, straight-line code (no lambdas / conditionals --
  // hipcc demotes captured aggregates to scratch and then waits vmcnt(0) after every load).
  u32x4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define GM_GLOAD(KT)                                                                   \
  do {                                                                                 \
    const size_t ao_ = (size_t)(KT) * GM_BK, wo_ = (size_t)(KT) * w_step;              \
    ra0 = *reinterpret_cast<const u32x4 *>(a_src[0] + ao_);                            \
    ra1 = *reinterpret_cast<const u32x4 *>(a_src[1] + ao_);                            \
    ra2 = *reinterpret_cast<const u32x4 *>(a_src[2] + ao_);                            \
    ra3 = *reinterpret_cast<const u32x4 *>(a_src[3] + ao_);                            \
    rw0 = *reinterpret_cast<const u32x4 *>(w_src[0] + wo_);                            \
    rw1 = *reinterpret_cast<const u32x4 *>(w_src[1] + wo_);                            \
    rw2 = *reinterpret_cast<const u32x4 *>(w_src[2] + wo_);                            \
    rw3 = *reinterpret_cast<const u32x4 *>(w_src[3] + wo_);                            \
  } while (0)
#define GM_LSTORE(BUF)                                                                 \
  do {                                                                                 \
    uint16_t *da_ = sA + (BUF) * A_TILE, *dw_ = sW + (BUF) * W_TILE;                   \
    *reinterpret_cast<u32x4 *>(da_ + a_dst[0]) = ra0;                                  \
    *reinterpret_cast<u32x4 *>(da_ + a_dst[1]) = ra1;                                  \
    *reinterpret_cast<u32x4 *>(da_ + a_dst[2]) = ra2;                                  \
    *reinterpret_cast<u32x4 *>(da_ + a_dst[3]) = ra3;                                  \
    *reinterpret_cast<u32x4 *>(dw_ + w_dst[0]) = rw0;                                  \
    *reinterpret_cast<u32x4 *>(dw_ + w_dst[1]) = rw1;                                  \
    *reinterpret_cast<u32x4 *>(dw_ + w_dst[2]) = rw2;                                  \
    *reinterpret_cast<u32x4 *>(dw_ + w_dst[3]) = rw3;                                  \
  } while (0)

  GM_GLOAD(0);
  GM_LSTORE(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // prefetch the next K-tile (the last iteration re-reads its own tile: harmless, keeps the
    // loop body branch-free so the loads stay in flight across the MFMA block)
    const int kn = (kt + 1 < nk) ? kt + 1 : kt;
    GM_GLOAD(kn);
    __builtin_amdgcn_sched_barrier(0);  // keep the 8 loads ABOVE the MFMA block (hipcc sinks them)

    const uint16_t *ca = sA + buf * A_TILE, *cw = sW + buf * W_TILE;
#pragma unroll
    for (int kk = 0; kk < GM_BK / 16; ++kk) {
      u32x4 fa[2], fw[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        fa[mi] = *reinterpret_cast<const u32x4 *>(ca + a_frag_off + mi * 32 * GM_LDK + kk * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (W_KMAJOR) {
          fw[ni] = *reinterpret_cast<const u32x4 *>(cw + wk_frag_off + ni * 32 * GM_LDK + kk * 16);
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
    GM_LSTORE(buf ^ 1);
    __syncthreads();
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
template <typename T, bool KM>
static int launch_gemm_act(const GemmArgs &a, int act, int grid, size_t lds, hipStream_t st) {
  switch (act) {
    case TUTEL_ACT_NONE: hipLaunchKernelGGL((expert_gemm_kernel<T, KM, TUTEL_ACT_NONE>), dim3(grid), dim3(GM_THREADS), lds, st, a); break;
    case TUTEL_ACT_RELU: hipLaunchKernelGGL((expert_gemm_kernel<T, KM, TUTEL_ACT_RELU>), dim3(grid), dim3(GM_THREADS), lds, st, a); break;
    case TUTEL_ACT_GELU: hipLaunchKernelGGL((expert_gemm_kernel<T, KM, TUTEL_ACT_GELU>), dim3(grid), dim3(GM_THREADS), lds, st, a); break;
    case TUTEL_ACT_SILU: hipLaunchKernelGGL((expert_gemm_kernel<T, KM, TUTEL_ACT_SILU>), dim3(grid), dim3(GM_THREADS), lds, st, a); break;
    default: tutel_set_error("tutel_amd_expert_gemm: unknown activation %d", act); return -1;
  }
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
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
  if (E_loc == 0 || R == 0) return 0;
  TUTEL_REQUIRE(A && W && D, "tutel_amd_expert_gemm: null pointer");
  TUTEL_REQUIRE(K % GM_BK == 0, "tutel_amd_expert_gemm: K=%d must be a multiple of %d", K, GM_BK);
  TUTEL_REQUIRE(N % 8 == 0, "tutel_amd_expert_gemm: N=%d must be a multiple of 8", N);
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

  const size_t lds_k = (size_t)(2 * GM_BM * GM_LDK + 2 * GM_BN * GM_LDK) * 2;
  const size_t lds_n = (size_t)(2 * GM_BM * GM_LDK + 2 * GM_BK * GM_LDN) * 2;
  static bool attr_done = false;
  if (!attr_done) {  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
#define OPTIN(T, KM, ACT, BYTES) (void)hipFuncSetAttribute((const void *)expert_gemm_kernel<T, KM, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES))
#define OPTIN_ALL(T)                                                              \
  OPTIN(T, true, TUTEL_ACT_NONE, lds_k); OPTIN(T, true, TUTEL_ACT_RELU, lds_k);   \
  OPTIN(T, true, TUTEL_ACT_GELU, lds_k); OPTIN(T, true, TUTEL_ACT_SILU, lds_k);   \
  OPTIN(T, false, TUTEL_ACT_NONE, lds_n); OPTIN(T, false, TUTEL_ACT_RELU, lds_n); \
  OPTIN(T, false, TUTEL_ACT_GELU, lds_n); OPTIN(T, false, TUTEL_ACT_SILU, lds_n)
    OPTIN_ALL(bf16_t);
    OPTIN_ALL(f16_t);
#undef OPTIN_ALL
#undef OPTIN
    (void)hipGetLastError();
    attr_done = true;
  }

  if (dtype == TUTEL_BF16)
    return w_kmajor ? launch_gemm_act<bf16_t, true>(a, act, grid, lds_k, st) : launch_gemm_act<bf16_t, false>(a, act, grid, lds_n, st);
  return w_kmajor ? launch_gemm_act<f16_t, true>(a, act, grid, lds_k, st) : launch_gemm_act<f16_t, false>(a, act, grid, lds_n, st);
}
