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
//   * [K,N]-major weights (batched_fc2_w) can be consumed as stored: the k-contiguous fragment the
//     MFMA wants is produced by gfx950's transposing LDS read (training-mode modules; eval-mode
//     modules hand in a k-major copy laid out once, experts/ffn.py KMajorCache);
//   * row addressing folds the expert-parallel [W,E_loc,C,M] <-> [E_loc,W*C,M] permutes
//     (communicate.py:606-622) into the loads/stores.
// Three kernels share this file: the register-staged 128 x 128 kernel described above, its LDS-DMA
// sibling (global_load_lds, no VGPR / ds_write staging), and the 256-row-tile LDS-DMA kernel for
// more than 128 rows per expert (256 x 256, or 256 x 128 on a three-slot ring), where the bound is
// the L2 -> CU path rather than HBM.  launch_gemm() picks; every kernel walks k in the same order
// for a given output element, so the choice never changes a bit of the result.
#include <stdlib.h>

#include <mutex>

#include "common.h"

#include "gemm_dev.h"

// Register-staged 128 x 128 kernel: K-tile depth 64, double-buffered LDS (one barrier per K-tile), 2 blocks per CU.
// NT = non-temporal weight loads, ROT = K-tile rotation when the problem asks for it.
template <typename T, bool W_KMAJOR, int ACT, bool NT, bool ROT>
__global__ __launch_bounds__(GM_THREADS, 2) void expert_gemm_kernel(GemmArgs p) {
  constexpr int BK = 64, NBUF = 2;
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
      if (p.a_rows != nullptr) {  // fused fast_encode: bucket row -> token row of x (or the zero row)
        const int q = p.a_rows[(size_t)e * p.R + gr];
        a_src[i] = (q >= 0 ? reinterpret_cast<const uint16_t *>(p.A) + (size_t)(q % p.a_rows_mod) * p.lda
                           : reinterpret_cast<const uint16_t *>(p.a_zero)) + kc * 8;
      }
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
  GM_PRELOAD_BIAS();
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
  // K-tile rotation per PAIR of N-tiles: the 256-column kernel below shares one token tile between the
  // pair and therefore one k order; using the same order here makes every kernel produce bit-identical
  // sums for a given output element, whatever tile size the row count selects
  const int rot = (ROT && p.rot_on) ? (int)(((long long)((nt >> 1) + 3 * e) * nk / ((p.ntn + 1) >> 1)) % nk) : 0;

  // Prefetch registers: straight-line unrolled code over fixed-size arrays (no lambdas, no
  // conditionals around the loads -- hipcc otherwise demotes them to scratch / waits vmcnt(0)).
  // One register set: the loads of K-tile kt+1 are issued before tile kt is multiplied and written to
  // LDS after it (a second set, prefetch distance 2, was measured and did not pay).
  u32x4 ra0[NLA], rw0[NLW];
#define GM_GLOAD(RA, RW, KT)                                                           \
  do {                                                                                 \
    int kr_ = (KT); kr_ = kr_ < nk ? kr_ : nk - 1; /* past the end: re-read the last tile */ \
    kr_ += rot; kr_ = kr_ >= nk ? kr_ - nk : kr_;                                      \
    const size_t ao_ = (size_t)kr_ * BK, wo_ = (size_t)kr_ * w_step;                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NLA; ++i_) RA[i_] = ld16<false>(a_src[i_] + ao_); \
    _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) RW[i_] = ld16<NT>(w_src[i_] + wo_);    \
  } while (0)
#define GM_LSTORE(RA, RW, BUF)                                                         \
  do {                                                                                 \
    uint16_t *da_ = sA + (BUF) * A_TILE, *dw_ = sW + (BUF) * W_TILE;                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NLA; ++i_) *reinterpret_cast<u32x4 *>(da_ + a_dst[i_]) = RA[i_]; \
    _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) *reinterpret_cast<u32x4 *>(dw_ + w_dst[i_]) = RW[i_]; \
  } while (0)
#define GM_COMPUTE(BUF)                                                                \
  do {                                                                                 \
    const uint16_t *ca = sA + (BUF) * A_TILE, *cw = sW + (BUF) * W_TILE;               \
    /* all 16 fragment reads of the K-tile are issued before the first MFMA: LDS latency is */ \
    /* paid once per tile, not once per 16-deep slice (a skeleton with this shape streams  */ \
    /* at 6 TB/s, tools/wstream_bench.hip; the 4-reads/4-MFMAs form stalled 4x per tile).  */ \
    u32x4 fa[BK / 16][2], fw[BK / 16][2];                                              \
    _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk) {                           \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                 \
        fa[kk][mi] = *reinterpret_cast<const u32x4 *>(ca + a_frag_off + mi * 32 * LDK + kk * 16); \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                               \
        if (W_KMAJOR) {                                                                \
          fw[kk][ni] = *reinterpret_cast<const u32x4 *>(cw + wk_frag_off + ni * 32 * LDK + kk * 16); \
        } else {                                                                       \
          const uint16_t *ptr = cw + wt_frag_off + kk * 16 * GM_LDN + ni * 32;         \
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                        \
              (__attribute__((address_space(3))) s16x4_t *)(ptr));                     \
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                        \
              (__attribute__((address_space(3))) s16x4_t *)(ptr + 4 * GM_LDN));        \
          u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi); \
          u32x4 f = {lo2[0], lo2[1], hi2[0], hi2[1]};                                  \
          fw[kk][ni] = f;                                                              \
        }                                                                              \
      }                                                                                \
    }                                                                                  \
    __builtin_amdgcn_sched_barrier(0); /* hipcc otherwise re-interleaves reads and MFMAs */ \
    _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk)                             \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                 \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                               \
          acc[ni][mi] = Mma<T>::run(fw[kk][ni], fa[kk][mi], acc[ni][mi]);              \
  } while (0)

  GM_GLOAD(ra0, rw0, 0);
  GM_LSTORE(ra0, rw0, 0);
  __syncthreads();                      // tile 0 in LDS buffer 0
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    GM_GLOAD(ra0, rw0, kt + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ABOVE the MFMA block (hipcc sinks them)
    GM_COMPUTE(buf);
    __builtin_amdgcn_sched_barrier(0);
    GM_LSTORE(ra0, rw0, buf ^ 1);
    __syncthreads();
  }
#undef GM_GLOAD
#undef GM_LSTORE
#undef GM_COMPUTE

  gemm_epilogue<T, ACT>(p, acc, bias_r, e, m0, n0, wm, wn, l31, kg, row_limit);
}

// -------------------------------------------------------------------------------------------
// LDS-DMA variant (global_load_lds): tiles go HBM/L2 -> LDS without passing through VGPRs or the
// ds_write path.  On-hardware ablation of the register-staged kernel above (tools/gemm_sweep.py):
// its ds_write_b128 staging alone costs 23-44 us of 140 at the headline shape and 100 of 320 us at
// R = 1024 rows/expert (1.07 GB through a ~79 B/clk/CU store path), so it is removed here.
//
// LDS-DMA writes lane i's 16 bytes at (wave-uniform base + 16*i): the LDS image of a tile is
// LINEAR (no row padding).  Bank conflicts are avoided by permuting which 16-byte chunk of a
// row each lane FETCHES (source-side XOR swizzle; same 128 B / 256 B global segment per row, so
// coalescing is unchanged) and applying the same XOR when fragments are read:
//   [rows][64 k] tiles (128 B rows):  chunk c of row r lives at position c ^ ((r >> 1) & 7)
//       -> ds_read_b128 of 32 consecutive rows x one chunk column is conflict-free
//   [64 k][128 n] weight tile (256 B rows): chunk c of row r lives at position c ^ ((r & 3) << 2)
//       -> the 4(k) x 16(n) blocks of ds_read_b64_tr_b16 cover all 64 banks exactly once
// 2 LDS stages x (16 KB tokens + 16 KB weights) = 64 KB per block, 2 blocks per CU.
// -------------------------------------------------------------------------------------------
// (Round 4 gave this kernel a three-slot weight ring with a counted s_waitcnt vmcnt + bare s_barrier instead of the per-K-tile
// __syncthreads drain, 80 KB per block, still two blocks per CU -- VERDICT r3's proposal.  Measured: 113.1 / 112.9 us against
// 112.4 / 111.9 us for this two-stage form at the headline shape (profiles/r04_headline_ab.json): with two blocks per CU the other
// block's tile is in flight whenever this one drains, and the memory system is saturated either way.  Not kept.  What did pay is
// fewer bytes crossing L2 -> LDS per weight byte: the 128 x 256 tile of expert_gemm_big_kernel<.., BM = 128> below.)
template <typename T, bool W_KMAJOR, int ACT, bool NT, bool ROT>
__global__ __launch_bounds__(GM_THREADS, 2) void expert_gemm_glds_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *sA = reinterpret_cast<uint16_t *>(smem);  // [2][GL_STAGE]
  uint16_t *sW = sA + 2 * GL_STAGE;                   // [2][GL_STAGE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

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

  // ---- DMA source pointers: wave `wid` issues pieces j = wid*4 + i (i < 4) of each tile; piece j
  // of a [rows][64k] tile = rows 8j..8j+7 (1 KiB), lane -> (row 8j + lane/8, position lane%8).
  const uint16_t *a_src[4], *w_src[4];
  int gr4[4], slot4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gr4[i] = min(m0 + 8 * (wid * 4 + i) + (lane >> 3), p.R - 1);
  gather_rows4(p, e, gr4, slot4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * (wid * 4 + i) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int gr = gr4[i];
    a_src[i] = Ae + (size_t)(gr / p.a_rpw) * p.a_stride_w + (size_t)(gr % p.a_rpw) * p.lda + c * 8;
    if (p.a_rows != nullptr) {  // fused fast_encode: bucket row -> token row of x (or the zero row)
      const int q = slot4[i];
      a_src[i] = (q >= 0 ? reinterpret_cast<const uint16_t *>(p.A) + (size_t)(q % p.a_rows_mod) * p.lda
                         : reinterpret_cast<const uint16_t *>(p.a_zero)) + c * 8;
    }
    if (W_KMAJOR) {
      const int gn = min(n0 + r, p.N - 1);
      w_src[i] = We + (size_t)gn * p.ldw + c * 8;
    } else {
      // piece j of the [64k][128n] tile = k-rows 4j..4j+3, lane -> (row 4j + lane/16, position lane%16)
      const int kr = 4 * (wid * 4 + i) + (lane >> 4);
      const int cn = (lane & 15) ^ ((kr & 3) << 2);
      const int gn = min(n0 + cn * 8, p.N - 8);
      w_src[i] = We + (size_t)kr * p.ldw + gn;
    }
  }
  const size_t w_step = W_KMAJOR ? (size_t)GL_BK : (size_t)GL_BK * p.ldw;
  const int piece0 = wid * 4 * 512;  // element offset of this wave's first 1 KiB piece inside a tile

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment read offsets (elements)
  const int l31 = lane & 31, kg = lane >> 5;
  GM_PRELOAD_BIAS();
  const int sw = (l31 >> 1) & 7;  // (row >> 1) & 7 with row = 32*x + l31
  int frag_k[4];                   // chunk position of (kk, kg) for this lane's row, in elements
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_k[kk] = (((kk * 2 + kg) ^ sw) << 3);
  const int a_row = (wm * 64 + l31) * GL_BK;  // + mi*32*64
  const int wk_row = (wn * 64 + l31) * GL_BK; // + ni*32*64
  // [k][n] tile via ds_read_b64_tr_b16 (see the register-staged kernel for the block geometry)
  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2;
  const int c_lo = wn * 8 + (g16 & 1) * 2 + ((i16 & 3) >> 1);          // 16B chunk of ni = 0, before swizzle
  const int wt_row = ((g16 >> 1) * 8 + q4) * GM_BN;                     // + (kk*16 + h*4) * 128
  const int wt_c0 = (((c_lo) ^ (q4 << 2)) << 3) + (i16 & 1) * 4;        // element offset inside the row, ni = 0
  const int wt_c1 = (((c_lo + 4) ^ (q4 << 2)) << 3) + (i16 & 1) * 4;    // ni = 1

  const int nk = p.K / GL_BK;
  // K-tile rotation per PAIR of N-tiles: the 256-column kernel below shares one token tile between the
  // pair and therefore one k order; using the same order here makes every kernel produce bit-identical
  // sums for a given output element, whatever tile size the row count selects
  const int rot = (ROT && p.rot_on) ? (int)(((long long)((nt >> 1) + 3 * e) * nk / ((p.ntn + 1) >> 1)) % nk) : 0;

#define GL_ISSUE(KT, BUF)                                                              \
  do {                                                                                 \
    int kr_ = (KT) + rot; kr_ = kr_ >= nk ? kr_ - nk : kr_;                            \
    const size_t ao_ = (size_t)kr_ * GL_BK, wo_ = (size_t)kr_ * w_step;                \
    uint16_t *da_ = sA + (BUF) * GL_STAGE + piece0, *dw_ = sW + (BUF) * GL_STAGE + piece0; \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) glds16(a_src[i_] + ao_, da_ + i_ * 512, false); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) glds16(w_src[i_] + wo_, dw_ + i_ * 512, NT);    \
  } while (0)

#define GL_LOAD_FRAGS(FA, FW, KK)                                                      \
  do {                                                                                 \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                   \
      FA[mi] = *reinterpret_cast<const u32x4 *>(ca + a_row + mi * 32 * GL_BK + frag_k[KK]); \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                 \
      if (W_KMAJOR) {                                                                  \
        FW[ni] = *reinterpret_cast<const u32x4 *>(cw + wk_row + ni * 32 * GL_BK + frag_k[KK]); \
      } else {                                                                         \
        const uint16_t *ptr = cw + wt_row + (KK) * 16 * GM_BN + (ni ? wt_c1 : wt_c0);  \
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
#define GL_MMA(FA, FW)                                                                 \
  do {                                                                                 \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                   \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                 \
        acc[ni][mi] = Mma<T>::run(FW[ni], FA[mi], acc[ni][mi]);                        \
  } while (0)

  GL_ISSUE(0, 0);
  __syncthreads();  // with a DMA in flight this is vmcnt(0) + barrier: tile 0 is in LDS stage 0

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GL_ISSUE(kt + 1, buf ^ 1);  // block-uniform; next tile streams in during the MFMAs

    const uint16_t *ca = sA + buf * GL_STAGE, *cw = sW + buf * GL_STAGE;
    // all fragment reads of the K-tile first, then the 16 MFMAs (LDS latency paid once per tile)
    u32x4 fa[4][2], fw[4][2];
    GL_LOAD_FRAGS(fa[0], fw[0], 0);
    GL_LOAD_FRAGS(fa[1], fw[1], 1);
    GL_LOAD_FRAGS(fa[2], fw[2], 2);
    GL_LOAD_FRAGS(fa[3], fw[3], 3);
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise re-interleaves reads and MFMAs
    GL_MMA(fa[0], fw[0]);
    GL_MMA(fa[1], fw[1]);
    GL_MMA(fa[2], fw[2]);
    GL_MMA(fa[3], fw[3]);

    __syncthreads();  // all waves done with stage `buf`; next tile's DMA has landed (vmcnt(0))
  }
#undef GL_ISSUE
#undef GL_LOAD_FRAGS
#undef GL_MMA

  gemm_epilogue<T, ACT>(p, acc, bias_r, e, m0, n0, wm, wn, l31, kg, row_limit);
}

// One (expert, M-tile, N-tile) per workgroup: gemm_big_tile (gemm_dev.h) under the XCD-aware work order.
template <typename T, bool W_KMAJOR, int ACT, int NI, int NS, bool BUF = false, int BM = GB_BM, bool FL = false>
__global__ __launch_bounds__(BM * 2, 2) void expert_gemm_big_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = gridDim.x;
  int w;
  {
    const int b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, pos = b >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  gemm_big_tile<T, W_KMAJOR, ACT, NI, NS, BUF, BM, FL>(p, w / (p.ntm * p.ntn), w % p.ntm, (w / p.ntm) % p.ntn, smem);
}

// -------------------------------------------------------------------------------------------
// 256 x 256 "ping-pong" kernel (k-major weights): the MFMA-bound regime's main kernel.
//
// PMC of expert_gemm_big_kernel above (profiles/r01_pmc_big_gemm.txt): MFMA busy 37 % of SIMD cycles,
// 46 % of wave cycles parked in s_waitcnt -- every K-tile ends in a __syncthreads that drains all
// LDS-DMA (vmcnt(0)), and both waves of a SIMD read fragments / run MFMAs at the same time.  Here:
//   * the 8 waves form two groups (waves 0-3, 4-7: one wave of each per SIMD) that run the SAME
//     instruction stream half a phase apart: while one group issues its 8 MFMAs of a phase (256 cycles
//     of the SIMD's matrix pipe) the other group issues the phase's LDS-DMA, fragment reads and waits.
//     Every s_barrier flips the roles; the matrix pipe always has exactly one wave feeding it.
//   * a K-tile (64 deep) is 4 phases; phase q multiplies the wave's 64 token rows with its q-th 32-column
//     strip of weights (8 x v_mfma_32x32x16).  Token fragments are read once per K-tile (phase 0) and
//     kept in registers; each phase reads 4 weight fragments.
//   * LDS holds two K-tiles (2 x (32 KB tokens + 32 KB weights)), recycled piecewise: the token tile
//     is dead after phase 0, weight strip q after phase q, and each piece is re-filled for tile j+2 two
//     phases after its last read.  DMA is never drained: every phase issues exactly 2 LDS-DMA
//     instructions per wave and waits with a COUNTED s_waitcnt vmcnt(6) (the data a phase reads was
//     issued >= 4 phases = 8 instructions earlier), raw s_barrier, no __syncthreads in the loop.
// Hazards (p = phase index, one barrier interval = half a phase; group 1 runs one interval late):
//   RAW: data read in phase p was issued in phases <= p-4; each wave's vmcnt(6) in phase p-1 retires its
//        own pieces, and the barrier(s) between that wait and any reader's phase p order the rest.
//   WAR: the DMA issued in phase p overwrites pieces last read in phases <= p-2; those reads were
//        retired (lgkmcnt(0)) by both groups before the barrier that precedes phase p's issue.
// Same k order per output element as the other kernels (rotation per 256-column tile, kk ascending):
// results are bit-identical to theirs.
// -------------------------------------------------------------------------------------------
#define PP_BUF (4 * GL_STAGE)   // elements per LDS K-tile buffer: [256][64] tokens + [256][64] weights = 64 KB

// SPLITK (round 5): launches whose 256 x 256 tiles cover only half the chip (96 .. 191 tiles: one pipeline stage of an 8-way
// expert-parallel rank is 128) give every tile to TWO workgroups, neighbours in the XCD-aware work order (same XCD, dispatched
// back to back), each running the K loop over one half of K at the full tile's flop-per-byte.  At the end each hands the partial
// accumulators of ONE 128-column group to its partner (the four waves of that column group store 128 fp32 per lane with
// system-scope write-through stores, 1 KB per instruction and wave, then raise a flag) and finishes the other column group: the
// four remaining waves wait for the partner's flag, add its partial (lower-K half + upper-K half: one fp32 add per element, the
// same whichever workgroup performs it) and run the epilogue on 64 x 128 each.  The hand-over is 128 KB out and 128 KB in per
// workgroup.  Results differ from the unsplit kernel's in the last bits of the fp32 sum (two half sums added instead of one
// running sum); a split launch is deterministic and reproduces itself.
template <typename T, int ACT, bool W_ONCE, bool RAGGED = false, bool EARLY_BIAS = false, bool SPLITK = false>
__global__ __launch_bounds__(GB_THREADS, 2) void expert_gemm_pp_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *lds = reinterpret_cast<uint16_t *>(smem);  // [2][ tokens 2*GL_STAGE | weights 2*GL_STAGE ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;  // 64-row group, 128-column group

  const int nb = gridDim.x;
  int w;
  {
    const int b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, pos = b >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int half = SPLITK ? (w & 1) : 0;  // which half of K this workgroup multiplies (and which 128-column group it finishes)
  if (SPLITK) w >>= 1;
  const int tile_id = w;
  const int mt = w % p.ntm;
  const int nt = (w / p.ntm) % p.ntn;
  const int e = w / (p.ntm * p.ntn);
  const int m0 = mt * GB_BM, n0 = nt * 256;

  int row_limit = p.R;
  if (p.row_counts != nullptr) {
    int c = p.row_counts[e];
    c = (c + p.row_align - 1) / p.row_align * p.row_align;
    row_limit = min(row_limit, c);
  }
  if (m0 >= row_limit) return;

  const uint16_t *Ae = reinterpret_cast<const uint16_t *>(p.A) + (size_t)e * p.a_stride_e;
  const uint16_t *We = reinterpret_cast<const uint16_t *>(p.W) + (size_t)e * p.w_stride_e;

  // ---- DMA sources.  A piece = 8 LDS rows x 128 B (one wave instruction); lane -> (row lane/8, 16-byte
  // position lane%8 holding chunk (lane%8) ^ ((row >> 1) & 7) of the row: the bank-conflict swizzle).
  // Token tile: LDS row = tile row.  Pieces {2w, 2w+1} (first half) and {16+2w, 17+2w} (second half).
  // Weight tile: LDS row = (strip q, column group wn, column i) -> (2q + wn)*32 + i, so that strip q of
  // BOTH column groups is one contiguous 8 KB range; pieces {2w, 2w+1} (strips 0,1), {16+2w, 17+2w} (2,3).
  // Issued as `buffer_load_dwordx4 ... lds`: descriptor in SGPRs, a per-lane 32-bit byte offset that never
  // changes, and the K-tile offset as the scalar offset operand -- NO vector ALU work per issue.  (With 64-bit
  // per-lane pointers every issue needed two v_lshl_add_u64, and VALU instructions of the wave in its memory
  // part starve behind the partner wave's MFMAs: 315 cycles per 2 issues measured with s_memtime, vs ~100
  // with both waves in their memory part.)  Out-of-range offsets return ZEROS into LDS (probed on gfx950,
  // tools/scratch/oob_lds.hip): that is the all-zero row of an empty bucket slot in the fused fast_encode.
  int a_off[4], w_off[4];
  int gr4[4], slot4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gr4[i] = min(m0 + 8 * ((i >> 1) * 16 + 2 * wid + (i & 1)) + (lane >> 3), p.R - 1);
  gather_rows4(p, e, gr4, slot4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = (i >> 1) * 16 + 2 * wid + (i & 1);
    const int r = 8 * piece + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    {
      const int gr = gr4[i];
      size_t off = (size_t)(gr / p.a_rpw) * p.a_stride_w + (size_t)(gr % p.a_rpw) * p.lda;  // elements from Ae
      if (p.a_rows != nullptr) {
        const int q = slot4[i];
        off = q >= 0 ? (size_t)(q % p.a_rows_mod) * p.lda : (size_t)0x3ffff800u;            // elements from p.A; empty -> out of range
      }
      if (m0 + r >= row_limit) off = (size_t)0x3ffff800u;  // rows past the expert's row count: zeros, no memory traffic
      a_off[i] = (int)(unsigned)(off * 2 + c * 16);
    }
    {
      const int col = ((r >> 5) & 1) * 128 + (r >> 6) * 32 + (r & 31);  // LDS row r -> column of the 256-column tile
      const int gn = min(n0 + col, p.N - 1);
      w_off[i] = (int)(unsigned)(((size_t)gn * p.ldw) * 2 + c * 16);
    }
  }
  // descriptors: raw buffers (stride 0); the token descriptor's size is the span of the rows it addresses (the token
  // array when rows are gathered, else this expert's rows), so that the marker offset above is out of range
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(p.a_rows != nullptr ? reinterpret_cast<const uint16_t *>(p.A) : Ae), 0,
      p.a_span_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(We), 0, -1, 0x00020000);
  const int piece_lo = 2 * wid * 512, piece_hi = (16 + 2 * wid) * 512;  // element offsets of the wave's pieces in a 32 KB tile
  // W_ONCE (one M-tile per expert and the chip covered): every weight byte is fetched by exactly one block ->
  // no-allocate loads (see expert_gemm_big_kernel); a template axis so the issue path has no branch

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, kg = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  int frag_k[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_k[kk] = (((kk * 2 + kg) ^ sw) << 3);
  // RAGGED (dropless capacity 157 on a 256-row tile, megablocks row counts): the MFMAs of a 32-row group entirely
  // past the expert's row count are skipped (wave-uniform branch; its rows arrive as zeros, its accumulators are never
  // stored).  A separate instantiation: the branches cost the full-tile kernel registers and schedule.
  const bool mi_on[2] = {__builtin_amdgcn_readfirstlane(m0 + wm * 64) < row_limit,
                         __builtin_amdgcn_readfirstlane(m0 + wm * 64 + 32) < row_limit};
  const int a_row = (wm * 64 + l31) * GL_BK;                    // + mi*32*64
  const int w_row = 2 * GL_STAGE + (wn * 32 + l31) * GL_BK;     // + q*64*64, weights follow the token tile

  const int nk = SPLITK ? p.K / (2 * GL_BK) : p.K / GL_BK;  // K-tiles of THIS workgroup
  const int kt0 = SPLITK ? half * nk : 0;                   // its first K-tile (split launches never rotate: R >= 256)
  const int rot = (!SPLITK && p.rot_on) ? (int)(((long long)(nt + 3 * e) * nk / p.ntn) % nk) : 0;

#define PP_KOFF(J, KO)                                     \
  int KO;                                                  \
  {                                                        \
    int kr_ = (J) + rot; kr_ = kr_ >= nk ? kr_ - nk : kr_; \
    KO = (kr_ + kt0) * (GL_BK * 2); /* bytes */            \
  }
  // LDS-DMA issue of one half of a tile (2 instructions per wave): HALF 0 -> pieces 2w,2w+1; 1 -> 16+2w,17+2w
#define PP_ISSUE_A(J, BUF, HALF)                                                         \
  do {                                                                                   \
    PP_KOFF(J, ko_);                                                                     \
    uint16_t *d_ = lds + (BUF) * PP_BUF + ((HALF) ? piece_hi : piece_lo);                \
    bdma16<false>(rs_a, a_off[2 * (HALF)], ko_, d_);                                     \
    bdma16<false>(rs_a, a_off[2 * (HALF) + 1], ko_, d_ + 512);                           \
  } while (0)
#define PP_ISSUE_W(J, BUF, HALF)                                                         \
  do {                                                                                   \
    PP_KOFF(J, ko_);                                                                     \
    uint16_t *d_ = lds + (BUF) * PP_BUF + 2 * GL_STAGE + ((HALF) ? piece_hi : piece_lo); \
    bdma16<W_ONCE>(rs_w, w_off[2 * (HALF)], ko_, d_);                                    \
    bdma16<W_ONCE>(rs_w, w_off[2 * (HALF) + 1], ko_, d_ + 512);                          \
  } while (0)

  u32x4 fa[4][2], fw[4];
  // one phase: [DMA issue] [fragment reads] [counted DMA wait] barrier [8 MFMAs at raised priority] barrier
#define PP_PHASE(MODE, Q, BUF, STEADY, ISSUE)                                                  \
  do {                                                                                   \
    { ISSUE; }                                                                           \
    const uint16_t *cb_ = lds + (BUF) * PP_BUF;                                          \
    if ((Q) == 0) {                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                   \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                 \
          fa[kk][mi] = *reinterpret_cast<const u32x4 *>(cb_ + a_row + mi * 32 * GL_BK + frag_k[kk]); \
    }                                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                     \
      fw[kk] = *reinterpret_cast<const u32x4 *>(cb_ + w_row + (Q) * 64 * GL_BK + frag_k[kk]); \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    if (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                         \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
    __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    __builtin_amdgcn_s_setprio(1);                                                       \
    if ((MODE) == 2) {                                                                   \
      _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                   \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                 \
          acc[Q][mi] = Mma<T>::run(fw[kk], fa[kk][mi], acc[Q][mi]);                      \
    } else if ((MODE) == 1) {                                                            \
      _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                   \
        acc[Q][0] = Mma<T>::run(fw[kk], fa[kk][0], acc[Q][0]);                           \
    }                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    __builtin_amdgcn_s_barrier();                                                        \
  } while (0)

  // ---- prologue: tiles 0 (tokens + weights) and 1 (tokens) in flight; wait for tile 0 only
  PP_ISSUE_A(0, 0, 0); PP_ISSUE_A(0, 0, 1);
  PP_ISSUE_W(0, 0, 0); PP_ISSUE_W(0, 0, 1);
  if (nk > 1) {
    PP_ISSUE_A(1, 1, 0); PP_ISSUE_A(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wid >= 4) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier interval behind group 0

  // the K loop, as a macro over MODE = number of live 32-row groups of this wave (2: both, 1: the first, 0: none)
#define PP_MAINLOOP(MODE)                                                                \
  do {                                                                                   \
    int j = 0;                                                                           \
    /* steady state: two K-tiles per iteration (static LDS buffer indices), every issue real */ \
    for (; j + 3 < nk; j += 2) {                                                         \
      PP_PHASE(MODE, 0, 0, true, PP_ISSUE_W(j + 1, 1, 0));                               \
      PP_PHASE(MODE, 1, 0, true, PP_ISSUE_W(j + 1, 1, 1));                               \
      PP_PHASE(MODE, 2, 0, true, PP_ISSUE_A(j + 2, 0, 0));                               \
      PP_PHASE(MODE, 3, 0, true, PP_ISSUE_A(j + 2, 0, 1));                               \
      PP_PHASE(MODE, 0, 1, true, PP_ISSUE_W(j + 2, 0, 0));                               \
      PP_PHASE(MODE, 1, 1, true, PP_ISSUE_W(j + 2, 0, 1));                               \
      PP_PHASE(MODE, 2, 1, true, PP_ISSUE_A(j + 3, 1, 0));                               \
      PP_PHASE(MODE, 3, 1, true, PP_ISSUE_A(j + 3, 1, 1));                               \
    }                                                                                    \
    /* tail (last <= 3 tiles): issues only while tiles remain, every wait drains; one copy of the four phases with */ \
    /* the LDS buffer chosen at run time (two static copies made hipcc hoist ~40 address registers into scratch)   */ \
    for (; j < nk; ++j) {                                                                \
      const int cur = __builtin_amdgcn_readfirstlane(j & 1), nxt = cur ^ 1;              \
      const bool more1 = j + 1 < nk, more2 = j + 2 < nk;                                 \
      PP_PHASE(MODE, 0, cur, false, if (more1) PP_ISSUE_W(j + 1, nxt, 0));               \
      PP_PHASE(MODE, 1, cur, false, if (more1) PP_ISSUE_W(j + 1, nxt, 1));               \
      PP_PHASE(MODE, 2, cur, false, if (more2) PP_ISSUE_A(j + 2, cur, 0));               \
      PP_PHASE(MODE, 3, cur, false, if (more2) PP_ISSUE_A(j + 2, cur, 1));               \
    }                                                                                    \
  } while (0)
  // EARLY_BIAS: the epilogue's 16 bias loads per lane go out before the K loop (32 more live registers) instead of after it,
  // where their L2 round trip is exposed once per tile
  uint2 bias_r[4][4];
#define PP_LOAD_BIAS()                                                                                \
  do {                                                                                                \
    const uint16_t *be_ = p.bias ? reinterpret_cast<const uint16_t *>(p.bias) + (size_t)e * p.bias_stride_e : nullptr; \
    _Pragma("unroll") for (int ni_ = 0; ni_ < 4; ++ni_)                                               \
      _Pragma("unroll") for (int rg_ = 0; rg_ < 4; ++rg_) {                                           \
        int n_ = n0 + wn * 128 + ni_ * 32 + rg_ * 8 + kg * 4;                                         \
        n_ = n_ < p.N ? n_ : p.N - 4;                                                                 \
        bias_r[ni_][rg_] = be_ ? *reinterpret_cast<const uint2 *>(be_ + n_) : make_uint2(0u, 0u);     \
      }                                                                                               \
  } while (0)
  if (EARLY_BIAS) PP_LOAD_BIAS();
  // RAGGED: one branch per wave OUTSIDE the loop picks the loop version (every version runs the same DMA issues,
  // waits and barriers; only the MFMA blocks differ), so the loop bodies stay branch-free
  if (!RAGGED || mi_on[1]) PP_MAINLOOP(2);
  else if (mi_on[0]) PP_MAINLOOP(1);
  else PP_MAINLOOP(0);
#undef PP_MAINLOOP
  if (wid < 4) __builtin_amdgcn_s_barrier();  // group 0 catches up: equal barrier counts for all waves
#undef PP_PHASE
#undef PP_ISSUE_A
#undef PP_ISSUE_W
#undef PP_KOFF

  if (SPLITK) {
    // hand-over slots: [tile][half][wm] -> 128 fp32 per lane laid out [accumulator register 0..127][lane] (256 contiguous bytes per
    // wave instruction) + one flag word.  Slot (tile, h, wm) is WRITTEN by workgroup h (its waves of column group 1 - h) and
    // READ by workgroup 1 - h (its waves of column group 1 - h).
    const size_t slot = ((size_t)tile_id * 2 + half) * 4 + wm;
    if (wn != half) {  // giver
      uint32_t *dst = reinterpret_cast<uint32_t *>(p.sk_ws) + slot * (128 * 64) + lane;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(dst + ((ni * 2 + mi) * 16 + r) * 64, __float_as_uint(acc[ni][mi][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every write-through store of this wave has completed at system scope
      if (lane == 0) __hip_atomic_store(p.sk_flags + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (p.d_peer != nullptr) peer_canary_store(p.d_peer, p.d_can);  // (its block-wide barrier: every wave of the block passes one)
      return;
    }
    const size_t pslot = ((size_t)tile_id * 2 + (1 - half)) * 4 + wm;  // the partner's slot for my column group
    {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(p.sk_flags + pslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 400000000LL) __builtin_trap();  // 4 s of the 100 MHz clock: the partner workgroup never ran -- cannot happen with in-order dispatch
      }
    }
    asm volatile("" ::: "memory");
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.sk_ws) + pslot * (128 * 64) + lane;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float o = __uint_as_float(__hip_atomic_load(src + ((ni * 2 + mi) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
          acc[ni][mi][r] = half == 0 ? acc[ni][mi][r] + o : o + acc[ni][mi][r];  // (lower-K half) + (upper-K half)
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(p.sk_flags + pslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // consumed: the slot is free for the next launch
  }
  if (!EARLY_BIAS) PP_LOAD_BIAS();
#undef PP_LOAD_BIAS
  if ((p.ldd & 7) || (p.d_stride_e & 7) || (p.d_stride_w & 7) || (reinterpret_cast<uintptr_t>(p.D) & 15)) {
    gemm_epilogue<T, ACT, 4>(p, acc, bias_r, e, m0, n0, wm, wn, l31, kg, row_limit);  // rows not 16-byte aligned
    return;
  }
  // after the final barrier no wave reads the K-tile buffers any more and every DMA has landed: LDS is free
  gemm_epilogue_lds<T, ACT>(p, acc, bias_r, smem + wid * (64 * EP_PITCH), e, m0, n0, wm, wn, lane, row_limit);
}

// > 64 KB of dynamic LDS needs hipFuncSetAttribute -- once per (kernel, DEVICE): the attribute belongs to the function object
// of the current device (a per-process flag left every device but the first without it)
bool tutel_lds_optin(const void *kern, size_t lds) {
  constexpr int MAXK = 256, MAXD = 64;
  static const void *seen[MAXK];
  static uint64_t done[MAXK];  // bit d: device d has the attribute
  static int n = 0;
  // launches may come from several host threads (one per device is common): the table is shared
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXD) dev = -1;
  int i = 0;
  for (; i < n; ++i)
    if (seen[i] == kern) break;
  if (i < n && dev >= 0 && (done[i] >> dev & 1)) return true;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipGetLastError();
  if (e != hipSuccess) {
    tutel_set_error("tutel_amd_expert_gemm: cannot opt in to %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    return false;
  }
  if (i == n && n < MAXK) { seen[n] = kern; done[n] = 0; ++n; }
  if (i < MAXK && dev >= 0) done[i] |= 1ull << dev;
  return true;
}

template <typename T, int ACT, bool W_ONCE, bool RAGGED = false, bool EARLY_BIAS = false>
static int launch_pp_cfg(const GemmArgs &b, hipStream_t st) {
  const size_t lds = (size_t)8 * 64 * EP_PITCH;  // 136 KB: the epilogue staging (8 waves x 64 rows x 272 B) > the two K-tile buffers (128 KB)
  static_assert((size_t)8 * 64 * EP_PITCH >= (size_t)2 * PP_BUF * 2, "LDS request must cover the K-tile buffers");
  auto kern = expert_gemm_pp_kernel<T, ACT, W_ONCE, RAGGED, EARLY_BIAS>;
  if (!tutel_lds_optin((const void *)kern, lds)) return -1;
  hipLaunchKernelGGL(kern, dim3(b.E_loc * b.ntm * b.ntn), dim3(GB_THREADS), lds, st, b);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
}

// ---- split-K launches: workspace per (device, stream) -------------------------------------------------------------------------
// 256 KB of partial accumulators + 8 flag words per tile.  Launches on one stream are ordered, so one workspace per stream is
// enough (the two stages of an overlapped pipeline run on two side streams: two workspaces); it grows on demand and is never
// freed while the process lives.  hipMalloc is not allowed while the stream is being captured: a graph capture must have been
// preceded by an eager call of the same shape (GraphedForward warms up before it captures) -- otherwise the launcher falls back
// to the unsplit grid, which is always correct.
struct SplitKWs { int device; hipStream_t stream; float *ws; uint32_t *flags; size_t tiles; };
static std::mutex g_sk_mu;
static SplitKWs g_sk[64];
static int g_sk_n = 0;
static bool splitk_workspace(hipStream_t st, size_t tiles, float **ws, uint32_t **flags) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(g_sk_mu);
  SplitKWs *e = nullptr;
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk[i].device == dev && g_sk[i].stream == st) e = &g_sk[i];
  if (e != nullptr && e->tiles >= tiles) {
    *ws = e->ws;
    *flags = e->flags;
    return true;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;  // no allocation inside a capture
  if (e == nullptr) {
    if (g_sk_n >= 64) return false;
    e = &g_sk[g_sk_n];
    *e = SplitKWs{dev, st, nullptr, nullptr, 0};
  } else {
    (void)hipStreamSynchronize(st);  // the old buffers may still be in use by work queued on this stream
    (void)hipFree(e->ws);
    (void)hipFree(e->flags);
    e->ws = nullptr; e->flags = nullptr; e->tiles = 0;
  }
  float *w = nullptr;
  uint32_t *f = nullptr;
  if (hipMalloc((void **)&w, tiles * 2 * 4 * (128 * 64) * sizeof(float)) != hipSuccess ||
      hipMalloc((void **)&f, tiles * 2 * 4 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(f, 0, tiles * 2 * 4 * sizeof(uint32_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    if (w) (void)hipFree(w);
    if (f) (void)hipFree(f);
    return false;
  }
  e->ws = w; e->flags = f; e->tiles = tiles;
  if (e == &g_sk[g_sk_n]) ++g_sk_n;
  *ws = w;
  *flags = f;
  return true;
}

// 0: launched split; 1: not applicable here (the caller launches the unsplit grid); < 0: error
template <typename T, int ACT>
static int launch_pp_splitk(const GemmArgs &a, hipStream_t st) {
  GemmArgs b = a;
  b.ntm = (a.R + GB_BM - 1) / GB_BM;
  b.ntn = (a.N + 255) / 256;
  const size_t tiles = (size_t)b.E_loc * b.ntm * b.ntn;
  const int nk = a.K / GL_BK;
  // whole 256-row tiles only (no ragged variant), an even number of K-tiles with at least four per half (the steady-state loop),
  // a multiple of four tiles (so that the two workgroups of a tile are neighbours on one XCD in the work order), 16-byte rows
  if (a.row_counts != nullptr || a.R % GB_BM != 0 || (nk & 1) || nk < 8 || (tiles & 3) || a.rot_on) return 1;
  if ((a.ldd & 7) || (a.d_stride_e & 7) || (a.d_stride_w & 7) || (reinterpret_cast<uintptr_t>(a.D) & 15)) return 1;
  if (!splitk_workspace(st, tiles, &b.sk_ws, &b.sk_flags)) return 1;
  const size_t lds = (size_t)8 * 64 * EP_PITCH;
  auto kern = expert_gemm_pp_kernel<T, ACT, false, false, false, true>;
  if (!tutel_lds_optin((const void *)kern, lds)) return -1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(2 * tiles)), dim3(GB_THREADS), lds, st, b);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
}

template <typename T, int ACT>
static int launch_pp(const GemmArgs &a, hipStream_t st) {
  GemmArgs b = a;
  b.ntm = (a.R + GB_BM - 1) / GB_BM;
  b.ntn = (a.N + 255) / 256;
  // a last M-tile with >= 32 padding rows, or per-expert row counts: the variant that skips padded 32-row groups
  const int tail_rows = b.R % GB_BM;
  const bool ragged = b.row_counts != nullptr || (tail_rows != 0 && tail_rows <= GB_BM - 32);
  // bias fetched before the K loop (full tiles; +0.5-1.7 % on every MFMA-bound shape, profiles/r03_pp_bias_ab.json); 0 = after it, for A/B
  const bool early = tutel_get_option(TUTEL_OPT_GEMM_PERSIST) != 0;
  if (b.ntm == 1 && (long long)b.E_loc * b.ntn >= 256)
    return ragged ? launch_pp_cfg<T, ACT, true, true>(b, st) : (early ? launch_pp_cfg<T, ACT, true, false, true>(b, st) : launch_pp_cfg<T, ACT, true>(b, st));
  return ragged ? launch_pp_cfg<T, ACT, false, true>(b, st) : (early ? launch_pp_cfg<T, ACT, false, false, true>(b, st) : launch_pp_cfg<T, ACT, false>(b, st));
}

template <typename T, bool KM, int ACT, int NI, int NS = 2, bool BUF = false, int BM = GB_BM, bool FL = false>
static int launch_big(const GemmArgs &a, hipStream_t st) {
  GemmArgs b = a;
  b.ntm = (a.R + BM - 1) / BM;
  b.ntn = (a.N + NI * 64 - 1) / (NI * 64);
  const size_t lds_k = (size_t)NS * (BM / 128 + NI / 2) * GL_STAGE * 2 + (FL ? 16384 : 0), lds_e = BUF ? (size_t)(BM / 32) * 64 * (NI * 64 + 16) : 0;
  const size_t lds = lds_k > lds_e ? lds_k : lds_e;
  auto kern = expert_gemm_big_kernel<T, KM, ACT, NI, NS, BUF, BM, FL>;
  if (!tutel_lds_optin((const void *)kern, lds)) return -1;
  hipLaunchKernelGGL(kern, dim3(a.E_loc * b.ntm * b.ntn), dim3(BM * 2), lds, st, b);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
}

template <typename T, bool KM, int ACT>
static int launch_glds(const GemmArgs &a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((expert_gemm_glds_kernel<T, KM, ACT, true, true>), dim3(grid), dim3(GM_THREADS),
                     (size_t)4 * GL_STAGE * 2, st, a);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
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
static constexpr size_t gemm_lds_bytes(bool kmajor) {
  return (size_t)(2 * GM_BM * (64 + 8) + 2 * (kmajor ? GM_BN * (64 + 8) : 64 * GM_LDN)) * 2;
}

template <typename T, bool KM, int ACT>
static int launch_cfg(const GemmArgs &a, int grid, hipStream_t st) {
  const size_t lds = gemm_lds_bytes(KM);
  auto kern = expert_gemm_kernel<T, KM, ACT, true, true>;
  if (lds > 65536 && !tutel_lds_optin((const void *)kern, lds)) return -1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(GM_THREADS), lds, st, a);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_gemm");
  return 0;
}

// Kernel choice per weight layout, from on-hardware A/B at the headline shape and at R = 1024
// rows/expert (tools/gemm_sweep.py, MI355X, round 1):
//   k-major W (fc1): LDS-DMA kernel          126-132 us  (register-staged: 136-140 us)
//   n-major W (fc2): register-staged kernel  124-128 us  (LDS-DMA: 128-130 us; its tr-read image
//                                                         has residual bank conflicts)
// both with non-temporal weight loads and the rotated K order (rows are 4 KB apart in both layouts;
// without the per-block stagger all resident blocks sit at the same k offset: k-major 158 -> 143 us,
// n-major 142 -> 130 us when the weights really come from HBM, i.e. fc1/fc2 alternating as in the
// layer -- a warm Infinity Cache hides this, so A/B runs must alternate the two weight sets).  What did NOT pay (kept out of the tree, see git history): BK=128, single
// LDS buffer at 3 blocks/CU, prefetch distance 2 (two register sets), a 128x256 8-wave 3-stage
// LDS-DMA ring, a single-stage LDS-DMA variant at 4 blocks/CU.  Ablation: compute alone 69 us, loads+staging alone 80-115 us -- the remaining
// loss is phase serialisation inside a block, not DRAM (pure loads of the same pattern: 77-90 us).
// tutel_amd_set_option(TUTEL_OPT_GEMM_IMPL, 0|1) forces register-staged | LDS-DMA for A/B runs.
template <typename T, bool KM, int ACT>
static int launch_gemm(const GemmArgs &a, int grid, hipStream_t st) {
  const int impl = tutel_get_option(TUTEL_OPT_GEMM_IMPL), big = tutel_get_option(TUTEL_OPT_GEMM_TILE);
  const bool ring256_ok = KM && a.fits32 && a.N >= 256 && a.R <= GM_BM;
  const bool ring256 = ring256_ok && (impl == 4 || (impl < 0 && (long long)a.E_loc * ((a.N + 255) / 256) >= 256));
  // Fused location FIRST: a request for it (or the eligibility query, fl_loc == NULL) must never fall into one of the launches
  // below -- only the 128 x 256 ring kernel has the fused form (capacity <= 128 rows per expert), and a query launches nothing.
  // K >= 128: the prologue's `s_waitcnt vmcnt(16)` counts the weight DMA of TWO K-tiles behind the idx bytes (with one
  // K-tile only 8 follow and the wait would not cover them).
  if (a.fl_idx8 != nullptr) {
    if (!(ring256 && KM && big <= 0 && a.a_rows != nullptr && a.row_counts == nullptr && a.fl_n >= 1 && a.fl_n <= 15360 && a.E_loc <= 128 && a.K >= 2 * GL_BK)) {
      tutel_set_error("tutel_expert_gemm_gather_fl: this launch does not take the fused-location ring kernel");
      return TUTEL_AMD_ENOTSUP;
    }
    if (a.fl_loc == nullptr) return 0;  // eligibility query
    return launch_big<T, true, ACT, 4, 3, true, 128, KM>(a, st);   // (FL = KM: the n-major instantiations never reach this line)
  }
  // R > 128 rows per expert: the 256-row tiles (more flop per byte crossing L2 -> CU) -- provided the grid
  // still covers the chip: one such block occupies a CU, so with fewer than ~3/4 x 256 blocks CUs sit idle.
  // 256 x 256 first, 256 x 128 when only that fills the chip (a pipeline stage of the overlapped
  // all-to-all is half a GEMM), else the 128-tile kernels.  big = 1 forces 256 x 256, 2 / 3 force 256 x 128.
  if (a.N >= GM_BN && big != 0) {
    const long long mt256 = (long long)a.E_loc * ((a.R + GB_BM - 1) / GB_BM);
    const long long t256 = mt256 * ((a.N + 255) / 256), t128 = mt256 * ((a.N + 127) / 128);
    // more than one 128-row tile per expert (R > 128) already pays: the 128-tile kernels would stream every
    // weight tile once per M-tile (dropless capacity 157 at the headline shape: fc1 214 us vs 118 at R = 128)
    // Beside a co-running collective (a stage of the overlapped pipeline: RCCL's kernels hold whole CUs on the other stream) a grid of
    // one workgroup per CU turns into two rounds as soon as a few CUs are taken: the 256 x 128 ring kernel at 4 x 1024 x 2048^2 goes
    // from 42 us to 70 us with 16 CUs held and to 57 us beside a device copy, the 128-block 256 x 256 grid stays at 52 us
    // (tools/contention_probe.py, profiles/r03_contention.json).  So with the co-run hint the half-filled 256 x 256 grid wins.
    const bool corun_pp = tutel_gemm_corun() && big < 0 && a.R > GM_BM && t256 >= 96 && t256 < 192;
    // Round 5: 96 .. 191 tiles of 256 x 256 (half the chip) CAN run two workgroups per tile, each over half of K (expert_gemm_pp_kernel
    // <.., SPLITK>).  Built, correct (tests/test_ops_gpu.py::test_split_k_pingpong_kernel) and measured SLOWER than what it was to replace:
    // the stage GEMM 4 x 1024 x 2048^2 takes 52.6-58.6 us split against 40.8 us on the 256 x 128 ring and 50.5 us on the unsplit half-chip
    // grid -- the hand-over of 128 KB out + 128 KB in per workgroup through system-scope accesses costs more than the half K loop it saves
    // (profiles/r05_splitk_probe.json).  So it is opt-in only: TUTEL_OPT_GEMM_SPLITK = 1.
    if (KM && a.fits32 && a.mul == nullptr && a.R > GM_BM && t256 >= 96 && t256 < 192 && (big < 0 || big == 4) &&
        tutel_get_option(TUTEL_OPT_GEMM_SPLITK) == 1) {
      const int rc = launch_pp_splitk<T, ACT>(a, st);
      if (rc <= 0) return rc;
    }
    // What the same A/B did show: with 129 .. 191 tiles the 256 x 128 ring needs TWO rounds of workgroups (2 x t256 > 256 CUs) where the
    // unsplit 256 x 256 grid needs one at 1.5 x the flop per byte -- 5 x 1024 x 2048 x 4096: 124 us on the ring, 88.5 us on 160 ping-pong
    // tiles (695 -> 971 TFLOP/s).  At <= 128 tiles the ring's single round wins (40.8 vs 50.5 us at 128).
    const bool pp_one_round = big < 0 && a.R > GM_BM && t256 > 128 && t256 < 192;
    if (KM && a.fits32 && (big == 4 || corun_pp || pp_one_round || (big < 0 && a.R > GM_BM && t256 >= 192))) return launch_pp<T, ACT>(a, st);
    if (big == 1 || big == 4 || (big < 0 && a.R > GM_BM && t256 >= 192)) return launch_big<T, KM, ACT, 4>(a, st);
    // 256 x 128: a three-slot ring (3 x 48 KB of LDS) keeps two tiles in flight: +3-4 % over two slots on the
    // stage shapes it is chosen for (tools/stage_probe.py); big = 2 forces the two-slot form for A/B runs
    if (KM && big == 2) return launch_big<T, true, ACT, 2, 2>(a, st);
    if (KM && (big == 3 || (big < 0 && a.R > GM_BM && t128 >= 192)))
      return a.fits32 && tutel_get_option(TUTEL_OPT_GEMM_IMPL) != 2 ? launch_big<T, true, ACT, 2, 3, true>(a, st) : launch_big<T, true, ACT, 2, 3>(a, st);
  }
  // Round 4: all rows of an expert in one M-tile (R <= 128: weight streaming from HBM is the bound) and k-major weights -> the
  // 128 x 256 tile on a three-slot ring, one 4-wave block per CU: the token tile crosses L2 -> LDS once per 256 columns instead of
  // once per 128 and 96 KB of DMA are in flight per CU.  Headline shape (64 x 128 rows, 2048^2): fc1 112.4 -> 108.1 us, fc2
  // 111.9 -> 106.6 us, the forward 264.0 -> 258.4 us, same bits (profiles/r04_headline_ab.json).  Only when the grid still covers
  // the chip (one 144 KB block per CU): with fewer tiles the 128 x 128 kernel's twice as many blocks keep more CUs busy.
  // TUTEL_OPT_GEMM_IMPL = 4 forces it (where it applies), 1 forces the 128 x 128 LDS-DMA kernel.
  if (ring256) return launch_big<T, true, ACT, 4, 3, true, 128>(a, st);
  const bool use_dma = impl < 0 ? KM : (impl == 1 || impl == 4);
  if (use_dma) return launch_glds<T, KM, ACT>(a, grid, st);
  return launch_cfg<T, KM, ACT>(a, grid, st);
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

// Argument checks + the GemmArgs block of one grouped GEMM (shared with expert_ffn.hip, which runs two of them in one launch).
// 0: *out is filled; 1: nothing to do (E_loc == 0 or R == 0); < 0: error (tutel_amd_last_error).
int tutel_gemm_args(const void *A, int64_t a_stride_e, int64_t a_stride_w, int a_rows_per_w, int lda, const void *W, int w_kmajor,
                    int64_t w_stride_e, int ldw, const void *bias, int64_t bias_stride_e, void *D, int64_t d_stride_e, int64_t d_stride_w,
                    int d_rows_per_w, int ldd, int E_loc, int R, int N, int K, int dtype, int act, const int32_t *row_counts, int row_align,
                    const int32_t *a_rows, int a_rows_mod, const void *a_zero, const void *mul, const uint64_t *d_peer, int64_t d_peer_off,
                    const PeerCanary *d_can, const uint8_t *fl_idx8, int fl_n, int32_t *fl_loc, GemmArgs *out) {
  (void)w_kmajor; (void)act;
  TUTEL_REQUIRE(dtype == TUTEL_BF16 || dtype == TUTEL_F16, "tutel_amd_expert_gemm: dtype must be bf16 or fp16 (got %d)", dtype);
  TUTEL_REQUIRE(E_loc >= 0 && R >= 0 && N >= 1 && K >= 1, "tutel_amd_expert_gemm: bad sizes E_loc=%d R=%d N=%d K=%d", E_loc, R, N, K);
  TUTEL_REQUIRE(K % 64 == 0, "tutel_amd_expert_gemm: K=%d must be a multiple of 64", K);
  TUTEL_REQUIRE(N % 8 == 0, "tutel_amd_expert_gemm: N=%d must be a multiple of 8", N);
  if (E_loc == 0 || R == 0) return 1;
  TUTEL_REQUIRE(A && W && (D || d_peer), "tutel_amd_expert_gemm: null pointer");
  TUTEL_REQUIRE(d_peer == nullptr || (mul == nullptr && d_peer_off % 16 == 0), "tutel_amd_expert_gemm: peer stores exclude the gated form");
  TUTEL_REQUIRE(a_rows_per_w >= 1 && d_rows_per_w >= 1, "tutel_amd_expert_gemm: rows_per_w must be >= 1");
  TUTEL_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldd % 4 == 0 && a_stride_e % 8 == 0 && a_stride_w % 8 == 0 &&
                    w_stride_e % 8 == 0 && d_stride_e % 4 == 0 && d_stride_w % 4 == 0 && bias_stride_e % 4 == 0,
                "tutel_amd_expert_gemm: leading dimensions / strides must keep rows 16-byte aligned");
  TUTEL_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)D % 8) == 0 && ((uintptr_t)bias % 8) == 0,
                "tutel_amd_expert_gemm: pointers must be 16-byte aligned");
  TUTEL_REQUIRE(row_counts == nullptr || row_align >= 1, "tutel_amd_expert_gemm: row_align must be >= 1");

  GemmArgs &a = *out;
  a.A = A; a.a_stride_e = a_stride_e; a.a_stride_w = a_stride_w; a.a_rpw = a_rows_per_w; a.lda = lda;
  a.W = W; a.w_stride_e = w_stride_e; a.ldw = ldw;
  a.bias = bias; a.bias_stride_e = bias_stride_e;
  a.D = D; a.d_stride_e = d_stride_e; a.d_stride_w = d_stride_w; a.d_rpw = d_rows_per_w; a.ldd = ldd;
  a.E_loc = E_loc; a.R = R; a.N = N; a.K = K;
  a.row_counts = row_counts; a.row_align = row_align < 1 ? 1 : row_align;
  a.a_rows = a_rows; a.a_rows_mod = a_rows_mod; a.a_zero = a_zero;
  a.a_span_bytes = 0;
  bool fits32 = true;  // the ping-pong kernel addresses rows with 32-bit byte offsets from the operand bases
  {
    const long long span = a_rows != nullptr ? (long long)a_rows_mod * lda * 2
                                             : ((long long)((R - 1) / a_rows_per_w) * a_stride_w + (long long)a_rows_per_w * lda) * 2;
    fits32 = span < 0x7ffff000LL;
    a.a_span_bytes = (int)span;
  }
  fits32 = fits32 && ((long long)N * ldw + K) * 2 < 0xffffff00LL;
  a.fits32 = fits32;
  // K-tile rotation (each block starts its K loop at a different tile) is a property of the PROBLEM, not of the kernel
  // that runs it, so every kernel produces the same bits for a given problem:
  //   R < 256 rows per expert (weight streaming from HBM dominates, 4 KB-strided rows): ON -- without the stagger all
  //     resident blocks sit at the same k offset and hot-spot HBM channels (fc1 158 -> 143 us at the headline shape in
  //     round 1; dropless 64 x 160 rows: 134 us on, 153 off);
  //   R >= 256 (full 256-row tiles, operands re-read through L2): OFF -- blocks that share a token tile then walk it
  //     together and the later ones hit in L2 (ping-pong kernel 65.7 -> 62.6 us at 8 x 1024 x 2048 x 2048, 242 -> 224 at
  //     4096^2, 85 -> 79 at 32 x 256 rows).
  a.rot_on = R < GB_BM;
  a.sgather = tutel_get_option(TUTEL_OPT_GEMM_GATHER) != 0 && (long long)E_loc * R * 4 < 0x7fffffffLL;
  {  // store policy of the LDS epilogue; the descriptor form needs every byte offset inside an expert's output below 2^31
    const int ds = tutel_get_option(TUTEL_OPT_GEMM_STORE);
    const long long span = ((long long)((R - 1) / d_rows_per_w) * (d_stride_w < 0 ? -d_stride_w : d_stride_w) + (long long)(d_rows_per_w < R ? d_rows_per_w : R) * ldd + N) * 2;
    a.d_store = (ds == 0 || span >= 0x7fffffffLL) ? 0 : (ds == 2 ? 2 : 1);   // automatic: write-through
    if (d_peer != nullptr) {  // peer rows: write-through only (a non-temporal hint means nothing to a peer's memory), one source rank per store instruction
      const long long pspan = ((long long)E_loc * d_stride_e + (long long)d_rows_per_w * ldd + N) * 2;
      a.d_store = (ds != 0 && d_rows_per_w % 8 == 0 && pspan < 0x7fffffffLL) ? 1 : 0;
    }
  }
  a.mul = mul;
  a.fl_idx8 = fl_idx8; a.fl_n = fl_n; a.fl_loc = fl_loc;
  a.d_peer = d_peer; a.d_peer_off = d_peer_off;
  a.d_can = d_can != nullptr ? *d_can : PeerCanary{nullptr, 0, 0, 0};
  TUTEL_REQUIRE(((uintptr_t)mul % 8) == 0, "tutel_amd_expert_gemm_glu: gating operand must be 8-byte aligned");
  TUTEL_REQUIRE(a_rows == nullptr || (a_rows_mod >= 1 && a_zero != nullptr && ((uintptr_t)a_zero % 16) == 0),
                "tutel_amd_expert_gemm_gather: need a_rows_mod >= 1 and a 16-byte aligned zero row");
  a.ntm = (R + GM_BM - 1) / GM_BM;
  a.ntn = (N + GM_BN - 1) / GM_BN;
  return 0;
}

static int expert_gemm_impl(const void *A, int64_t a_stride_e, int64_t a_stride_w,
                                     int a_rows_per_w, int lda, const void *W, int w_kmajor,
                                     int64_t w_stride_e, int ldw, const void *bias,
                                     int64_t bias_stride_e, void *D, int64_t d_stride_e,
                                     int64_t d_stride_w, int d_rows_per_w, int ldd, int E_loc,
                                     int R, int N, int K, int dtype, int act,
                                     const int32_t *row_counts, int row_align,
                                     const int32_t *a_rows, int a_rows_mod, const void *a_zero,
                                     const void *mul, tutel_stream_t stream, const uint64_t *d_peer = nullptr,
                                     int64_t d_peer_off = 0, const PeerCanary *d_can = nullptr, const uint8_t *fl_idx8 = nullptr,
                                     int fl_n = 0, int32_t *fl_loc = nullptr) {
  GemmArgs a;
  const int brc = tutel_gemm_args(A, a_stride_e, a_stride_w, a_rows_per_w, lda, W, w_kmajor, w_stride_e, ldw, bias, bias_stride_e, D, d_stride_e,
                                  d_stride_w, d_rows_per_w, ldd, E_loc, R, N, K, dtype, act, row_counts, row_align, a_rows, a_rows_mod, a_zero, mul,
                                  d_peer, d_peer_off, d_can, fl_idx8, fl_n, fl_loc, &a);
  if (brc != 0) return brc < 0 ? brc : 0;
  long long grid_ll = (long long)E_loc * a.ntm * a.ntn;
  TUTEL_REQUIRE(grid_ll < 0x7fffffffLL, "tutel_amd_expert_gemm: grid too large");
  const int grid = (int)grid_ll;
  hipStream_t st = (hipStream_t)stream;
  // per-stage timing: the launch with a fused activation is fc1, the one without is fc2 (callers that know better --
  // the native pipeline -- set a hint)
  auto go = [&]() -> int {
    if (dtype == TUTEL_BF16)
      return w_kmajor ? launch_gemm_act<bf16_t, true>(a, act, grid, st) : launch_gemm_act<bf16_t, false>(a, act, grid, st);
    return w_kmajor ? launch_gemm_act<f16_t, true>(a, act, grid, st) : launch_gemm_act<f16_t, false>(a, act, grid, st);
  };
  if (fl_idx8 != nullptr && fl_loc == nullptr) return go();  // eligibility query of the fused-location path: nothing is launched, nothing timed
  StageScope stage(act != TUTEL_ACT_NONE ? TUTEL_STAGE_FC1 : TUTEL_STAGE_FC2, st);
  return go();
}

extern "C" int tutel_amd_expert_gemm(const void *A, int64_t a_stride_e, int64_t a_stride_w,
                                     int a_rows_per_w, int lda, const void *W, int w_kmajor,
                                     int64_t w_stride_e, int ldw, const void *bias,
                                     int64_t bias_stride_e, void *D, int64_t d_stride_e,
                                     int64_t d_stride_w, int d_rows_per_w, int ldd, int E_loc,
                                     int R, int N, int K, int dtype, int act,
                                     const int32_t *row_counts, int row_align,
                                     tutel_stream_t stream) {
  return expert_gemm_impl(A, a_stride_e, a_stride_w, a_rows_per_w, lda, W, w_kmajor, w_stride_e, ldw, bias,
                          bias_stride_e, D, d_stride_e, d_stride_w, d_rows_per_w, ldd, E_loc, R, N, K, dtype, act,
                          row_counts, row_align, nullptr, 0, nullptr, nullptr, stream);
}

int tutel_expert_gemm_peer(const void *A, int64_t a_stride_e, int64_t a_stride_w, int a_rows_per_w, int lda, const void *W,
                           int w_kmajor, int64_t w_stride_e, int ldw, const void *bias, int64_t bias_stride_e,
                           const uint64_t *d_peer, int64_t d_peer_off, int64_t d_stride_e, int d_rows_per_w, int ldd, int E_loc,
                           int R, int N, int K, int dtype, int act, const PeerCanary &can, hipStream_t st) {
  TUTEL_REQUIRE(d_peer != nullptr, "tutel_expert_gemm_peer: null peer table");
  return expert_gemm_impl(A, a_stride_e, a_stride_w, a_rows_per_w, lda, W, w_kmajor, w_stride_e, ldw, bias, bias_stride_e, nullptr,
                          d_stride_e, 0, d_rows_per_w, ldd, E_loc, R, N, K, dtype, act, nullptr, 1, nullptr, 0, nullptr, nullptr,
                          (tutel_stream_t)st, d_peer, d_peer_off, &can);
}

// tutel_amd_expert_gemm_gather with the locations computed INSIDE the launch (see the FL comment at expert_gemm_big_kernel): the
// kernel fills slot_map [E_loc * R] and loc [n] itself from idx8 [n] (n = k * T bytes, buffer padded to 16).  loc == NULL: only
// answers whether this shape takes the fused kernel (0) or not (TUTEL_AMD_ENOTSUP) -- nothing is launched.
int tutel_expert_gemm_gather_fl(const void *X, int ldx, int32_t *slot_map, int T, const void *zero_row, const void *W, int64_t w_stride_e,
                                int ldw, const void *bias, int64_t bias_stride_e, void *D, int64_t d_stride_e, int ldd, int E_loc, int R,
                                int N, int K, int dtype, int act, const uint8_t *idx8, int n, int32_t *loc, hipStream_t st) {
  TUTEL_REQUIRE(slot_map != nullptr && idx8 != nullptr && T >= 1 && ((uintptr_t)idx8 & 15) == 0, "tutel_expert_gemm_gather_fl: bad arguments");
  if (tutel_get_option(TUTEL_OPT_FUSED_LOCATION) == 0) return TUTEL_AMD_ENOTSUP;
  return expert_gemm_impl(X, 0, 0, R > 0 ? R : 1, ldx, W, 1, w_stride_e, ldw, bias, bias_stride_e, D, d_stride_e, 0, R > 0 ? R : 1, ldd, E_loc,
                          R, N, K, dtype, act, nullptr, 1, slot_map, T, zero_row, nullptr, (tutel_stream_t)st, nullptr, 0, nullptr, idx8, n, loc);
}

extern "C" int tutel_amd_expert_gemm_glu(const void *A, int64_t a_stride_e, int64_t a_stride_w,
                                         int a_rows_per_w, int lda, const void *W, int w_kmajor,
                                         int64_t w_stride_e, int ldw, const void *bias,
                                         int64_t bias_stride_e, const void *G, void *D, int64_t d_stride_e,
                                         int64_t d_stride_w, int d_rows_per_w, int ldd, int E_loc,
                                         int R, int N, int K, int dtype, int act,
                                         const int32_t *row_counts, int row_align,
                                         tutel_stream_t stream) {
  TUTEL_REQUIRE(G != nullptr || E_loc == 0 || R == 0, "tutel_amd_expert_gemm_glu: null gating operand");
  return expert_gemm_impl(A, a_stride_e, a_stride_w, a_rows_per_w, lda, W, w_kmajor, w_stride_e, ldw, bias,
                          bias_stride_e, D, d_stride_e, d_stride_w, d_rows_per_w, ldd, E_loc, R, N, K, dtype, act,
                          row_counts, row_align, nullptr, 0, nullptr, G, stream);
}

extern "C" int tutel_amd_expert_gemm_gather(const void *X, int ldx, const int32_t *slot_map, int T,
                                            const void *zero_row, const void *W, int w_kmajor,
                                            int64_t w_stride_e, int ldw, const void *bias,
                                            int64_t bias_stride_e, void *D, int64_t d_stride_e, int ldd,
                                            int E_loc, int R, int N, int K, int dtype, int act,
                                            const int32_t *row_counts, int row_align,
                                            tutel_stream_t stream) {
  TUTEL_REQUIRE(slot_map != nullptr && T >= 1, "tutel_amd_expert_gemm_gather: need a slot map and T >= 1");
  return expert_gemm_impl(X, 0, 0, R > 0 ? R : 1, ldx, W, w_kmajor, w_stride_e, ldw, bias, bias_stride_e, D,
                          d_stride_e, 0, R > 0 ? R : 1, ldd, E_loc, R, N, K, dtype, act, row_counts, row_align,
                          slot_map, T, zero_row, nullptr, stream);
}
