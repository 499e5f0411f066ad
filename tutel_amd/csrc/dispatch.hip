// dispatch.hip -- fast_encode / fast_decode / gate-grad for gfx950.
//
// These are permutations (HBM-bound byte movement, no MFMA):
//   encode : bucket-major.  One wave per bucket row (e*C + c): look the owning (choice, token)
//            up in slot_map, stream the token row with 16-byte loads/stores, or store zeros for
//            an empty row.  One pass writes every byte of [E*C, M] exactly once -- the reference
//            does torch.zeros + k token-major launches (fast_dispatch.py:26-28) in fp32.
//            Algorithmic bytes: (T + E*C) * M * s.
//   decode : token-major.  One wave per token: gather its k bucket rows, multiply by the gate
//            and sum IN THE REFERENCE'S ORDER in fp32 (product rounded, then added: the
//            reference materialises k fp32 temps and adds them, fast_dispatch.py:61-66), round
//            once.  Algorithmic bytes: (n_kept + T) * M * s.
// Rows are processed as 16-byte vectors (8 x bf16/fp16 or 4 x fp32 per lane, 1 KiB per wave
// instruction); rows whose byte length is not a multiple of 16 take the scalar tail path.
#include "common.h"
#include "routing_dev.h"

#define DP_THREADS 256
#define DP_WAVES 4
#define TUTEL_DECODE_DEFAULT 2   // non-temporal stores: -1..2 us of a 263 us forward (profiles/r03_routing_fused_and_decode_ab.txt)


__device__ __forceinline__ float load_gate(const void *g, int gate_dtype, size_t i) {
  if (gate_dtype == TUTEL_F32) return reinterpret_cast<const float *>(g)[i];
  if (gate_dtype == TUTEL_BF16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t *>(g)[i]);
  return (float)reinterpret_cast<const _Float16 *>(g)[i];
}

// -------------------------------------------------------------------------------------------
// encode
// -------------------------------------------------------------------------------------------
// an fp32 intermediate that must EXIST as an fp32 value before it is narrowed: without it hipcc folds (_Float16)(g * (float)h) of the
// per-element (model_dim % 8 != 0) paths into v_fma_mixlo_f16 -- the exact product rounded ONCE, to fp16 -- where the reference rounds
// the product to fp32 first and narrows that (it dispatches in fp32, fast_dispatch.py:94-96).  The two differ when the gate carries more
// than 11 significant bits: fp32 gates over fp16 rows, found by the mixed-dtype cases of tests/test_fuzz_gpu.py.  No instruction.
__device__ __forceinline__ float f32_value(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <typename T>
__global__ __launch_bounds__(DP_THREADS) void encode_kernel(const T *__restrict__ x,
                                                           const int32_t *__restrict__ slot_map,
                                                           const void *__restrict__ gates,
                                                           int gate_dtype, int Tn, int M,
                                                           int n_slots, int capacity, int num_experts,
                                                           int chunk_rows, int expert_slice, int ep_world,
                                                           T *__restrict__ out, EncodePeer peer, int wt) {
  constexpr int VN = Vec<T>::N;
  typedef __attribute__((ext_vector_type(4))) uint32_t enc_u32x4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * DP_WAVES + (threadIdx.x >> 6)));  // (uniform: the row's descriptor below lives in SGPRs)
  const int nwaves = gridDim.x * DP_WAVES;
  const int nvec = M / VN;  // full 16-byte vectors per row (rows are 16B aligned when M % VN == 0)
  const bool vec_ok = (M % VN) == 0;

  for (int slot = peer.slot0 + wave; slot < n_slots; slot += nwaves) {
    // output row `slot` in the requested bucket order -> row e*C + l of the plain slot map (the inverse of
    // the decode kernel's addressing: chunk-major [C/c][E][c] or expert-sliced [E_loc/s][W][s][C])
    int plain = slot;
    if (chunk_rows > 0) {
      const int per = num_experts * chunk_rows, ci = slot / per, rem = slot % per;
      plain = (rem / chunk_rows) * capacity + ci * chunk_rows + rem % chunk_rows;
    } else if (expert_slice > 0) {
      const int blk = slot / capacity, l = slot % capacity, j = blk % expert_slice, iw = blk / expert_slice;
      const int e = (iw % ep_world) * (num_experts / ep_world) + (iw / ep_world) * expert_slice + j;
      plain = e * capacity + l;
    }
    int q = slot_map[plain];  // wave-uniform
    q = __builtin_amdgcn_readfirstlane(q);
    T *dst = out + (size_t)slot * M;
    if (peer.tab != nullptr) {
      // peer stores (csrc/ep.hip, IPC transport): bucket row `slot` of the exchange layout [stage][dst rank][rows] is written
      // straight into the receive buffer of the rank that owns the expert, where an all-to-all would have delivered it:
      // block <my rank> of its [stage][src rank][rows] array.  peer.tab[w] = base of rank w's segment in THIS process.
      const int blk = slot / peer.rows, rem = slot % peer.rows, st = blk / ep_world, w = blk % ep_world;
      dst = reinterpret_cast<T *>(peer.tab[w] + peer.off) + ((size_t)(st * ep_world + peer.rank) * peer.rows + rem) * M;
    }
    // wt (TUTEL_OPT_GEMM_STORE, as the expert GEMMs' output tiles): the row leaves with write-through (sc0 sc1) buffer stores through a
    // descriptor over the row -- it streams out (over xGMI when dst is a peer's buffer) while the kernel runs, instead of staying dirty
    // in this XCD's L2 until it is evicted or the kernel ends.  Compiler-generated stores: the hazard wait states are the compiler's.
    const bool wts = wt != 0 && vec_ok;   // wave-uniform
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dst, 0, wts ? M * (int)sizeof(T) : 0, 0x00020000);
#define ENC_ST16(D, I, V)                                                                                      \
    do {                                                                                                       \
      if (wts) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(enc_u32x4, (V)), rs_o, (I) * 16, 0, 17); \
      else (D)[(I)] = (V);                                                                                     \
    } while (0)
    if (q < 0) {
      if (vec_ok) {
        vec16 z = {{0u, 0u, 0u, 0u}};
        vec16 *d = reinterpret_cast<vec16 *>(dst);
        for (int i = lane; i < nvec; i += 64) ENC_ST16(d, i, z);
      } else {
        for (int i = lane; i < M; i += 64) dst[i] = Elem<T>::from_f32(0.f);
      }
      continue;
    }
    const int t = q % Tn;
    const T *src = x + (size_t)t * M;
    if (gates == nullptr) {  // is_postscore=True: pure copy (x * 1.0 is exact)
      if (vec_ok) {
        const vec16 *s = reinterpret_cast<const vec16 *>(src);
        vec16 *d = reinterpret_cast<vec16 *>(dst);
        int i = lane;
        for (; i + 192 < nvec; i += 256) {  // 4 independent 16B loads in flight per lane
          vec16 a = s[i], b = s[i + 64], c = s[i + 128], e = s[i + 192];
          ENC_ST16(d, i, a); ENC_ST16(d, i + 64, b); ENC_ST16(d, i + 128, c); ENC_ST16(d, i + 192, e);
        }
        for (; i < nvec; i += 64) { vec16 a = s[i]; ENC_ST16(d, i, a); }
      } else {
        for (int i = lane; i < M; i += 64) dst[i] = src[i];
      }
    } else {
      const float g = load_gate(gates, gate_dtype, (size_t)q);
      if (vec_ok) {
        const vec16 *s = reinterpret_cast<const vec16 *>(src);
        vec16 *d = reinterpret_cast<vec16 *>(dst);
        for (int i = lane; i < nvec; i += 64) {
          vec16 v = s[i];
          float f[VN];
          Vec<T>::unpack(v, f);
#pragma unroll
          for (int u = 0; u < VN; ++u) f[u] = mul_rn(g, f[u]);
          Vec<T>::pack(f, v);
          ENC_ST16(d, i, v);
        }
      } else {
        for (int i = lane; i < M; i += 64) dst[i] = Elem<T>::from_f32(f32_value(mul_rn(g, Elem<T>::to_f32(src[i]))));
      }
    }
#undef ENC_ST16
  }
  if (peer.tab != nullptr) peer_canary_store(peer.tab, peer.can);  // IPC transport: epoch canaries behind the rows (common.h)
}

// -------------------------------------------------------------------------------------------
// decode
// -------------------------------------------------------------------------------------------
// SPLIT waves share one token (each takes 1/SPLIT of the row): more, shorter waves -- the whole grid is resident at once at the
// headline shape, so the kernel is one dependent chain (index loads -> row loads -> stores) and shorter per-wave chains overlap
// better.  NTS: the combined rows are written with non-temporal stores (they are not re-read by this layer).
template <typename T, int KMAX, int SPLIT, bool NTS>
__device__ __forceinline__ void decode_body(const T *__restrict__ buf, const int32_t *__restrict__ idx,
                                            const int32_t *__restrict__ loc, const void *__restrict__ gates,
                                            int gate_dtype, int Tn, int M, int k, int capacity, int num_experts,
                                            int chunk_rows, int expert_slice, int ep_world, T *__restrict__ out, int bid, int nblk) {
  constexpr int VN = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = bid * DP_WAVES + (threadIdx.x >> 6);
  const int nwaves = nblk * DP_WAVES;
  const int nvec = M / VN;
  const bool vec_ok = (M % VN) == 0;
  const int per = SPLIT == 1 ? nvec : (((nvec + SPLIT - 1) / SPLIT + 63) / 64 * 64);  // vectors per wave of a token

  for (int wt = wave; wt < Tn * SPLIT; wt += nwaves) {
    const int t = wt / SPLIT, part = wt % SPLIT;
    // per-choice row pointer (nullptr = dropped) and gate, wave-uniform
    const T *rows[KMAX];
    float g[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      rows[j] = nullptr;
      g[j] = 0.f;
      if (j < k) {
        int e = idx[(size_t)j * Tn + t], l = loc[(size_t)j * Tn + t];
        if (l < capacity && e >= 0 && l >= 0) {
          // bucket row: [E][C] (plain), chunk-major [C/c][E][c], or expert-sliced [E_loc/s][W][s][C]
          // (the two layouts of the overlapped all-to-all)
          if (expert_slice > 0) {
            const int e_loc = num_experts / ep_world, w = e / e_loc, el = e % e_loc;
            e = ((el / expert_slice) * ep_world + w) * expert_slice + el % expert_slice;
          }
          size_t r = (chunk_rows > 0) ? ((size_t)(l / chunk_rows) * num_experts + e) * chunk_rows + (l % chunk_rows)
                                      : (size_t)e * capacity + l;
          rows[j] = buf + r * M;
          g[j] = gates ? load_gate(gates, gate_dtype, (size_t)j * Tn + t) : 1.0f;
        }
      }
    }
    T *dst = out + (size_t)t * M;
    if (vec_ok) {
      vec16 *d = reinterpret_cast<vec16 *>(dst);
      // UNR vectors per lane per pass: all UNR * k loads are issued before the first use (8 x 16 B in flight per lane at
      // k = 2; two in flight left the kernel at 4.2 TB/s although its input was just written by fc2)
      constexpr int UNR = KMAX * SPLIT >= 8 ? 1 : (8 / (KMAX * SPLIT));
      // loads are unconditional (a dropped choice reads bucket row 0 and its value is discarded): a branch around each
      // load would make hipcc wait for every load separately.  KMAX is the exact k for k <= 8 (a KMAX of the
      // next power of two issued up to 78 % of its loads for nothing)
      const T *src[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) src[j] = rows[j] ? rows[j] : buf;
      auto combine = [&](const vec16 (&v)[KMAX], vec16 &o) {
        float acc[VN];
#pragma unroll
        for (int u = 0; u < VN; ++u) acc[u] = 0.f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          if (j >= k) break;
          float f[VN];
          if (rows[j]) {
            Vec<T>::unpack(v[j], f);
#pragma unroll
            for (int u = 0; u < VN; ++u) f[u] = mul_rn(g[j], f[u]);
          } else {
#pragma unroll
            for (int u = 0; u < VN; ++u) f[u] = 0.f;
          }
          if (j == 0) {
#pragma unroll
            for (int u = 0; u < VN; ++u) acc[u] = f[u];
          } else {
#pragma unroll
            for (int u = 0; u < VN; ++u) acc[u] = add_rn(acc[u], f[u]);
          }
        }
        Vec<T>::pack(acc, o);
      };
      auto put = [&](int i, const vec16 &o) {
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
        if (NTS) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, o), reinterpret_cast<u32x4_t *>(d + i));
        else d[i] = o;
      };
      const int hi = min(nvec, (part + 1) * per);
      int i = part * per + lane;
      for (; i + 64 * (UNR - 1) < hi; i += 64 * UNR) {
        vec16 v[UNR][KMAX];
#pragma unroll
        for (int q = 0; q < UNR; ++q)
#pragma unroll
          for (int j = 0; j < KMAX; ++j)
            v[q][j] = reinterpret_cast<const vec16 *>(src[j])[i + 64 * q];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          vec16 o;
          combine(v[q], o);
          put(i + 64 * q, o);
        }
      }
      for (; i < hi; i += 64) {
        vec16 v[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) v[j] = reinterpret_cast<const vec16 *>(src[j])[i];
        vec16 o;
        combine(v, o);
        put(i, o);
      }
    } else if (part == 0) {
      for (int i = lane; i < M; i += 64) {
        float acc = 0.f;
        for (int j = 0; j < k && j < KMAX; ++j) {
          float f = rows[j] ? f32_value(mul_rn(g[j], Elem<T>::to_f32(rows[j][i]))) : 0.f;
          acc = (j == 0) ? f : add_rn(acc, f);
        }
        dst[i] = Elem<T>::from_f32(f32_value(acc));
      }
    }
  }
}

template <typename T, int KMAX, int SPLIT, bool NTS>
__global__ __launch_bounds__(DP_THREADS) void decode_kernel(const T *__restrict__ buf, const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ loc, const void *__restrict__ gates,
                                                           int gate_dtype, int Tn, int M, int k, int capacity, int num_experts,
                                                           int chunk_rows, int expert_slice, int ep_world, T *__restrict__ out) {
  decode_body<T, KMAX, SPLIT, NTS>(buf, idx, loc, gates, gate_dtype, Tn, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, out,
                                   (int)blockIdx.x, (int)gridDim.x);
}

// decode + the routing "finish" in one launch (fused-location path, ep.hip): block 0 -- dispatched first: the finish is a longer
// dependent chain (~5 us) than one decode block -- computes dispatch_count, the maximum expert load and the gshard loss from the
// routing workspace, what block 0 of location_kernel does when that kernel runs, off the critical path (nothing on the device
// waits for those three results); the other blocks are the decode above.
template <typename T, int KMAX>
__global__ __launch_bounds__(DP_THREADS) void decode_fin_kernel(const T *__restrict__ buf, const int32_t *__restrict__ idx,
                                                               const int32_t *__restrict__ loc, const void *__restrict__ gates,
                                                               int gate_dtype, int Tn, int M, int k, int capacity, int num_experts,
                                                               T *__restrict__ out, RouteFinish fin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
  if (blockIdx.x == 0) {  // block-uniform
    route_finish_block(fin, fin_smem);
    return;
  }
  decode_body<T, KMAX, 1, true>(buf, idx, loc, gates, gate_dtype, Tn, M, k, capacity, num_experts, 0, 0, 1, out, (int)blockIdx.x - 1,
                                (int)gridDim.x - 1);
}

// any k (k > 16; upstream's ATen chain takes any top-k, fast_dispatch.py:145-148): the same arithmetic with a run-time loop over the
// choices -- row pointers are recomputed per choice instead of living in registers, loads are not batched.  Not a hot path.
template <typename T>
__global__ __launch_bounds__(DP_THREADS) void decode_anyk_kernel(const T *__restrict__ buf, const int32_t *__restrict__ idx,
                                                                const int32_t *__restrict__ loc, const void *__restrict__ gates,
                                                                int gate_dtype, int Tn, int M, int k, int capacity, int num_experts,
                                                                int chunk_rows, int expert_slice, int ep_world, T *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * DP_WAVES + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * DP_WAVES;
  for (int t = wave; t < Tn; t += nwaves) {
    T *dst = out + (size_t)t * M;
    for (int i = lane; i < M; i += 64) {
      float acc = 0.f;
      for (int j = 0; j < k; ++j) {
        int e = idx[(size_t)j * Tn + t];
        const int l = loc[(size_t)j * Tn + t];
        float f = 0.f;
        if (l < capacity && e >= 0 && l >= 0) {
          if (expert_slice > 0) {
            const int e_loc = num_experts / ep_world, w = e / e_loc, el = e % e_loc;
            e = ((el / expert_slice) * ep_world + w) * expert_slice + el % expert_slice;
          }
          const size_t r = (chunk_rows > 0) ? ((size_t)(l / chunk_rows) * num_experts + e) * chunk_rows + (l % chunk_rows)
                                            : (size_t)e * capacity + l;
          const float g = gates ? load_gate(gates, gate_dtype, (size_t)j * Tn + t) : 1.0f;
          f = f32_value(mul_rn(g, Elem<T>::to_f32(buf[r * M + i])));
        }
        acc = (j == 0) ? f : add_rn(acc, f);
      }
      dst[i] = Elem<T>::from_f32(f32_value(acc));
    }
  }
}

// -------------------------------------------------------------------------------------------
// gate gradient: one wave per (choice, token); fp32 accumulate, butterfly reduce.
// (The reference accumulates serially in the data dtype, custom_kernel.cpp:318-321; a parallel
// fp32 reduction differs in summation order only -- tolerance stated in the test.)
// -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(DP_THREADS) void gate_grad_kernel(const T *__restrict__ x,
                                                              const T *__restrict__ buf,
                                                              const int32_t *__restrict__ idx,
                                                              const int32_t *__restrict__ loc,
                                                              int Tn, int M, int k, int capacity,
                                                              float *__restrict__ ggate) {
  constexpr int VN = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * DP_WAVES + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * DP_WAVES;
  const int n = k * Tn;
  const int nvec = M / VN;
  const bool vec_ok = (M % VN) == 0;
  for (int q = wave; q < n; q += nwaves) {
    const int t = q % Tn;
    int e = idx[q], l = loc[q];
    float acc = 0.f;
    if (l < capacity && e >= 0 && l >= 0) {
      const T *row = buf + ((size_t)e * capacity + l) * M;
      const T *xr = x + (size_t)t * M;
      if (vec_ok) {
        for (int i = lane; i < nvec; i += 64) {
          float a[VN], b[VN];
          Vec<T>::unpack(reinterpret_cast<const vec16 *>(row)[i], a);
          Vec<T>::unpack(reinterpret_cast<const vec16 *>(xr)[i], b);
#pragma unroll
          for (int u = 0; u < VN; ++u) acc += a[u] * b[u];
        }
      } else {
        for (int i = lane; i < M; i += 64) acc += Elem<T>::to_f32(row[i]) * Elem<T>::to_f32(xr[i]);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) ggate[q] = acc;
  }
}

// -------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------
static inline int dp_grid(int rows) {
  int blocks = (rows + DP_WAVES - 1) / DP_WAVES;
  const int cap = 256 * 8;  // 256 CUs x 8 blocks: grid-stride beyond that
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

// internal (csrc/ep.hip): fast_encode with optional peer stores and a slot range [slot0, n_slots)
int tutel_encode_launch(const void *x, int dtype, const int32_t *slot_map, const void *gates, int gate_dtype, int T, int M,
                        int n_slots, int capacity, int num_experts, int chunk_rows, int expert_slice, int ep_world, void *out,
                        const EncodePeer &peer, hipStream_t st) {
  TUTEL_REQUIRE(dtype_ok(dtype), "tutel_amd_fast_encode: unsupported dtype %d", dtype);
  TUTEL_REQUIRE(chunk_rows >= 0 && expert_slice >= 0 && !(chunk_rows > 0 && expert_slice > 0), "tutel_amd_fast_encode: bad bucket order");
  if (chunk_rows > 0 || expert_slice > 0) {
    TUTEL_REQUIRE(num_experts >= 1 && capacity >= 1 && (long long)num_experts * capacity >= n_slots,
                  "tutel_amd_fast_encode: a bucket order needs n_slots <= num_experts * capacity (got %d, %d x %d)", n_slots, num_experts, capacity);
    TUTEL_REQUIRE(chunk_rows == 0 || capacity % chunk_rows == 0, "tutel_amd_fast_encode: chunk_rows=%d must divide capacity=%d", chunk_rows, capacity);
    TUTEL_REQUIRE(expert_slice == 0 || (ep_world >= 1 && num_experts % ep_world == 0 && (num_experts / ep_world) % expert_slice == 0),
                  "tutel_amd_fast_encode: expert_slice=%d must divide the local experts of each of %d ranks", expert_slice, ep_world);
  }
  TUTEL_REQUIRE(gates == nullptr || dtype_ok(gate_dtype), "tutel_amd_fast_encode: unsupported gate dtype %d", gate_dtype);
  TUTEL_REQUIRE(T >= 0 && M >= 1 && n_slots >= 0 && peer.slot0 >= 0, "tutel_amd_fast_encode: bad sizes T=%d M=%d n_slots=%d", T, M, n_slots);
  TUTEL_REQUIRE(peer.tab == nullptr || (peer.rows >= 1 && ep_world >= 1 && peer.rank >= 0 && peer.rank < ep_world),
                "tutel_amd_fast_encode: bad peer-store arguments");
  if (n_slots - peer.slot0 <= 0) return 0;
  TUTEL_REQUIRE(slot_map && (out || peer.tab) && (x || T == 0), "tutel_amd_fast_encode: null pointer");
  TUTEL_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && (peer.off % 16) == 0, "tutel_amd_fast_encode: x/out must be 16-byte aligned");
  StageScope stage(TUTEL_STAGE_ENCODE, st);
  int grid = dp_grid(n_slots - peer.slot0);
  int Tn = T > 0 ? T : 1;
  const int wt = tutel_get_option(TUTEL_OPT_GEMM_STORE) != 0 && (long long)M * 4 < 0x7fffffffLL ? 1 : 0;   // write-through row stores (automatic: on)
  if (dtype == TUTEL_F32)
    hipLaunchKernelGGL(encode_kernel<float>, dim3(grid), dim3(DP_THREADS), 0, st, (const float *)x, slot_map, gates, gate_dtype, Tn, M, n_slots, capacity, num_experts, chunk_rows, expert_slice, ep_world, (float *)out, peer, wt);
  else if (dtype == TUTEL_BF16)
    hipLaunchKernelGGL(encode_kernel<bf16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const bf16_t *)x, slot_map, gates, gate_dtype, Tn, M, n_slots, capacity, num_experts, chunk_rows, expert_slice, ep_world, (bf16_t *)out, peer, wt);
  else
    hipLaunchKernelGGL(encode_kernel<f16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const f16_t *)x, slot_map, gates, gate_dtype, Tn, M, n_slots, capacity, num_experts, chunk_rows, expert_slice, ep_world, (f16_t *)out, peer, wt);
  TUTEL_CHECK_LAUNCH("tutel_amd_fast_encode");
  return 0;
}

extern "C" int tutel_amd_fast_encode(const void *x, int dtype, const int32_t *slot_map,
                                     const void *gates, int gate_dtype, int T, int M, int n_slots,
                                     int capacity, int num_experts, int chunk_rows, int expert_slice,
                                     int ep_world, void *out, tutel_stream_t stream) {
  if (chunk_rows > 0 || expert_slice > 0)
    TUTEL_REQUIRE((long long)num_experts * capacity == n_slots,
                  "tutel_amd_fast_encode: a bucket order needs n_slots == num_experts * capacity (got %d, %d x %d)", n_slots, num_experts, capacity);
  EncodePeer none = {nullptr, 0, 0, 1, 0, {nullptr, 0, 0, 0}};
  return tutel_encode_launch(x, dtype, slot_map, gates, gate_dtype, T, M, n_slots, capacity, num_experts, chunk_rows, expert_slice,
                             ep_world, out, none, (hipStream_t)stream);
}

template <typename T, int SPLIT, bool NTS>
static void launch_decode_cfg(const void *buf, const int32_t *idx, const int32_t *loc, const void *gates,
                              int gate_dtype, int Tn, int M, int k, int capacity, int num_experts, int chunk_rows, int expert_slice,
                              int ep_world, void *out, hipStream_t st) {
  int grid = dp_grid(Tn * SPLIT);
#define DEC(KM) hipLaunchKernelGGL((decode_kernel<T, KM, SPLIT, NTS>), dim3(grid), dim3(DP_THREADS), 0, st, (const T *)buf, idx, loc, gates, gate_dtype, Tn, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, (T *)out)
  switch (k) {
    case 1: DEC(1); break;
    case 2: DEC(2); break;
    case 3: DEC(3); break;
    case 4: DEC(4); break;
    case 5: DEC(5); break;
    case 6: DEC(6); break;
    case 7: DEC(7); break;
    case 8: DEC(8); break;
    default:
      if (k <= 12) DEC(12);
      else DEC(16);
  }
#undef DEC
}

// TUTEL_OPT_DECODE (A/B on hardware, tools/decode_probe.py): -1 automatic; bit 0 = two waves per token, bit 1 = non-temporal stores
template <typename T>
static void launch_decode(const void *buf, const int32_t *idx, const int32_t *loc, const void *gates,
                          int gate_dtype, int Tn, int M, int k, int capacity, int num_experts, int chunk_rows, int expert_slice,
                          int ep_world, void *out, hipStream_t st) {
  int mode = tutel_get_option(TUTEL_OPT_DECODE);
  if (mode < 0) mode = TUTEL_DECODE_DEFAULT;
  const bool split = (mode & 1) && (size_t)M * sizeof(T) >= 2048;  // rows of at least two 1 KiB wave-loads
#define GO(S, N) launch_decode_cfg<T, S, N>(buf, idx, loc, gates, gate_dtype, Tn, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, out, st)
  if (split) { if (mode & 2) GO(2, true); else GO(2, false); }
  else { if (mode & 2) GO(1, true); else GO(1, false); }
#undef GO
}

extern "C" int tutel_amd_fast_decode(const void *buf, int dtype, const int32_t *idx,
                                     const int32_t *loc, const void *gates, int gate_dtype, int T,
                                     int M, int k, int capacity, int num_experts, int chunk_rows,
                                     int expert_slice, int ep_world, void *out, tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype_ok(dtype), "tutel_amd_fast_decode: unsupported dtype %d", dtype);
  TUTEL_REQUIRE(gates == nullptr || dtype_ok(gate_dtype), "tutel_amd_fast_decode: unsupported gate dtype %d", gate_dtype);
  TUTEL_REQUIRE(T >= 0 && M >= 1 && k >= 1 && capacity >= 0, "tutel_amd_fast_decode: bad sizes T=%d M=%d k=%d C=%d", T, M, k, capacity);
  TUTEL_REQUIRE(chunk_rows >= 0 && (chunk_rows == 0 || (num_experts >= 1 && capacity % chunk_rows == 0)),
                "tutel_amd_fast_decode: chunk_rows=%d must divide capacity=%d (num_experts=%d)", chunk_rows, capacity, num_experts);
  TUTEL_REQUIRE(expert_slice >= 0 && (expert_slice == 0 || (chunk_rows == 0 && ep_world >= 1 && num_experts >= 1 &&
                                                         num_experts % ep_world == 0 && (num_experts / ep_world) % expert_slice == 0)),
                "tutel_amd_fast_decode: expert_slice=%d must divide the %d local experts of each of %d ranks (and excludes chunk_rows)",
                expert_slice, ep_world > 0 ? num_experts / ep_world : 0, ep_world);
  if (T == 0) return 0;
  TUTEL_REQUIRE(idx && loc && out && (buf || capacity == 0), "tutel_amd_fast_decode: null pointer");
  if (capacity == 0 || buf == nullptr) {  // nothing was dispatched: every token combines to the zero vector
    hipError_t e = hipMemsetAsync(out, 0, (size_t)T * M * dtype_size(dtype), (hipStream_t)stream);
    TUTEL_REQUIRE(e == hipSuccess, "tutel_amd_fast_decode: memset failed: %s", hipGetErrorString(e));
    return 0;
  }
  TUTEL_REQUIRE(((uintptr_t)buf % 16) == 0 && ((uintptr_t)out % 16) == 0, "tutel_amd_fast_decode: buf/out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  StageScope stage(TUTEL_STAGE_DECODE, st);
  if (k > 16) {   // past the register-resident kernels' top-k: the run-time-k kernel
    const int grid = dp_grid(T);
    if (dtype == TUTEL_F32) hipLaunchKernelGGL(decode_anyk_kernel<float>, dim3(grid), dim3(DP_THREADS), 0, st, (const float *)buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, (float *)out);
    else if (dtype == TUTEL_BF16) hipLaunchKernelGGL(decode_anyk_kernel<bf16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const bf16_t *)buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, (bf16_t *)out);
    else hipLaunchKernelGGL(decode_anyk_kernel<f16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const f16_t *)buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, (f16_t *)out);
    TUTEL_CHECK_LAUNCH("tutel_amd_fast_decode");
    return 0;
  }
  if (dtype == TUTEL_F32) launch_decode<float>(buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, out, st);
  else if (dtype == TUTEL_BF16) launch_decode<bf16_t>(buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, out, st);
  else launch_decode<f16_t>(buf, idx, loc, gates, gate_dtype, T, M, k, capacity, num_experts, chunk_rows, expert_slice, ep_world, out, st);
  TUTEL_CHECK_LAUNCH("tutel_amd_fast_decode");
  return 0;
}

// internal (common.h): fast_decode of plain [E, C] buckets + the routing finish in ONE launch (k <= 8, bf16 / fp16)
int tutel_decode_finish_launch(const void *buf, int dtype, const int32_t *idx, const int32_t *loc, const void *gates, int gate_dtype, int T, int M,
                               int k, int capacity, int num_experts, void *out, const RouteFinish &fin, hipStream_t st) {
  TUTEL_REQUIRE((dtype == TUTEL_BF16 || dtype == TUTEL_F16) && k >= 1 && k <= 8 && T >= 1 && capacity >= 1 && buf && idx && loc && out,
                "tutel_decode_finish_launch: bad arguments");
  TUTEL_REQUIRE(((uintptr_t)buf % 16) == 0 && ((uintptr_t)out % 16) == 0, "tutel_decode_finish_launch: buf/out must be 16-byte aligned");
  StageScope stage(TUTEL_STAGE_DECODE, st);
  const int grid = dp_grid(T) + 1;
  const size_t lds = route_finish_lds(fin.E, fin.k);
#define DECF(TT, KM)                                                                                                              \
  do {                                                                                                                            \
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)decode_fin_kernel<TT, KM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((decode_fin_kernel<TT, KM>), dim3(grid), dim3(DP_THREADS), lds, st, (const TT *)buf, idx, loc, gates, gate_dtype, T, M, \
                       k, capacity, num_experts, (TT *)out, fin);                                                                  \
  } while (0)
#define DECK(TT)                                                                                                                  \
  switch (k) {                                                                                                                    \
    case 1: DECF(TT, 1); break;                                                                                                   \
    case 2: DECF(TT, 2); break;                                                                                                   \
    case 3: DECF(TT, 3); break;                                                                                                   \
    case 4: DECF(TT, 4); break;                                                                                                   \
    case 5: DECF(TT, 5); break;                                                                                                   \
    case 6: DECF(TT, 6); break;                                                                                                   \
    case 7: DECF(TT, 7); break;                                                                                                   \
    default: DECF(TT, 8);                                                                                                         \
  }
  if (dtype == TUTEL_BF16) { DECK(bf16_t); } else { DECK(f16_t); }
#undef DECK
#undef DECF
  TUTEL_CHECK_LAUNCH("tutel_decode_finish_launch");
  return 0;
}

extern "C" int tutel_amd_gate_grad(const void *x, const void *buf, int dtype, const int32_t *idx,
                                   const int32_t *loc, int T, int M, int k, int capacity,
                                   float *ggate, tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype_ok(dtype), "tutel_amd_gate_grad: unsupported dtype %d", dtype);
  TUTEL_REQUIRE(T >= 0 && M >= 1 && k >= 1 && capacity >= 0, "tutel_amd_gate_grad: bad sizes");
  if (T == 0) return 0;
  TUTEL_REQUIRE(x && idx && loc && ggate && (buf || capacity == 0), "tutel_amd_gate_grad: null pointer");
  TUTEL_REQUIRE(((uintptr_t)buf % 16) == 0 && ((uintptr_t)x % 16) == 0, "tutel_amd_gate_grad: x/buf must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  StageScope stage(TUTEL_STAGE_OTHER, st);
  int grid = dp_grid(k * T);
  if (dtype == TUTEL_F32)
    hipLaunchKernelGGL(gate_grad_kernel<float>, dim3(grid), dim3(DP_THREADS), 0, st, (const float *)x, (const float *)buf, idx, loc, T, M, k, capacity, ggate);
  else if (dtype == TUTEL_BF16)
    hipLaunchKernelGGL(gate_grad_kernel<bf16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const bf16_t *)x, (const bf16_t *)buf, idx, loc, T, M, k, capacity, ggate);
  else
    hipLaunchKernelGGL(gate_grad_kernel<f16_t>, dim3(grid), dim3(DP_THREADS), 0, st, (const f16_t *)x, (const f16_t *)buf, idx, loc, T, M, k, capacity, ggate);
  TUTEL_CHECK_LAUNCH("tutel_amd_gate_grad");
  return 0;
}

