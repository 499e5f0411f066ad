// routing.hip -- top-k gating + compute_location for gfx950.
//
// Reference behaviour restated (tutel/impls/fast_dispatch.py:143-178, losses.py:7-19,
// jit_kernels/gating.py:19-24, custom_kernel.cpp:822-872): softmax -> torch.topk -> k one-hot
// [T,E] int64 masks -> k column cumsums -> masked row sums.  That is ~35 ATen launches and
// k x 2 MiB of one-hot traffic for what is, per token, "which experts, and what is my stable
// rank among the tokens that chose the same expert".
//
// MI355X design (latency-bound, ~1 MB of traffic, so: few launches, wave64 primitives, no MFMA):
//   K1 gate_topk_kernel : one wave per token row, lane = expert (E/64 experts per lane);
//                         softmax by wave butterfly, top-k by k wave-argmax rounds with the
//                         (score desc, expert index asc) order; per-tile expert histograms in
//                         LDS -> ws; per-tile score column sums (for l_aux) -> ws.
//   K2 location_kernel  : block b = token tile b.  Prefix over the (<=128) tile histograms
//                         gives the tile's base per (choice, expert); inside the tile a wave
//                         ranks 64 tokens at a time with ballot/popcount over the distinct
//                         experts present (a counting-sort rank, no one-hot, no scan of [T,E]).
//                         Emits loc, the bucket->token slot map, dispatch_count, max count and
//                         the gshard loss.
// Index outputs are integers and bit-exact against the reference CPU path; gates follow its
// per-op rounding in the scores dtype.
#include "common.h"
#include "routing_dev.h"
#include "topk_ties.h"

#define GT_THREADS 1024  // gate_topk: 16 waves per tile, 4 interleaved tokens per wave
#define GT_WAVES 16
#define RT_MAX_TILES 128
#define RT_MAX_K 16
#define RT_MAX_E 4096
#define GT_COL_SLAB 1024  // experts per pass of the column-sum reduction through LDS (bounds the LDS footprint for large E)

static inline int rt_tile(int T) {
  int per = (T + 64 * RT_MAX_TILES - 1) / (64 * RT_MAX_TILES);
  if (per < 1) per = 1;
  return 64 * per;
}
static inline int rt_ntiles(int T) { return (T + rt_tile(T) - 1) / rt_tile(T); }

extern "C" size_t tutel_amd_routing_workspace_bytes(int T, int E, int k) {
  if (T <= 0 || E <= 0 || k <= 0) return 0;
  size_t nt = (size_t)rt_ntiles(T);
  return nt * ((size_t)k * E * sizeof(int32_t) + (size_t)E * sizeof(float));
}

// -------------------------------------------------------------------------------------------
// K1: softmax (optional) + top-k + tile histogram + tile column sums
// -------------------------------------------------------------------------------------------
template <typename T, int EPL, int GT_BATCH>
__global__ __launch_bounds__(GT_THREADS) void gate_topk_kernel(
    const T *__restrict__ in, int apply_softmax, int Tn, int E, int k, int normalize, int tile,
    T *__restrict__ scores_out, int32_t *__restrict__ idx, T *__restrict__ gates,
    int32_t *__restrict__ ws_hist, float *__restrict__ ws_colsum, int32_t *__restrict__ clear_map,
    int clear_n, int tie_mode) {
  using CT = typename Elem<T>::ct;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *s_hist = reinterpret_cast<int32_t *>(smem);             // [k][E]
  float *s_col = reinterpret_cast<float *>(smem) + (size_t)k * E;  // [GT_WAVES][min(E, GT_COL_SLAB)]
  // tie_mode (topk_ties.h) = the number of replay slots LDS holds, one row + one queue each: GT_WAVES (every wave its own) up to
  // ~1000 experts, fewer past that -- the waves wid % slots of a slot then take turns (s_lock).  The slots share their bytes with s_col,
  // which is only used after the token loop.
  const int tie_slots = tie_mode > 0 ? tie_mode : GT_WAVES, tie_slot = (int)(threadIdx.x >> 6) % tie_slots;
  CT *s_tv = reinterpret_cast<CT *>(smem + (((size_t)k * E * 4 + 7) & ~(size_t)7)) + (size_t)tie_slot * E;               // [slots][E]
  uint16_t *s_tp = reinterpret_cast<uint16_t *>(reinterpret_cast<CT *>(smem + (((size_t)k * E * 4 + 7) & ~(size_t)7)) + (size_t)tie_slots * E) +
                   (size_t)tie_slot * E;                                                                                 // [slots][E]
  __shared__ int s_lock[GT_WAVES];   // 1: a wave is replaying a row in this slot

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;
  const int t0 = b * tile, t1 = min(Tn, t0 + tile);

  // the bucket->token map of the NEXT kernel starts out empty (-1): cleared here, one slice per
  // block, instead of a separate fill launch (K2 runs strictly after this kernel on the stream)
  if (clear_map != nullptr) {
    const int per = (clear_n + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c0 = b * per, c1 = min(clear_n, c0 + per);
    for (int i = c0 + tid; i < c1; i += GT_THREADS) clear_map[i] = -1;
  }

  for (int i = tid; i < k * E; i += GT_THREADS) s_hist[i] = 0;
  if (tid < GT_WAVES) s_lock[tid] = 0;
  __syncthreads();

  float colacc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) colacc[j] = 0.f;

  // wave `wid` owns tokens t0 + wid + GT_WAVES*i.  GT_BATCH tokens are processed INTERLEAVED:
  // every cross-lane step below is issued for all of them before the next step, so the
  // ~36 dependent ds_bpermute round trips per token overlap instead of adding up.
  for (int tb = t0 + wid; tb < t1; tb += GT_WAVES * GT_BATCH) {
    CT v[GT_BATCH][EPL];
    bool live[GT_BATCH];
    int tt[GT_BATCH];
#pragma unroll
    for (int u = 0; u < GT_BATCH; ++u) {
      tt[u] = tb + u * GT_WAVES;
      live[u] = tt[u] < t1;  // wave-uniform
      const T *row = in + (size_t)min(tt[u], Tn - 1) * E;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        int e = lane + 64 * j;
        v[u][j] = (e < E) ? Elem<T>::to_f32(row[e]) : -INFINITY;
      }
    }
    if (apply_softmax) {
      CT m[GT_BATCH], s[GT_BATCH];
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u) {
        m[u] = -INFINITY;
#pragma unroll
        for (int j = 0; j < EPL; ++j) m[u] = ct_max(m[u], v[u][j]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < GT_BATCH; ++u) m[u] = ct_max(m[u], __shfl_xor(m[u], o, 64));
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u) {
        s[u] = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          int e = lane + 64 * j;
          v[u][j] = (e < E) ? ct_exp(v[u][j] - m[u]) : 0.f;
          s[u] += v[u][j];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < GT_BATCH; ++u) s[u] += __shfl_xor(s[u], o, 64);
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u)
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          int e = lane + 64 * j;
          if (e < E) {
            T r = Elem<T>::from_f32(v[u][j] / s[u]);
            v[u][j] = Elem<T>::to_f32(r);
            if (scores_out && live[u]) scores_out[(size_t)tt[u] * E + e] = r;
          } else {
            v[u][j] = -INFINITY;
          }
        }
    }
    unsigned long long nanm[GT_BATCH];  // bit j: this lane's j-th expert holds a NaN
#pragma unroll
    for (int u = 0; u < GT_BATCH; ++u) {
      nanm[u] = 0;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        int e = lane + 64 * j;
        if (e < E && live[u]) colacc[j] += (float)v[u][j];  // token order u = 0..3: deterministic
        if (v[u][j] != v[u][j]) { v[u][j] = -INFINITY; nanm[u] |= 1ull << j; }  // NaN sorts last here (tie_mode replays such rows: first, as in ATen)
      }
    }

    // k rounds of wave arg-max, order: score desc, expert index asc.  tie_mode: one more round finds the (k + 1)-th score -- two
    // equal neighbours among the k + 1 largest is the only way this order can differ from torch.topk's.
    unsigned long long taken[GT_BATCH];   // bit j: this lane's j-th expert was picked in an earlier round (EPL <= 64)
    CT myg[GT_BATCH], prev[GT_BATCH];
    int myidx[GT_BATCH];
    bool tied[GT_BATCH];
#pragma unroll
    for (int u = 0; u < GT_BATCH; ++u) { taken[u] = 0; myg[u] = 0; myidx[u] = -1; prev[u] = 0; tied[u] = false; }
    const int rounds = tie_mode ? k + 1 : k;
    for (int c = 0; c < rounds; ++c) {
      CT bv[GT_BATCH];
      int be[GT_BATCH];
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u) {
        bv[u] = -INFINITY;
        be[u] = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          int e = lane + 64 * j;
          bool ok = (e < E) && !((taken[u] >> j) & 1ull);
          if (ok && (v[u][j] > bv[u] || (v[u][j] == bv[u] && e < be[u]))) { bv[u] = v[u][j]; be[u] = e; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < GT_BATCH; ++u) {
          CT ov = __shfl_xor(bv[u], o, 64);
          int oe = __shfl_xor(be[u], o, 64);
          if (ov > bv[u] || (ov == bv[u] && oe < be[u])) { bv[u] = ov; be[u] = oe; }
        }
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u) {
        if (c > 0 && bv[u] == prev[u] && be[u] != 0x7fffffff) tied[u] = true;  // wave-uniform
        prev[u] = bv[u];
        if (c < k) {
          if ((be[u] & 63) == lane) taken[u] |= 1ull << (be[u] >> 6);
          if (lane == c) { myg[u] = bv[u]; myidx[u] = be[u]; }
        }
      }
    }
    if (tie_mode) {
#pragma unroll
      for (int u = 0; u < GT_BATCH; ++u) {
        if (__ballot(nanm[u] != 0ull) != 0ull) tied[u] = true;
        if (tied[u] && live[u]) {  // wave-uniform: lane 0 replays ATen's CPU top-k over the row
          const bool shared_slot = tie_slots < GT_WAVES;   // block-uniform
          if (shared_slot) {
            // the slot is another wave's too: taken with an LDS compare-and-swap by lane 0 (the holder never waits on anything inside,
            // so the spin ends), the slot's previous contents ordered before ours by the acquire
            if (lane == 0)
              while (atomicCAS(&s_lock[tie_slot], 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
          }
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            int e = lane + 64 * j;
            if (e < E) {
              s_tv[e] = ((nanm[u] >> j) & 1ull) ? (CT)NAN : v[u][j];
              s_tp[e] = (uint16_t)e;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if constexpr (EPL <= 4) {
            // up to 256 experts: the whole wave replays the row with the queue in registers -- expert e already sits in lane e & 63,
            // slot e >> 6, which IS the queue's layout (topk_ties.h, WaveQueue); the rank tables take the bytes behind the row's copy
            using Rep = typename TkRepOf<T>::type;
            WaveQueue<EPL, Rep> wq;
            wq.sel = reinterpret_cast<uint8_t *>(s_tp);   // [E] uint16 per wave = the 2 * E bytes of the two tables
            wq.seln = E;
#pragma unroll
            for (int j = 0; j < EPL; ++j) wq.r[j] = Rep::make(((nanm[u] >> j) & 1ull) ? (CT)NAN : v[u][j], lane + 64 * j);
            AtenTopk<WaveQueue<EPL, Rep>> tsw(wq);
            tsw.run(E, k);
            if (lane < k) {
              myidx[u] = Rep::id_of(wq.r[0].unpack());   // choice c: queue position c = lane c, slot 0
              myg[u] = s_tv[myidx[u]];
            }
          } else {
            if (lane == 0) {
              LdsQueue<CT, uint16_t> lq{s_tv, s_tp};
              AtenTopk<LdsQueue<CT, uint16_t>> ts(lq);
              ts.run(E, k);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < k) {
              myidx[u] = s_tp[lane];
              myg[u] = s_tv[myidx[u]];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (shared_slot) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's reads of the slot are done (lgkmcnt(0)) before it is handed on
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&s_lock[tie_slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    // gates: raw score, optionally normalised by clamp(((0+g0)+g1)+..., eps) in dtype T.
#pragma unroll
    for (int u = 0; u < GT_BATCH; ++u) {
      CT denom = __shfl(myg[u], 0, 64);
      for (int c = 1; c < k; ++c) denom = round_to<T>(denom + __shfl(myg[u], c, 64));
      if (lane < k && live[u]) {
        CT g = myg[u];
        if (normalize && k > 1) {
          CT d = ct_max(denom, (CT)Elem<T>::eps());
          if (denom != denom) d = denom;  // torch.clamp keeps NaN
          g = g / d;
        }
        gates[(size_t)lane * Tn + tt[u]] = Elem<T>::from_f32(g);
        idx[(size_t)lane * Tn + tt[u]] = myidx[u];
        atomicAdd(&s_hist[lane * E + myidx[u]], 1);
      }
    }
  }

  __syncthreads();
  for (int i = tid; i < k * E; i += GT_THREADS) ws_hist[(size_t)b * k * E + i] = s_hist[i];
  // per-expert column sums: the 16 waves' partial sums meet in LDS, one slab of experts at a time, added in wave order
  const int slab = E < GT_COL_SLAB ? E : GT_COL_SLAB;
  for (int e0 = 0; e0 < E; e0 += slab) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      int e = lane + 64 * j;
      if (e >= e0 && e < e0 + slab && e < E) s_col[wid * slab + (e - e0)] = colacc[j];
    }
    __syncthreads();
    for (int e = e0 + tid; e < e0 + slab && e < E; e += GT_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < GT_WAVES; ++w) s += s_col[w * slab + (e - e0)];
      ws_colsum[(size_t)b * E + e] = s;
    }
    __syncthreads();
  }
}

// -------------------------------------------------------------------------------------------
// tile histograms from an externally supplied idx[k,T] (hist_ready == 0 path)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RT_THREADS) void tile_hist_kernel(const int32_t *__restrict__ idx,
                                                              int Tn, int E, int k, int tile,
                                                              int32_t *__restrict__ ws_hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *s_hist = reinterpret_cast<int32_t *>(smem);
  const int tid = threadIdx.x, b = blockIdx.x;
  const int t0 = b * tile, t1 = min(Tn, t0 + tile);
  for (int i = tid; i < k * E; i += RT_THREADS) s_hist[i] = 0;
  __syncthreads();
  for (int j = 0; j < k; ++j)
    for (int t = t0 + tid; t < t1; t += RT_THREADS) {
      int e = idx[(size_t)j * Tn + t];
      if (e >= 0 && e < E) atomicAdd(&s_hist[j * E + e], 1);
    }
  __syncthreads();
  for (int i = tid; i < k * E; i += RT_THREADS) ws_hist[(size_t)b * k * E + i] = s_hist[i];
}

// -------------------------------------------------------------------------------------------
// K2: locations.  The three phases are device functions over NW waves.  (Round 3 also ran them from inside the top-k kernel
// behind a hand-rolled grid barrier; measured equal to two launches, and a grid barrier without a residency check is a hang
// waiting for a partitioned device, so round 4 removed it.)
// -------------------------------------------------------------------------------------------
// phase 3: stable rank inside the tile: a wave handles one choice, 64 tokens per step.  `lidx` (optional): the tile's expert
// ids in LDS, [k][t1 - t0 rounded up to the tile]; `e_first`: the wave's first 64 ids when the caller loaded them early.
template <int NW, bool COH = false>
__device__ __forceinline__ void loc_rank(int tid, int t0, int t1, int Tn, int E, int k, const int32_t *__restrict__ idx,
                                         const int32_t *lidx, int lidx_stride, int e_first, bool have_first, int32_t *s_cur,
                                         int32_t *__restrict__ loc, int capacity, int32_t *__restrict__ slot_map) {
  const int lane = tid & 63, wid = tid >> 6;
  int ebits = 0;
  while ((1 << ebits) < E) ++ebits;
  for (int j = wid; j < k; j += NW) {
    int32_t *cur = s_cur + j * E;
    for (int c0 = t0; c0 < t1; c0 += 64) {
      int t = c0 + lane;
      int e;
      if (have_first && j == wid && c0 == t0) e = e_first;
      else if (t >= t1) e = -1;
      else e = lidx != nullptr ? lidx[j * lidx_stride + (t - t0)] : idx[(size_t)j * Tn + t];
      bool valid = (e >= 0) && (e < E);
      // lanes holding the same expert: AND over the bits of the expert id of (ballot of that bit,
      // complemented where my bit is 0) -- ceil(log2 E) ballots instead of one loop iteration per
      // distinct expert present in the wave.
      unsigned long long same = __ballot(valid);
      for (int bit = 0; bit < ebits; ++bit) {
        const unsigned long long bb = __ballot(valid && ((e >> bit) & 1));
        same &= ((e >> bit) & 1) ? bb : ~bb;
      }
      const int rank = __popcll(same & ((1ull << lane) - 1ull));
      const int cnt = __popcll(same);
      const bool leader = valid && (__ffsll((long long)same) - 1 == lane);
      int base = valid ? cur[e] : 0;
      int l = base + rank;
      if (leader) cur[e] = base + cnt;
      if (t < t1) {
        loc[(size_t)j * Tn + t] = valid ? l : 0;
        if (slot_map != nullptr && valid && l < capacity) st_i32<COH>(slot_map + (size_t)e * capacity + l, j * Tn + t);
      }
    }
  }
}

__global__ __launch_bounds__(RT_THREADS) void location_kernel(
    const int32_t *__restrict__ idx, int Tn, int E, int k, int tile, int ntiles,
    const int32_t *__restrict__ ws_hist, const float *__restrict__ ws_colsum,
    int32_t *__restrict__ loc, int32_t *__restrict__ dispatch_count, int32_t *__restrict__ stats,
    void *__restrict__ l_aux, int l_aux_dtype, int capacity, int32_t *__restrict__ slot_map) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *s_cur = reinterpret_cast<int32_t *>(smem);  // [k][E] running absolute location
  int32_t *s_tot = s_cur + (size_t)k * E;              // [k][E] per-choice totals
  float *s_parts = reinterpret_cast<float *>(smem) + (size_t)2 * k * E;  // [parts][E], see launch
  __shared__ float s_red[RT_WAVES];
  __shared__ int s_redi[RT_WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;
  const int t0 = b * tile, t1 = min(Tn, t0 + tile);

  // 0. loads that depend on nothing computed here go out first, so that their round trip overlaps the one of step 1
  //    instead of following it: this wave's first 64 expert ids (step 3) and, in block 0, a thread's share of the
  //    per-tile score column sums (step 4).  Same values, same summation order as loading them in place.
  int e_first = -1;
  if (wid < k && t0 + lane < t1) e_first = idx[(size_t)wid * Tn + t0 + lane];
  const int cs_parts = (E >= RT_THREADS) ? 1 : (RT_THREADS / E);
  const int cs_per = (ntiles + cs_parts - 1) / cs_parts;
  const bool cs_early = b == 0 && l_aux != nullptr && cs_parts * E <= RT_THREADS;  // one (expert, part) per thread
  float cs_first[16];
  if (cs_early && tid < cs_parts * E) {
    const int e = tid % E, a = (tid / E) * cs_per, z = min(ntiles, a + cs_per);
#pragma unroll
    for (int u = 0; u < 16; ++u) cs_first[u] = (a + u < z) ? ws_colsum[(size_t)(a + u) * E + e] : 0.f;
  }
  loc_prefix<RT_WAVES>(tid, b, E, k, ntiles, ws_hist, s_cur, s_tot, dispatch_count);
  loc_rank<RT_WAVES>(tid, t0, t1, Tn, E, k, idx, nullptr, 0, e_first, true, s_cur, loc, capacity, slot_map);
  if (b == 0) loc_finish<RT_WAVES>(tid, Tn, E, k, ntiles, ws_colsum, s_tot, s_parts, s_red, s_redi, cs_first, cs_early, stats, l_aux, l_aux_dtype);
}

// -------------------------------------------------------------------------------------------
// K1, small-E variant (E <= 128): SIXTEEN lanes per token instead of a whole wave.  Lane q of a
// 16-lane row owns the contiguous expert slice [q*EPQ, (q+1)*EPQ) in registers (EPQ = ceil(E/16)
// <= 8), so softmax max/sum and the top-k arg-max need four row-local exchange steps (xor 1,2,4,8:
// DPP row operations, ALU latency) instead of six ds_bpermute round trips, and only EPQ elements
// of serial per-lane work.  1024 threads = 64 tokens = one location tile; 16 waves per CU hide the
// dependent-VALU latency that a 4-wave block exposes (measured: 9.2 us with 4 lanes/token and
// 256-thread blocks vs the numbers in DESIGN.md for this layout).  Score column sums go through
// an LDS tile and are summed per expert in token order (deterministic).  Outputs are identical,
// bit for bit, to gate_topk_kernel.
// -------------------------------------------------------------------------------------------
#define GQ_LPT 16
#define GQ_THREADS 1024

// (Round 4 tried the gate PROJECTION inside this kernel -- x @ wg^T on MFMA with the fragments streamed straight from global memory,
// 16 waves = 2 token halves x E/32 expert tiles x K slices, partials reduced through LDS: 38.5 us against 15.7 us for the library
// GEMM + this kernel (profiles/r04_headline_ab.json).  One workgroup per 64-token tile is 64 workgroups, and fragment loads of
// 32 bytes per row and instruction are bound by the L1 request rate, not by HBM; a coalesced version is the expert GEMM's LDS-DMA
// pipeline plus a split-K reduction across workgroups, i.e. the library's skinny GEMM again.  Removed.)
// EPQ consecutive elements of a row as ONE wide load (the row offset q * EPQ is a multiple of EPQ elements and E == 16 * EPQ, so the
// address is aligned to the pack).  Per-element loads behind `e < E` predicates compile to a branch + a 2- or 4-byte load each.
template <typename V, int N> struct alignas((sizeof(V) * N) > 16 ? 16 : (sizeof(V) * N)) GqPack { V v[N]; };
template <typename V, int N> __device__ __forceinline__ void gq_load(const V *p, V (&o)[N]) {
  const GqPack<V, N> pk = *reinterpret_cast<const GqPack<V, N> *>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) o[i] = pk.v[i];
}

template <typename T, int EPQ>
__global__ __launch_bounds__(GQ_THREADS) void gate_topk_quad_kernel(
    const T *__restrict__ in, int apply_softmax, int Tn, int E, int k, int normalize, int tile,
    T *__restrict__ scores_out, int32_t *__restrict__ idx, T *__restrict__ gates,
    int32_t *__restrict__ ws_hist, float *__restrict__ ws_colsum, int32_t *__restrict__ clear_map,
    int clear_n, const float *__restrict__ part, int nsplit, T *__restrict__ logits_out, uint8_t *__restrict__ idx8, int tie_mode) {
  using CT = typename Elem<T>::ct;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ES = GQ_LPT * EPQ + 1;                                    // padded row of the score tile
  int32_t *s_hist = reinterpret_cast<int32_t *>(smem);                // [k][E]
  float *s_sc = reinterpret_cast<float *>(smem) + (size_t)k * E;      // [64][ES]
  // tie_mode (topk_ties.h): rows with a NaN or a tie among their k + 1 largest scores are replayed through ATen's CPU top-k, one row at
  // a time by the whole wave (the queue in registers, one position per lane)
  CT *s_tv = reinterpret_cast<CT *>(smem + ((((size_t)k * E + (size_t)64 * ES) * 4 + 7) & ~(size_t)7));   // [64][E] the rows to replay
  uint8_t *s_sel = reinterpret_cast<uint8_t *>(s_tv + (size_t)64 * E) + (size_t)(threadIdx.x >> 6) * 256;   // [16 waves][256] rank -> position tables
  __shared__ int s_nlist[2];   // rows handed to the replay in this pass / the next one
  __shared__ int s_list[64];   // their token slots

  const int tid = threadIdx.x, q = tid & (GQ_LPT - 1), tl = tid / GQ_LPT;  // tl = token slot 0..63
  const int b = blockIdx.x;
  const int t0 = b * tile, t1 = min(Tn, t0 + tile);

  if (clear_map != nullptr) {
    const int per = (clear_n + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c0 = b * per, c1 = min(clear_n, c0 + per);
    for (int i = c0 + tid; i < c1; i += GQ_THREADS) clear_map[i] = -1;
  }
  for (int i = tid; i < k * E; i += GQ_THREADS) s_hist[i] = 0;
  if (tid < 2) s_nlist[tid] = 0;
  float colsum = 0.f;  // thread e < E accumulates column e over the tile, in token order
  __syncthreads();

  int pass = -1;
  for (int ts = t0; ts < t1; ts += 64) {
    ++pass;
    const int t = ts + tl;
    const bool live = t < t1;
    CT v[EPQ];
    if (part != nullptr) {
      // logits = the gate projection's split-K partial sums (gate_proj.hip: part[s][t][e], fp32), added in split order and
      // rounded ONCE to the logits dtype -- what a `T`-typed F.linear with fp32 accumulation returns (gates/top.py:20-22).
      // Four splits' loads are in flight together (clamped addresses, the add is what is conditional).
      const float *row = part + (size_t)min(t, Tn - 1) * E + q * EPQ;
      const size_t sstride = (size_t)Tn * E;
      const bool exact = E == GQ_LPT * EPQ;   // block-uniform
      int ej[EPQ];                            // ragged E: clamped element offsets, loads stay unconditional
#pragma unroll
      for (int j = 0; j < EPQ; ++j) ej[j] = min(q * EPQ + j, E - 1) - q * EPQ;
      float a[EPQ];
#pragma unroll
      for (int j = 0; j < EPQ; ++j) a[j] = 0.f;
#define GQ_PART_SUM(LOAD)                                                                          \
      for (int s0 = 0; s0 < nsplit; s0 += 4) {                                                     \
        float b[4][EPQ];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
          const float *rs = row + (size_t)min(s0 + i, nsplit - 1) * sstride;                       \
          LOAD;                                                                                    \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
          if (s0 + i < nsplit) {                                                                   \
            _Pragma("unroll") for (int j = 0; j < EPQ; ++j) a[j] = (s0 + i == 0) ? b[i][j] : a[j] + b[i][j]; \
          }                                                                                        \
      }
      // two copies of the loop: merged into one, hipcc if-converts the two load forms into per-element loads again
      if (exact) {
        GQ_PART_SUM((gq_load<float, EPQ>(rs, b[i])));
      } else {
        GQ_PART_SUM(_Pragma("unroll") for (int j = 0; j < EPQ; ++j) b[i][j] = rs[ej[j]]);
      }
#undef GQ_PART_SUM
#pragma unroll
      for (int j = 0; j < EPQ; ++j) {
        int e = q * EPQ + j;
        const T r = Elem<T>::from_f32((CT)a[j]);
        if (logits_out && live && e < E) logits_out[(size_t)t * E + e] = r;
        v[j] = (e < E) ? Elem<T>::to_f32(r) : -INFINITY;
      }
    } else {
      const T *row = in + (size_t)min(t, Tn - 1) * E + q * EPQ;
      T raw[EPQ];
      if (E == GQ_LPT * EPQ && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {   // block-uniform
        gq_load<T, EPQ>(row, raw);
      } else {
#pragma unroll
        for (int j = 0; j < EPQ; ++j) raw[j] = row[min(q * EPQ + j, E - 1) - q * EPQ];
      }
#pragma unroll
      for (int j = 0; j < EPQ; ++j) {
        int e = q * EPQ + j;
        v[j] = (e < E) ? Elem<T>::to_f32(raw[j]) : -INFINITY;
      }
    }
    if (apply_softmax) {
      CT m = -INFINITY;
#pragma unroll
      for (int j = 0; j < EPQ; ++j) m = ct_max(m, v[j]);
#pragma unroll
      for (int o = 1; o < GQ_LPT; o <<= 1) m = ct_max(m, __shfl_xor(m, o, 64));
      CT s = 0;
#pragma unroll
      for (int j = 0; j < EPQ; ++j) {
        int e = q * EPQ + j;
        v[j] = (e < E) ? ct_exp(v[j] - m) : CT(0);
        s += v[j];
      }
#pragma unroll
      for (int o = 1; o < GQ_LPT; o <<= 1) s += __shfl_xor(s, o, 64);  // fixed butterfly order
#pragma unroll
      for (int j = 0; j < EPQ; ++j) {
        int e = q * EPQ + j;
        if (e < E) {
          T r = Elem<T>::from_f32(v[j] / s);
          v[j] = Elem<T>::to_f32(r);
          if (scores_out && live) scores_out[(size_t)t * E + e] = r;
        } else {
          v[j] = -INFINITY;
        }
      }
    }
    // score tile -> LDS for the deterministic column sums
    uint32_t nanm = 0;  // bit j: element j is a NaN
#pragma unroll
    for (int j = 0; j < EPQ; ++j) {
      int e = q * EPQ + j;
      if (e < E) s_sc[tl * ES + e] = live ? (float)v[j] : 0.f;
      if (v[j] != v[j]) { v[j] = -INFINITY; nanm |= 1u << j; }  // NaN sorts last here (tie_mode: such rows are replayed below, where it sorts FIRST as in ATen)
    }

    // k rounds of arg-max over the row's 16 lanes, order: score desc, expert index asc; choice c is parked on lane c of the row
    // (k <= 16).  tie_mode: one more round finds the (k + 1)-th score -- two equal neighbours among the k + 1 largest is the only way
    // the order above can differ from torch.topk's.
    uint32_t taken = 0;
    CT myg = 0, prev = 0;
    int myidx = 0;
    bool tied = false;
    const int rounds = tie_mode ? k + 1 : k;  // block-uniform
    for (int c = 0; c < rounds; ++c) {
      CT bv = -INFINITY;
      int be = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < EPQ; ++j) {
        int e = q * EPQ + j;
        bool ok = (e < E) && !((taken >> j) & 1u);
        if (ok && (v[j] > bv || (v[j] == bv && e < be))) { bv = v[j]; be = e; }
      }
#pragma unroll
      for (int o = 1; o < GQ_LPT; o <<= 1) {
        CT ov = __shfl_xor(bv, o, 64);
        int oe = __shfl_xor(be, o, 64);
        if (ov > bv || (ov == bv && oe < be)) { bv = ov; be = oe; }
      }
      if (c > 0 && bv == prev && be != 0x7fffffff) tied = true;  // (be == 0x7fffffff: the probe round of k == E found nothing)
      prev = bv;
      if (c < k) {
        if (be / EPQ == q) taken |= 1u << (be - q * EPQ);
        if (c == q) { myg = bv; myidx = be; }
        // the expert ids leave as soon as they are known, their store latency under the next round's shuffles (a row that turns out
        // to be tied is stored again by its replay, after the block barrier that orders the two)
        if (q == 0 && live) {
          idx[(size_t)c * Tn + t] = be;
          if (idx8 != nullptr) idx8[(size_t)c * Tn + t] = (uint8_t)be;  // byte copy for the in-GEMM location scan (expert_gemm.hip, FL)
        }
      }
    }
    if (tie_mode) {
      const int rowsh = (int)(threadIdx.x & 48);  // first lane of this row inside the wave
      if ((__ballot(nanm != 0) >> rowsh) & 0xffffull) tied = true;
      if (tied && live) {
        // the row is handed to the replay below: its scores (NaNs restored) go to LDS, its token slot onto this pass's list
#pragma unroll
        for (int j = 0; j < EPQ; ++j) {
          int e = q * EPQ + j;
          if (e < E) s_tv[tl * E + e] = ((nanm >> j) & 1u) ? (CT)NAN : v[j];
        }
        if (q == 0) s_list[atomicAdd(&s_nlist[pass & 1], 1)] = tl;
      }
    }
    // gates: raw score, optionally normalised by clamp(((0+g0)+g1)+..., eps), every sum rounded in dtype T
    {
      const int row0 = (int)(threadIdx.x & 48);
      CT denom = __shfl(myg, row0, 64);
      for (int c = 1; c < k; ++c) denom = round_to<T>(denom + __shfl(myg, row0 + c, 64));
      if (q < k && live && !(tie_mode && tied)) {
        CT g = myg;
        if (normalize && k > 1) {
          CT d = ct_max(denom, (CT)Elem<T>::eps());
          if (denom != denom) d = denom;  // torch.clamp keeps NaN
          g = g / d;
        }
        gates[(size_t)q * Tn + t] = Elem<T>::from_f32(g);
        atomicAdd(&s_hist[q * E + myidx], 1);
      }
    }
    __syncthreads();
    if (tid < E) {
      float s = colsum;
      for (int r = 0; r < 64; ++r) s += s_sc[r * ES + tid];
      colsum = s;
    }
    // ---- the replay of the tied rows (topk_ties.h): one row per wave at a time, by the waves that have no column to sum -- the
    // column sums above are a 64-step dependent chain on the first E / 64 waves, so a block with up to ~15 tied rows (the headline
    // shape has 2.7 per block on average) replays them in the shadow of work that was on its critical path anyway
    if (tie_mode) {
      const int wv = (int)(threadIdx.x >> 6), nbusy = (E + 63) >> 6, wl = (int)(threadIdx.x & 63);
      constexpr int SL = (GQ_LPT * EPQ + 63) / 64;  // register slots per lane: queue position i = (lane i & 63, slot i >> 6)
      const int nrep = s_nlist[pass & 1];
      for (int ent = wv - nbusy; ent >= 0 && ent < nrep; ent += GQ_THREADS / 64 - nbusy) {  // wave-uniform
        const int trow = s_list[ent], tt = ts + trow;
        using Rep = typename TkRepOf<T>::type;   // 16-bit scores: an order-preserving key and the index packed into one register
        WaveQueue<SL, Rep> wq;
        wq.sel = s_sel;
        wq.seln = E;
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
          const int e = wl + 64 * sl;
          wq.r[sl] = Rep::make(e < E ? s_tv[trow * E + e] : (CT)0, e);
        }
        AtenTopk<WaveQueue<SL, Rep>> tk(wq);
        tk.run(E, k);
        TkElem<CT> res;
        res.id = Rep::id_of(wq.r[0].unpack());
        res.v = s_tv[trow * E + (wl < k ? res.id : 0)];   // the score itself comes from the row (the queue may hold only its key)
        // choice c sits at queue position c: lane c, slot 0 (k <= 16)
        CT denom = tk_readlane(res.v, 0);
        for (int c = 1; c < k; ++c) denom = round_to<T>(denom + tk_readlane(res.v, c));
        if (wl < k) {
          CT g = res.v;
          if (normalize && k > 1) {
            CT d = ct_max(denom, (CT)Elem<T>::eps());
            if (denom != denom) d = denom;  // torch.clamp keeps NaN
            g = g / d;
          }
          gates[(size_t)wl * Tn + tt] = Elem<T>::from_f32(g);
          idx[(size_t)wl * Tn + tt] = res.id;
          if (idx8 != nullptr) idx8[(size_t)wl * Tn + tt] = (uint8_t)res.id;
          atomicAdd(&s_hist[wl * E + res.id], 1);
        }
      }
      if (tid == 0) s_nlist[(pass + 1) & 1] = 0;  // the next pass's list (last used by the previous pass: its readers are past their barrier)
    }
    __syncthreads();
  }
  for (int i = tid; i < k * E; i += GQ_THREADS) ws_hist[(size_t)b * k * E + i] = s_hist[i];
  if (tid < E) ws_colsum[(size_t)b * E + tid] = colsum;
}

// -------------------------------------------------------------------------------------------
// slot map from arbitrary (idx, loc)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slot_map_kernel(const int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ loc, int n,
                                                       int E, int capacity,
                                                       int32_t *__restrict__ slot_map) {
  int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  int e = idx[q], l = loc[q];
  if (e >= 0 && e < E && l >= 0 && l < capacity) slot_map[(size_t)e * capacity + l] = q;
}

// -------------------------------------------------------------------------------------------
// fast_cumsum_sub_one: block = 64 columns, 16 waves = 16 row chunks, two passes (the second
// pass re-reads the chunk from L2).
// -------------------------------------------------------------------------------------------
#define CS_WAVES 16
__global__ __launch_bounds__(CS_WAVES * 64) void cumsum_kernel(const int32_t *__restrict__ in,
                                                              int32_t *__restrict__ out, int Tn,
                                                              int E) {
  __shared__ int32_t s_part[CS_WAVES][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int rows_per = (Tn + CS_WAVES - 1) / CS_WAVES;
  const int r0 = wid * rows_per, r1 = min(Tn, r0 + rows_per);
  int acc = 0;
  if (col < E)
    for (int r = r0; r < r1; ++r) acc += in[(size_t)r * E + col];
  s_part[wid][lane] = acc;
  __syncthreads();
  int run = -1;
  for (int w = 0; w < wid; ++w) run += s_part[w][lane];
  if (col < E)
    for (int r = r0; r < r1; ++r) {
      run += in[(size_t)r * E + col];
      out[(size_t)r * E + col] = run;
    }
}

// -------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------
template <typename T>
static int launch_gate_topk(const void *in, int apply_softmax, int Tn, int E, int k, int normalize,
                            void *scores_out, int32_t *idx, void *gates, void *ws,
                            int32_t *clear_map, int clear_n, hipStream_t st, const float *part = nullptr, int nsplit = 0,
                            void *logits_out = nullptr, uint8_t *idx8 = nullptr) {
  const int tile = rt_tile(Tn), nt = rt_ntiles(Tn);
  int32_t *ws_hist = (int32_t *)ws;
  float *ws_col = (float *)(ws_hist + (size_t)nt * k * E);
  using CTh = typename Elem<T>::ct;
  // TUTEL_OPT_TIE_RULE: 1 / automatic = equal scores come out in the order of the reference's CPU torch.topk (topk_ties.h), 0 = lowest
  // expert index first (rounds 1-5).  The replay needs one row + one queue in LDS per concurrently replayed token: every wave of the
  // wave-per-token kernel has its own up to ~1000 experts (fp32 scores), past that the kernel is told how many slots fit and its waves
  // share them (at the limits of this file -- 4096 experts, k * E = 8192, fp64 scores -- three slots of 40 KB).
  int tie_mode = tutel_get_option(TUTEL_OPT_TIE_RULE) != 0;
  size_t lds = (size_t)k * E * 4 + 8;
  {
    const size_t col = (size_t)GT_WAVES * (E < GT_COL_SLAB ? E : GT_COL_SLAB) * 4, slot = (size_t)E * (sizeof(CTh) + 2);
    const size_t room = (size_t)160 * 1024 - lds - 256;   // (256: the kernel's static LDS)
    int slots = (int)(room / slot);
    if (slots > GT_WAVES) slots = GT_WAVES;
    if (slots < 1) tie_mode = 0;                          // (cannot happen for E <= RT_MAX_E, k * E <= 8192)
    const size_t tie = tie_mode ? (size_t)slots * slot : 0;
    lds += tie > col ? tie : col;
    if (tie_mode && E > 128) tie_mode = slots;            // the wave-per-token kernel's tie_mode is its slot count
  }
  if (E <= 128) {
    const int epq = (E + GQ_LPT - 1) / GQ_LPT;                 // 1..8
    const int epq_t = epq <= 1 ? 1 : (epq <= 2 ? 2 : (epq <= 4 ? 4 : 8));
    const size_t lds_q = ((size_t)k * E + (size_t)64 * (GQ_LPT * epq_t + 1)) * 4 + 8 + (tie_mode ? (size_t)64 * E * sizeof(CTh) + 16 * 256 : 0);
#define GQ_LAUNCH(EPQ)                                                                         \
    do {                                                                                       \
      if (lds_q > 65536) {                                                                     \
        (void)hipFuncSetAttribute((const void *)gate_topk_quad_kernel<T, EPQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q); \
        (void)hipGetLastError();                                                               \
      }                                                                                        \
      hipLaunchKernelGGL((gate_topk_quad_kernel<T, EPQ>), dim3(nt), dim3(GQ_THREADS), lds_q, st,            \
                         (const T *)in, apply_softmax, Tn, E, k, normalize, tile, (T *)scores_out,          \
                         idx, (T *)gates, ws_hist, ws_col, clear_map, clear_n, part, nsplit, (T *)logits_out, idx8, tie_mode); \
    } while (0)
    if (epq_t == 1) GQ_LAUNCH(1);
    else if (epq_t == 2) GQ_LAUNCH(2);
    else if (epq_t == 4) GQ_LAUNCH(4);
    else GQ_LAUNCH(8);
#undef GQ_LAUNCH
    TUTEL_CHECK_LAUNCH("tutel_amd_gate_topk");
    return 0;
  }
  TUTEL_REQUIRE(part == nullptr && idx8 == nullptr, "tutel_amd_gate_topk_partials: E = %d is past the 128 experts the partial-sum form covers", E);
  const int epl = (E + 63) / 64;
#define GT_LAUNCH(EPL)                                                                          \
  do {                                                                                          \
    if (lds > 65536) {                                                                          \
      (void)hipFuncSetAttribute((const void *)gate_topk_kernel<T, EPL, (EPL <= 4 ? 4 : (EPL <= 8 ? 2 : 1))>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
      (void)hipGetLastError();                                                                  \
    }                                                                                           \
    hipLaunchKernelGGL((gate_topk_kernel<T, EPL, (EPL <= 4 ? 4 : (EPL <= 8 ? 2 : 1))>), dim3(nt), dim3(GT_THREADS), lds, st,         \
                       (const T *)in, apply_softmax, Tn, E, k, normalize, tile,                 \
                       (T *)scores_out, idx, (T *)gates, ws_hist, ws_col, clear_map, clear_n, tie_mode);  \
  } while (0)
  if (epl <= 1) GT_LAUNCH(1);
  else if (epl <= 2) GT_LAUNCH(2);
  else if (epl <= 4) GT_LAUNCH(4);
  else if (epl <= 8) GT_LAUNCH(8);
  else if (epl <= 16) GT_LAUNCH(16);
  else if (epl <= 32) GT_LAUNCH(32);
  else GT_LAUNCH(64);
#undef GT_LAUNCH
  TUTEL_CHECK_LAUNCH("tutel_amd_gate_topk");
  return 0;
}

extern "C" int tutel_amd_gate_topk(const void *in, int dtype, int apply_softmax, int T, int E,
                                   int k, int normalize_gate, void *scores_out, int32_t *idx,
                                   void *gates, void *ws, size_t ws_bytes, int32_t *clear_map,
                                   int clear_n, tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype_ok(dtype) || dtype == TUTEL_F64, "tutel_amd_gate_topk: unsupported dtype %d", dtype);
  TUTEL_REQUIRE(T >= 0 && E >= 1 && E <= RT_MAX_E, "tutel_amd_gate_topk: need 1 <= E <= %d (got %d)", RT_MAX_E, E);
  TUTEL_REQUIRE(k >= 1 && k <= RT_MAX_K && k <= E, "tutel_amd_gate_topk: need 1 <= k <= min(E,%d) (got k=%d, E=%d)", RT_MAX_K, k, E);
  TUTEL_REQUIRE((size_t)k * E <= 8192, "tutel_amd_gate_topk: k*E = %d exceeds 8192", k * E);
  if (T == 0) return 0;
  TUTEL_REQUIRE(in && idx && gates && ws, "tutel_amd_gate_topk: null pointer");
  TUTEL_REQUIRE(ws_bytes >= tutel_amd_routing_workspace_bytes(T, E, k), "tutel_amd_gate_topk: workspace too small");
  TUTEL_REQUIRE(clear_n >= 0 && (clear_map != nullptr || clear_n == 0), "tutel_amd_gate_topk: bad clear_map");
  if (clear_n == 0) clear_map = nullptr;
  hipStream_t st = (hipStream_t)stream;
  StageScope stage(TUTEL_STAGE_GATE_TOPK, st);
  if (dtype == TUTEL_F64) return launch_gate_topk<double>(in, apply_softmax, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st);
  if (dtype == TUTEL_F32) return launch_gate_topk<float>(in, apply_softmax, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st);
  if (dtype == TUTEL_BF16) return launch_gate_topk<bf16_t>(in, apply_softmax, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st);
  return launch_gate_topk<f16_t>(in, apply_softmax, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st);
}

// internal (common.h): where the top-k kernel left the per-tile histograms / column sums of a (T, E, k) problem inside `ws`
void tutel_route_finish_args(int T, int E, int k, void *ws, RouteFinish *out) {
  const int nt = rt_ntiles(T);
  out->Tn = T; out->E = E; out->k = k; out->ntiles = nt;
  out->ws_hist = (const int32_t *)ws;
  out->ws_colsum = (const float *)((const int32_t *)ws + (size_t)nt * k * E);
}

// internal (common.h): either form of the top-k launch (logits `in`, or split-K partial sums), plus the byte copy of idx that the
// fused-location expert GEMM scans.  Arguments are the callers' (ep.hip), already validated by the public entry points' rules.
int tutel_gate_topk_launch(const void *in, const float *partials, int splits, int dtype, int T, int E, int k, int normalize_gate,
                           void *logits_out, int32_t *idx, void *gates, void *ws, int32_t *clear_map, int clear_n, uint8_t *idx8,
                           hipStream_t st) {
  TUTEL_REQUIRE(E <= 128 || (partials == nullptr && idx8 == nullptr), "tutel_gate_topk_launch: E = %d is past the 16-lanes-per-token kernel", E);
  if (clear_n == 0) clear_map = nullptr;
  StageScope stage(TUTEL_STAGE_GATE_TOPK, st);
  const int sm = 1;
  if (dtype == TUTEL_F32) return launch_gate_topk<float>(in, sm, T, E, k, normalize_gate, nullptr, idx, gates, ws, clear_map, clear_n, st, partials, splits, logits_out, idx8);
  if (dtype == TUTEL_BF16) return launch_gate_topk<bf16_t>(in, sm, T, E, k, normalize_gate, nullptr, idx, gates, ws, clear_map, clear_n, st, partials, splits, logits_out, idx8);
  return launch_gate_topk<f16_t>(in, sm, T, E, k, normalize_gate, nullptr, idx, gates, ws, clear_map, clear_n, st, partials, splits, logits_out, idx8);
}

extern "C" int tutel_amd_gate_topk_partials(const float *partials, int splits, int dtype, int T, int E, int k,
                                            int normalize_gate, void *logits_out, void *scores_out, int32_t *idx,
                                            void *gates, void *ws, size_t ws_bytes, int32_t *clear_map, int clear_n,
                                            tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype == TUTEL_F16 || dtype == TUTEL_BF16, "tutel_amd_gate_topk_partials: the logits dtype must be fp16 / bf16 (got %d)", dtype);
  TUTEL_REQUIRE(T >= 0 && E >= 1 && E <= 128, "tutel_amd_gate_topk_partials: need 1 <= E <= 128 (got %d)", E);
  TUTEL_REQUIRE(k >= 1 && k <= RT_MAX_K && k <= E, "tutel_amd_gate_topk_partials: need 1 <= k <= min(E,%d) (got k=%d, E=%d)", RT_MAX_K, k, E);
  TUTEL_REQUIRE(splits >= 1 && splits <= 64, "tutel_amd_gate_topk_partials: bad split count %d", splits);
  if (T == 0) return 0;
  TUTEL_REQUIRE(partials && idx && gates && ws, "tutel_amd_gate_topk_partials: null pointer");
  TUTEL_REQUIRE(ws_bytes >= tutel_amd_routing_workspace_bytes(T, E, k), "tutel_amd_gate_topk_partials: workspace too small");
  TUTEL_REQUIRE(clear_n >= 0 && (clear_map != nullptr || clear_n == 0), "tutel_amd_gate_topk_partials: bad clear_map");
  if (clear_n == 0) clear_map = nullptr;
  hipStream_t st = (hipStream_t)stream;
  StageScope stage(TUTEL_STAGE_GATE_TOPK, st);
  if (dtype == TUTEL_BF16)
    return launch_gate_topk<bf16_t>(nullptr, 1, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st, partials, splits, logits_out);
  return launch_gate_topk<f16_t>(nullptr, 1, T, E, k, normalize_gate, scores_out, idx, gates, ws, clear_map, clear_n, st, partials, splits, logits_out);
}

extern "C" int tutel_amd_compute_location(const int32_t *idx, int T, int E, int k, int hist_ready,
                                          void *ws, size_t ws_bytes, int32_t *loc,
                                          int32_t *dispatch_count, int32_t *stats, void *l_aux,
                                          int l_aux_dtype, int capacity, int32_t *slot_map,
                                          int slot_map_cleared, tutel_stream_t stream) {
  TUTEL_REQUIRE(T >= 0 && E >= 1 && E <= RT_MAX_E, "tutel_amd_compute_location: need 1 <= E <= %d (got %d)", RT_MAX_E, E);
  TUTEL_REQUIRE(k >= 1 && k <= RT_MAX_K, "tutel_amd_compute_location: need 1 <= k <= %d (got %d)", RT_MAX_K, k);
  TUTEL_REQUIRE((size_t)k * E <= 8192, "tutel_amd_compute_location: k*E = %d exceeds 8192", k * E);
  TUTEL_REQUIRE(dispatch_count != nullptr, "tutel_amd_compute_location: dispatch_count is null");
  TUTEL_REQUIRE(l_aux == nullptr || dtype_ok(l_aux_dtype), "tutel_amd_compute_location: bad l_aux dtype %d", l_aux_dtype);
  TUTEL_REQUIRE((long long)k * T < 0x7fffffffLL, "tutel_amd_compute_location: k*T overflows int32");
  TUTEL_REQUIRE(hist_ready || l_aux == nullptr, "tutel_amd_compute_location: l_aux needs the column sums written by tutel_amd_gate_topk (hist_ready=1)");
  hipStream_t st = (hipStream_t)stream;
  if (T == 0) {
    (void)hipMemsetAsync(dispatch_count, 0, (size_t)E * 4, st);
    if (stats) (void)hipMemsetAsync(stats, 0, 4, st);
    if (l_aux) (void)hipMemsetAsync(l_aux, 0, dtype_size(l_aux_dtype), st);
    if (slot_map && capacity > 0) (void)hipMemsetAsync(slot_map, 0xFF, (size_t)E * capacity * 4, st);
    return 0;
  }
  TUTEL_REQUIRE(idx && loc && ws, "tutel_amd_compute_location: null pointer");
  TUTEL_REQUIRE(ws_bytes >= tutel_amd_routing_workspace_bytes(T, E, k), "tutel_amd_compute_location: workspace too small");
  const int tile = rt_tile(T), nt = rt_ntiles(T);
  int32_t *ws_hist = (int32_t *)ws;
  float *ws_col = (float *)(ws_hist + (size_t)nt * k * E);
  StageScope stage(TUTEL_STAGE_LOCATION, st);
  if (!hist_ready) {
    hipLaunchKernelGGL(tile_hist_kernel, dim3(nt), dim3(RT_THREADS), (size_t)k * E * 4, st, idx, T, E, k, tile, ws_hist);
    TUTEL_CHECK_LAUNCH("tutel_amd_compute_location(hist)");
  }
  if (slot_map != nullptr && capacity > 0) {
    if (!slot_map_cleared) {
      hipError_t e = hipMemsetAsync(slot_map, 0xFF, (size_t)E * capacity * 4, st);
      TUTEL_REQUIRE(e == hipSuccess, "tutel_amd_compute_location: memset failed: %s", hipGetErrorString(e));
    }
  } else {
    slot_map = nullptr;
    capacity = 0;
  }
  const size_t lds_loc = ((size_t)2 * k * E + (size_t)(E > RT_THREADS ? E : RT_THREADS)) * 4;
  if (lds_loc > 65536) {
    (void)hipFuncSetAttribute((const void *)location_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_loc);
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(location_kernel, dim3(nt), dim3(RT_THREADS), lds_loc, st, idx, T, E, k,
                     tile, nt, ws_hist, ws_col, loc, dispatch_count, stats, l_aux, l_aux_dtype, capacity, slot_map);
  TUTEL_CHECK_LAUNCH("tutel_amd_compute_location");
  return 0;
}

extern "C" int tutel_amd_slot_map(const int32_t *idx, const int32_t *loc, int T, int E, int k,
                                  int capacity, int32_t *slot_map, tutel_stream_t stream) {
  TUTEL_REQUIRE(E >= 1 && k >= 1 && capacity >= 0 && T >= 0, "tutel_amd_slot_map: bad sizes");
  TUTEL_REQUIRE((long long)k * T < 0x7fffffffLL, "tutel_amd_slot_map: k*T overflows int32");
  hipStream_t st = (hipStream_t)stream;
  if (capacity == 0) return 0;
  TUTEL_REQUIRE(slot_map != nullptr, "tutel_amd_slot_map: null slot_map");
  StageScope stage(TUTEL_STAGE_OTHER, st);
  hipError_t e = hipMemsetAsync(slot_map, 0xFF, (size_t)E * capacity * 4, st);
  TUTEL_REQUIRE(e == hipSuccess, "tutel_amd_slot_map: memset failed: %s", hipGetErrorString(e));
  int n = k * T;
  if (n == 0) return 0;
  TUTEL_REQUIRE(idx && loc, "tutel_amd_slot_map: null pointer");
  hipLaunchKernelGGL(slot_map_kernel, dim3((n + 255) / 256), dim3(256), 0, st, idx, loc, n, E, capacity, slot_map);
  TUTEL_CHECK_LAUNCH("tutel_amd_slot_map");
  return 0;
}

extern "C" int tutel_amd_cumsum_sub_one(const int32_t *mask, int32_t *out, int T, int E,
                                        tutel_stream_t stream) {
  TUTEL_REQUIRE(T >= 0 && E >= 0, "tutel_amd_cumsum_sub_one: bad sizes");
  if (T == 0 || E == 0) return 0;
  TUTEL_REQUIRE(mask && out, "tutel_amd_cumsum_sub_one: null pointer");
  hipLaunchKernelGGL(cumsum_kernel, dim3((E + 63) / 64), dim3(CS_WAVES * 64), 0, (hipStream_t)stream, mask, out, T, E);
  TUTEL_CHECK_LAUNCH("tutel_amd_cumsum_sub_one");
  return 0;
}
