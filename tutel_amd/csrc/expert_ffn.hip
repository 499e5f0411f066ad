// expert_ffn.hip -- fc1 -> activation -> fc2 of every local expert in ONE persistent launch (SURVEY 8a row a5, section 7 hard part 3).
//
//   y[e] = act(x[e] @ W1[e]^T + b1[e]) @ W2[e] + b2[e]          (tutel/experts/ffn.py:114-120: two bmm + bias adds + activation)
//
// VERDICT r5 item 2 asked for this kernel; round 6 built it, proved it bit-identical to the two launches it would replace -- and
// MEASURED IT SLOWER, so it is opt-in (TUTEL_OPT_FFN_FUSED = 1), not the default.  What it is:
//   * ONE workgroup per CU stays resident and takes work items from a ticket counter; an item is one (expert, 256-column tile) of fc1
//     or of fc2 -- the SAME tile function the two-launch path runs (gemm_big_tile, gemm_dev.h), the same k order: the same bits;
//   * work is queued per XCD: queue x (x = the hardware's XCC id of the workgroup) holds experts x, x + 8, ...: first their fc1
//     tiles, then their fc2 tiles, so the tiles of an expert share its rows in one L2.  A workgroup whose own queue is dry takes from
//     the next one that has work (a snapshot of all eight ticket words rides along with every ticket);
//   * an fc2 item of expert e waits until all fc1 tiles of e have published: fc1's output tile leaves with write-through stores
//     (round 5), the producing workgroup waits for them (s_waitcnt vmcnt(0), block barrier) and bumps a per-expert counter; the
//     consumer's first thread polls it.  Tickets are taken in order and only by workgroups that can no longer block, so whatever an
//     item waits for was claimed earlier by a RESIDENT workgroup: no deadlock whatever the residency (shared device, fewer CUs);
//   * the next ticket is asked for between an item's K loop and its epilogue, its round trip hidden behind the epilogue; the control
//     words live in a small device buffer per (device, stream) and are reset by the last workgroup to leave.
// What was measured at the headline shape (64 experts x 128 rows, 2048^2, bf16; profiles/r06_ffn_*):
//   * synchronisation is NOT the cost: per workgroup and launch, tickets 1.1-1.3 us, polls 1.0-1.2 us, publishes 0.7-1.0 us of ~212 us;
//   * the tile is: 49.7-53.2 us per item inside the persistent launch against (106.4 + 103.4) / 4 = 52.4 us per tile in the two launches,
//     dispatch and kernel ramp included -- the same.  The launch is bound by what HBM + Infinity Cache deliver (5.6-5.7 TB/s here; the
//     same tile streaming 2.1 GB of weights that cannot hit the cache sustains 4.9-5.0 TB/s, tools/scratch/tile_bench.hip), and while
//     one CU sits in its prologue or epilogue (5.3 + 6.9 us of a 50 us tile, same probe) the OTHER CUs take the bandwidth: the
//     "wave-less cycles" between and around the two launches were never idle HBM time, so removing them buys nothing;
//   * what the fusion adds is a tail: every workgroup runs exactly 4 items, the last ones end 201-217 us after the first start, and the
//     kernel lasts as long as the slowest -- 224 us under rocprofv3 against 209.8 us for the two launches (+ ~2 us between them); the
//     forward 0.2575 ms against 0.2460 ms (three alternating pairs of bench.py runs).
// (Two dead ends on the way, kept out of the tree: indexing the two argument blocks with a run-time index makes hipcc select per LANE and
// wrap every LDS-DMA issue of the K loop in a waterfall loop -- tiles 61-68 us; two inlined tile instantiations under a one-wave-per-SIMD
// register budget put the accumulators in AGPRs and run 3-4 % slower.)
// R <= 128 rows per expert (one M-tile: the HBM-bound weight-streaming regime), k-major weights for both GEMMs.
#include <mutex>

#include "gemm_dev.h"

#define FFN_NQ 8          // work queues = XCDs of the part
#define FFN_THREADS 256   // the 128 x 256 ring tile: 4 waves

struct FfnArgs {
  GemmArgs g[2];   // [0] fc1 (activation; optional row gather / fused location), [1] fc2
  uint32_t *ctl;   // [FFN_NQ] tickets | [E_loc] fc1 tiles published per expert | [1] workgroups that left
  int xcc_queue;   // 1: queue = hardware XCC id, 0: queue = blockIdx & 7 (TUTEL_OPT_FFN_FUSED = 2, for A/B)
  int prefetch;    // 1: the next ticket is asked for between an item's K loop and its epilogue; 0: after the item (A/B)
  unsigned long long *dbg;  // optional [grid][8] per-workgroup 100 MHz tick sums: ticket, poll, fc1 tile, publish, fc2 tile, total, items, first queue
};

// LDS: the ring (3 x 48 KB) [+ 16 KB of the fused-location scan]; the two ticket words sit in the last 16 bytes
template <bool FL> static constexpr size_t ffn_lds_bytes() { return (size_t)3 * 3 * GL_STAGE * 2 + (FL ? 16384 : 16); }

// The argument block of the item's GEMM, every field forced into SCALAR registers.  Indexing `a.g[ph]` with a run-time ph makes hipcc
// load both blocks and pick per LANE (v_cndmask): the buffer descriptors of the tile's LDS-DMA then sit in vector registers and every
// `buffer_load ... lds` of the K loop is wrapped in a waterfall loop (v_readfirstlane + compare + branch per issue: tiles took 61-68 us
// instead of 50).  v_readfirstlane tells the compiler what it could not prove: the value is wave-uniform.
template <typename V> __device__ __forceinline__ V ffn_uni(V v) {
  static_assert(sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 1, "scalar fields only");
  if constexpr (sizeof(V) == 8) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __builtin_bit_cast(V, ((unsigned long long)hi << 32) | lo);
  } else if constexpr (sizeof(V) == 4) {
    return __builtin_bit_cast(V, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  } else {
    return (V)__builtin_amdgcn_readfirstlane((int)v);
  }
}
__device__ __forceinline__ GemmArgs ffn_pick(const GemmArgs &a0, const GemmArgs &a1, int ph) {
  GemmArgs p;
#define FFN_SEL(f) p.f = ffn_uni(ph ? a1.f : a0.f)
  FFN_SEL(A); FFN_SEL(a_stride_e); FFN_SEL(a_stride_w); FFN_SEL(a_rpw); FFN_SEL(lda);
  FFN_SEL(W); FFN_SEL(w_stride_e); FFN_SEL(ldw);
  FFN_SEL(bias); FFN_SEL(bias_stride_e);
  FFN_SEL(D); FFN_SEL(d_stride_e); FFN_SEL(d_stride_w); FFN_SEL(d_rpw); FFN_SEL(ldd);
  FFN_SEL(E_loc); FFN_SEL(R); FFN_SEL(N); FFN_SEL(K);
  FFN_SEL(row_counts); FFN_SEL(row_align);
  FFN_SEL(a_rows); FFN_SEL(a_rows_mod); FFN_SEL(a_zero); FFN_SEL(a_span_bytes);
  FFN_SEL(fits32); FFN_SEL(rot_on); FFN_SEL(sgather); FFN_SEL(d_store);
  FFN_SEL(fl_idx8); FFN_SEL(fl_n); FFN_SEL(fl_loc);
  FFN_SEL(ntm); FFN_SEL(ntn); FFN_SEL(act_rt);
#undef FFN_SEL
  p.mul = nullptr; p.d_peer = nullptr; p.d_peer_off = 0;          // (ffn_covers: neither the gated form nor peer stores come here)
  p.d_can = PeerCanary{nullptr, 0, 0, 0};
  p.sk_ws = nullptr; p.sk_flags = nullptr;
  return p;
}

// items of queue q: experts q, q + 8, ... -- first all their fc1 tiles, then all their fc2 tiles
__device__ __forceinline__ int ffn_queue_items(int q, int E, int nt1, int nt2, int *n1) {
  const int ne = q < E ? (E - q + FFN_NQ - 1) / FFN_NQ : 0;
  *n1 = ne * nt1;
  return ne * (nt1 + nt2);
}

// The next ticket is asked for INSIDE the current item, between its K loop and its epilogue (gemm_big_tile's `tail` hook): the returning
// atomic's round trip (1.1-1.3 us under a weight stream, guide row "dequeue") hides behind the epilogue.  Asked for there, never earlier:
// an item past its K loop cannot block any more, so a claimed ticket is always held by a workgroup that will get to it.  The same hook
// snapshots all eight ticket words (lanes 0-7), so that a workgroup whose queue has run dry knows without another round trip whether any
// queue has work left (round 6, first version: one atomic + barrier per item in front of it and eight failing ones at the end = 15 us
// of a 237 us launch).
struct FfnTail {
  uint32_t *tick;
  int q, tid;
  int *nxt, *snap;
  int on;
  __device__ __forceinline__ void operator()() const {
    if (!on) return;
    if (tid < FFN_NQ) *snap = (int)__hip_atomic_load(tick + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) *nxt = (int)__hip_atomic_fetch_add(tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

template <typename T, bool FL>
__global__ __launch_bounds__(FFN_THREADS, 2) void expert_ffn_kernel(FfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  volatile int *s_tk = reinterpret_cast<volatile int *>(smem + ffn_lds_bytes<FL>() - 16);  // [2][2]: ticket, queues-with-work mask (past the 15888 bytes the FL scan uses)
  const int tid = threadIdx.x;
  const int E = a.g[0].E_loc, nt1 = a.g[0].ntn, nt2 = a.g[1].ntn;
  uint32_t *tick = a.ctl, *done = a.ctl + FFN_NQ, *left = done + E;
  int q = (int)(blockIdx.x & (FFN_NQ - 1));
  if (a.xcc_queue) q = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (FFN_NQ - 1));  // hwreg(HW_REG_XCC_ID, 0, 4)
  const int q0 = q;
  long long d_tk = 0, d_poll = 0, d_t1 = 0, d_pub = 0, d_t2 = 0, d_items = 0;
  const long long d_begin = wall_clock64();
  // the first ticket: nothing to hide it behind
  int nxt = 0, snap = 0;
  if (tid == 0) nxt = (int)__hip_atomic_fetch_add(tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int it = 0, mask_valid = 0;
  for (;;) {
    // hand the ticket (and the snapshot of which queues still have work) to the whole workgroup.  The barrier also separates the previous
    // item's epilogue (LDS staging) from this item's first DMA into LDS.  Two LDS slots in turn: a wave is never more than one barrier
    // behind thread 0.
    const long long c0 = wall_clock64();
    {
      int my_n1;
      const bool has = tid < FFN_NQ && snap < ffn_queue_items(tid, E, nt1, nt2, &my_n1);
      const unsigned long long hm = __ballot(has);   // (wave 0 holds lanes 0-7; other waves write nothing)
      if (tid == 0) {
        s_tk[(it & 1) * 2] = nxt;
        s_tk[(it & 1) * 2 + 1] = mask_valid ? (int)(hm & 0xff) : 0xff;
      }
    }
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(s_tk[(it & 1) * 2]);
    const int qmask = __builtin_amdgcn_readfirstlane(s_tk[(it & 1) * 2 + 1]);
    ++it;
    int n1;
    const int total = ffn_queue_items(q, E, nt1, nt2, &n1);
    const long long c1 = wall_clock64();
    d_tk += c1 - c0;
    if (t >= total) {
      // this queue is dry: another one that the snapshot says has work (the snapshot can only err towards "has work"), else done
      const int others = qmask & ~(1 << q);
      if (others == 0) break;
      const int rot = ((others >> q) | (others << (FFN_NQ - q))) & 0xff;   // bit i: queue q + i
      q = (q + __builtin_ctz(rot)) & (FFN_NQ - 1);
      if (tid == 0) nxt = (int)__hip_atomic_fetch_add(tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < FFN_NQ) snap = (int)__hip_atomic_load(tick + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mask_valid = 1;
      continue;
    }
    ++d_items;
    const FfnTail tail{tick, q, tid, &nxt, &snap, a.prefetch};
    mask_valid = 1;
    // ONE instantiation of the tile runs both kinds of item (the activation is a block-uniform switch in its epilogue, the fused
    // location a run-time flag of the FL form): half the code, and the register allocation of the K loop stays that of the
    // one-tile-per-workgroup kernel (two inlined copies cost 25 % of the tile time: 61-68 us instead of 50)
    const int ph = __builtin_amdgcn_readfirstlane(t < n1 ? 0 : 1);
    const int u = ph ? t - n1 : t, ntp = ph ? nt2 : nt1;
    const int e = __builtin_amdgcn_readfirstlane(q + (u / ntp) * FFN_NQ), nt = __builtin_amdgcn_readfirstlane(u % ntp);
    long long c2 = c1;
    if (ph) {
      if (tid == 0) {
        const long long t0 = wall_clock64();
        while ((int)__hip_atomic_load(done + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nt1) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > 400000000LL) __builtin_trap();  // 4 s of the 100 MHz clock: cannot happen (tickets are taken in order)
        }
      }
      __syncthreads();
      c2 = wall_clock64();
      d_poll += c2 - c1;
    }
    const GemmArgs p = ffn_pick(a.g[0], a.g[1], ph);
    gemm_big_tile<T, true, GEMM_ACT_RUNTIME, 4, 3, true, 128, FL>(p, e, 0, nt, smem, tail);
    const long long c3 = wall_clock64();
    if (a.dbg != nullptr && tid == 0 && d_items <= 8) {
      unsigned long long *tl = a.dbg + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + (d_items - 1) * 2;
      tl[0] = (unsigned long long)c2; tl[1] = (unsigned long long)c3;
    }
    if (ph == 0) {
      // publish: every write-through store of this workgroup has completed, then the expert's counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(done + e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d_t1 += c3 - c2;
      d_pub += wall_clock64() - c3;
    } else {
      d_t2 += c3 - c2;
    }
    if (!a.prefetch) {
      const FfnTail late{tick, q, tid, &nxt, &snap, 1};
      late();
    }
  }
  if (a.dbg != nullptr && tid == 0) {
    unsigned long long *d = a.dbg + (size_t)blockIdx.x * 8;
    d[0] = d_tk; d[1] = d_poll; d[2] = d_t1; d[3] = d_pub; d[4] = d_t2; d[5] = wall_clock64() - d_begin; d[6] = d_items; d[7] = q0;
  }
  // the last workgroup to leave resets the control words for the next launch on this stream
  if (tid == 0) {
    const uint32_t n = __hip_atomic_fetch_add(left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == gridDim.x - 1)
      for (int i = 0; i < FFN_NQ + E + 1; ++i) __hip_atomic_store(a.ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- control words: one buffer per (device, stream), zeroed once, reset by every launch itself ------------------------------------
struct FfnCtl { int device; hipStream_t stream; uint32_t *ctl; int experts; };
static std::mutex g_ctl_mu;
static FfnCtl g_ctl[64];
static int g_ctl_n = 0;
static uint32_t *ffn_ctl(hipStream_t st, int E_loc) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_ctl_mu);
  FfnCtl *e = nullptr;
  for (int i = 0; i < g_ctl_n; ++i)
    if (g_ctl[i].device == dev && g_ctl[i].stream == st) e = &g_ctl[i];
  if (e != nullptr && e->experts >= E_loc) return e->ctl;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;  // no allocation inside a capture
  if (e == nullptr && g_ctl_n >= 64) return nullptr;
  const int cap = E_loc < 256 ? 256 : E_loc;
  uint32_t *c = nullptr;
  if (hipMalloc((void **)&c, (size_t)(FFN_NQ + cap + 1) * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(c, 0, (size_t)(FFN_NQ + cap + 1) * sizeof(uint32_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    if (c) (void)hipFree(c);
    return nullptr;
  }
  if (e == nullptr) {
    e = &g_ctl[g_ctl_n++];
    *e = FfnCtl{dev, st, nullptr, 0};
  } else {
    (void)hipStreamSynchronize(st);  // a launch queued on this stream may still use the old words
    (void)hipFree(e->ctl);
  }
  e->ctl = c;
  e->experts = cap;
  return c;
}

static int ffn_cus() {
  static int n[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (n[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n[dev] = v;
  }
  return n[dev];
}

template <typename T, bool FL> static int ffn_launch_cfg(const FfnArgs &a, int grid, hipStream_t st) {
  auto kern = expert_ffn_kernel<T, FL>;
  if (!tutel_lds_optin((const void *)kern, ffn_lds_bytes<FL>())) return -1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FFN_THREADS), ffn_lds_bytes<FL>(), st, a);
  TUTEL_CHECK_LAUNCH("tutel_amd_expert_ffn");
  return 0;
}

// Does this pair of GEMMs take the persistent kernel?  The conditions under which launch_gemm (expert_gemm.hip) picks the 128 x 256
// ring tile with its LDS epilogue for BOTH, so that the fused launch runs exactly the tiles the two launches would.
static bool ffn_covers(const GemmArgs &g1, const GemmArgs &g2) {
  auto ring = [](const GemmArgs &g) {
    return g.fits32 && g.N >= 256 && g.R <= GM_BM && (long long)g.E_loc * ((g.N + 255) / 256) >= 256 && g.mul == nullptr && g.d_peer == nullptr &&
           g.row_counts == nullptr && !((g.ldd & 7) || (g.d_stride_e & 7) || (g.d_stride_w & 7) || (reinterpret_cast<uintptr_t>(g.D) & 15));
  };
  return ring(g1) && ring(g2) && g1.E_loc == g2.E_loc && g1.R == g2.R && g1.d_store == 1 && tutel_get_option(TUTEL_OPT_GEMM_IMPL) < 0 &&
         tutel_get_option(TUTEL_OPT_GEMM_TILE) < 0;
}

// development probe: a device array [grid][8] of per-workgroup tick sums (see FfnArgs::dbg); NULL = off
static unsigned long long *g_ffn_dbg = nullptr;
extern "C" void tutel_amd_expert_ffn_debug(void *buf) { g_ffn_dbg = (unsigned long long *)buf; }

// internal (common.h): the fused FFN with every option of the two-launch path it replaces.  loc != NULL: fused location (idx8 [n]).
// query != 0: only answers (0 / TUTEL_AMD_ENOTSUP), launches nothing.
int tutel_expert_ffn(const void *X, int64_t x_stride_e, int ldx, const int32_t *slot_map, int T, const void *zero_row, const void *W1,
                     int64_t w1_stride_e, int ldw1, const void *b1, int64_t b1_stride_e, void *hid, int64_t hid_stride_e, int ldh,
                     const void *W2, int64_t w2_stride_e, int ldw2, const void *b2, int64_t b2_stride_e, void *D, int64_t d_stride_e, int ldd,
                     int E_loc, int R, int M, int H, int M_out, int dtype, int act, const uint8_t *idx8, int n, int32_t *loc, int query,
                     hipStream_t st) {
  // 1 / 2 / 3 = the persistent launch; 0 and AUTOMATIC = the two launches: measured faster (see the file comment)
  const int mode = tutel_get_option(TUTEL_OPT_FFN_FUSED);
  if (mode <= 0) return TUTEL_AMD_ENOTSUP;
  if (E_loc <= 0 || R <= 0) return TUTEL_AMD_ENOTSUP;
  FfnArgs a;
  const int rpw = R > 0 ? R : 1;
  int rc = tutel_gemm_args(X, slot_map ? 0 : x_stride_e, 0, rpw, ldx, W1, 1, w1_stride_e, ldw1, b1, b1_stride_e, hid, hid_stride_e, 0, rpw, ldh,
                           E_loc, R, H, M, dtype, act, nullptr, 1, slot_map, slot_map ? T : 0, slot_map ? zero_row : nullptr, nullptr, nullptr, 0,
                           nullptr, idx8, n, loc, &a.g[0]);
  if (rc != 0) return rc < 0 ? rc : TUTEL_AMD_ENOTSUP;
  rc = tutel_gemm_args(hid, hid_stride_e, 0, rpw, ldh, W2, 1, w2_stride_e, ldw2, b2, b2_stride_e, D, d_stride_e, 0, rpw, ldd, E_loc, R, M_out, H,
                       dtype, TUTEL_ACT_NONE, nullptr, 1, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, &a.g[1]);
  if (rc != 0) return rc < 0 ? rc : TUTEL_AMD_ENOTSUP;
  const bool fl = idx8 != nullptr;
  if (!ffn_covers(a.g[0], a.g[1])) return TUTEL_AMD_ENOTSUP;
  if (fl && !(slot_map != nullptr && n >= 1 && n <= 15360 && E_loc <= 128 && M >= 2 * GL_BK && ((uintptr_t)idx8 & 15) == 0 &&
              tutel_get_option(TUTEL_OPT_FUSED_LOCATION) != 0))
    return TUTEL_AMD_ENOTSUP;
  if (query) return 0;
  for (int i = 0; i < 2; ++i) {
    a.g[i].ntm = 1;
    a.g[i].ntn = (a.g[i].N + 255) / 256;
  }
  a.ctl = ffn_ctl(st, E_loc);
  if (a.ctl == nullptr) return TUTEL_AMD_ENOTSUP;  // (first use inside a stream capture: the two-launch path is always there)
  a.xcc_queue = mode == 2 ? 0 : 1;
  a.prefetch = mode == 3 ? 0 : 1;
  a.dbg = g_ffn_dbg;
  const long long items = (long long)E_loc * (a.g[0].ntn + a.g[1].ntn);
  const int cus = ffn_cus();
  const int grid = (int)(items < cus ? items : cus);
  StageScope stage(TUTEL_STAGE_FC1, st);
  TUTEL_REQUIRE(act >= TUTEL_ACT_NONE && act <= TUTEL_ACT_SILU, "tutel_amd_expert_ffn: unknown activation %d", act);
  a.g[0].act_rt = act;
  a.g[1].act_rt = TUTEL_ACT_NONE;
  if (dtype == TUTEL_BF16) return fl ? ffn_launch_cfg<bf16_t, true>(a, grid, st) : ffn_launch_cfg<bf16_t, false>(a, grid, st);
  return fl ? ffn_launch_cfg<f16_t, true>(a, grid, st) : ffn_launch_cfg<f16_t, false>(a, grid, st);
}

extern "C" int tutel_amd_expert_ffn(const void *X, int64_t x_stride_e, int ldx, const int32_t *slot_map, int T, const void *zero_row,
                                    const void *W1, int64_t w1_stride_e, int ldw1, const void *b1, int64_t b1_stride_e, void *hid,
                                    int64_t hid_stride_e, int ldh, const void *W2, int64_t w2_stride_e, int ldw2, const void *b2,
                                    int64_t b2_stride_e, void *D, int64_t d_stride_e, int ldd, int E_loc, int R, int M, int H, int M_out,
                                    int dtype, int act, tutel_stream_t stream) {
  TUTEL_REQUIRE(dtype == TUTEL_BF16 || dtype == TUTEL_F16, "tutel_amd_expert_ffn: dtype must be bf16 or fp16 (got %d)", dtype);
  TUTEL_REQUIRE(E_loc >= 0 && R >= 0 && M >= 1 && H >= 1 && M_out >= 1, "tutel_amd_expert_ffn: bad sizes");
  if (E_loc == 0 || R == 0) return 0;
  TUTEL_REQUIRE(X && W1 && hid && W2 && D, "tutel_amd_expert_ffn: null pointer");
  TUTEL_REQUIRE(slot_map == nullptr || (T >= 1 && zero_row != nullptr), "tutel_amd_expert_ffn: the row gather needs T >= 1 and a zero row");
  return tutel_expert_ffn(X, x_stride_e, ldx, slot_map, T, zero_row, W1, w1_stride_e, ldw1, b1, b1_stride_e, hid, hid_stride_e, ldh, W2,
                          w2_stride_e, ldw2, b2, b2_stride_e, D, d_stride_e, ldd, E_loc, R, M, H, M_out, dtype, act, nullptr, 0, nullptr, 0,
                          (hipStream_t)stream);
}
