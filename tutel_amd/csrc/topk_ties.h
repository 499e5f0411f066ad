// topk_ties.h -- which expert ids torch.topk returns ON THE CPU when scores are exactly equal (TUTEL_OPT_TIE_RULE).
//
// The reference routes with torch.topk(scores, k, dim=1) (tutel/impls/fast_dispatch.py:146-148); north_star asks for the token-to-expert
// assignment of its CPU path bit for bit, and with 16-bit gates 2 % of the rows at the headline shape carry an exact tie at the k / k+1
// boundary (SURVEY section 7 hard part 1) -- one tie moves every later slot of the two experts involved.  ATen's CPU kernel
// (aten/src/ATen/native/cpu/TopKImpl.h, topk_impl_loop) compares VALUES only, gt(x, y) = (isnan(x) && !isnan(y)) || x > y, and runs
//     k * 64 <= n :  std::partial_sort(q, q + k, q + n, gt)
//     else        :  std::nth_element(q, q + k - 1, q + n, gt);  std::sort(q, q + k - 1, gt)
// over q = [(value, index)] in index order: which of two equal scores comes out first is whatever libstdc++'s introselect / heap
// select leave there -- a pure function of the row, so it can be reproduced.  The top-k kernels (routing.hip) keep their parallel
// arg-max for the rows where the answer is unique and replay the two library routines only over the rows that carry a NaN or a tie
// among their k + 1 largest scores (one extra arg-max round detects them).  The routines below are statement-by-statement
// restatements of <bits/stl_algo.h> / <bits/stl_heap.h> (GCC 11..14), written once over a QUEUE POLICY that says where the
// (value, index) pairs live:
//   WaveQueue  one wave works on one row, queue position i in lane i & 63 (register slot i >> 6).  An element is value + index in two
//              registers -- or, for scores of a 16-bit dtype (where ties actually occur), ONE register: a 16-bit KEY above the index,
//              the key an order-preserving map of the score under ATen's comparator (NaN above everything, -0 = +0), so that gt() is
//              one unsigned compare and the scalar parts of the routines stay on the scalar unit.  Single elements move with
//              v_readlane / a one-lane move, all control flow is wave-uniform -- and the two loops that walk the whole
//              range, introselect's partition and heap_select's scan, are done for all positions AT ONCE with ballots (see
//              partition()).  Measured at the headline shape (170 of 4096 rows replayed): queue in LDS under one lane 47 us for the
//              top-k kernel (8 us without ties; every access a ~100-cycle dependent round trip), queue in registers with the
//              sequential loops 36 us (instruction count: ~5 k dependent instructions per row).  Used for E <= 128.
//   LdsQueue   one lane, queue of expert ids in LDS beside the read-only row: any E (the wave-per-token kernel, E > 128).
// The tests compare the kernels with live torch.topk on the CPU, element for element, on tie-heavy rows (tests/test_ops_gpu.py).
// k <= 16 here, so std::sort(q, q + k - 1) is its insertion-sort tail only (threshold 16).
#pragma once
#include "common.h"

// ---- elements and how they compare ------------------------------------------------------------------------------------------
// ATen: gt(x, y) = (isnan(x) && !isnan(y)) || x > y, on the VALUE only
template <typename CT> struct TkElem { CT v; int id; };
template <typename CT> __device__ __forceinline__ bool tk_gt(const TkElem<CT> &x, const TkElem<CT> &y) {
  return ((x.v != x.v) && !(y.v != y.v)) || x.v > y.v;
}
__device__ __forceinline__ float tk_readlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double tk_readlane(double x, int l) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// value + index in two registers (fp32 / fp64 scores)
template <typename CT> struct TkWide {
  using E = TkElem<CT>;   // an element as the routines hold it (wave-uniform when it came from get())
  CT v;
  int id;
  __device__ __forceinline__ static bool gt(const E &x, const E &y) { return tk_gt(x, y); }
  __device__ __forceinline__ static int id_of(const E &e) { return e.id; }
  __device__ __forceinline__ static TkWide make(CT v, int id) { return TkWide{v, id}; }
  __device__ __forceinline__ static TkWide pack(const E &e) { return TkWide{e.v, e.id}; }
  __device__ __forceinline__ E unpack() const { return E{v, id}; }
  __device__ __forceinline__ TkWide lane(int l) const { return TkWide{tk_readlane(v, l), __builtin_amdgcn_readlane(id, l)}; }   // uniform l
  __device__ __forceinline__ TkWide from(int l) const { return TkWide{__shfl(v, l, 64), __shfl(id, l, 64)}; }                    // per-lane l
  __device__ __forceinline__ static TkWide pick(bool c, const TkWide &a, const TkWide &b) { return TkWide{c ? a.v : b.v, c ? a.id : b.id}; }
};
// one register: (key16 << 16) | index, for scores that are exactly their 16 bits (T = bf16_t / f16_t).  key16 orders the scores as
// ATen's comparator does: every NaN -> 0xffff (all NaNs are equivalent and above +inf), -0 -> +0 (they compare equal), negative
// values -> the complement of their bits, the rest -> bits | 0x8000; equal keys <=> neither element is gt the other.
template <typename T> struct TkKeyed {
  using E = uint32_t;
  uint32_t w;
  __device__ __forceinline__ static bool gt(E x, E y) { return (x >> 16) > (y >> 16); }
  __device__ __forceinline__ static int id_of(E e) { return (int)(e & 0xffff); }
  __device__ __forceinline__ static TkKeyed make(float v, int id) {
    const T t = Elem<T>::from_f32(v);   // (v is a T-typed score held in fp32: exact)
    uint16_t b;
    __builtin_memcpy(&b, &t, 2);
    if (b == 0x8000) b = 0;
    uint32_t key = (b & 0x8000) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000);
    if (v != v) key = 0xffff;
    return TkKeyed{(key << 16) | (uint32_t)(id & 0xffff)};
  }
  __device__ __forceinline__ static TkKeyed pack(E e) { return TkKeyed{e}; }
  __device__ __forceinline__ E unpack() const { return w; }
  __device__ __forceinline__ TkKeyed lane(int l) const { return TkKeyed{(uint32_t)__builtin_amdgcn_readlane((int)w, l)}; }
  __device__ __forceinline__ TkKeyed from(int l) const { return TkKeyed{(uint32_t)__shfl((int)w, l, 64)}; }
  __device__ __forceinline__ static TkKeyed pick(bool c, const TkKeyed &a, const TkKeyed &b) { return TkKeyed{c ? a.w : b.w}; }
};
// the register representation the top-k kernels use for scores of dtype T
template <typename T> struct TkRepOf { using type = TkWide<typename Elem<T>::ct>; };
template <> struct TkRepOf<bf16_t> { using type = TkKeyed<bf16_t>; };
template <> struct TkRepOf<f16_t> { using type = TkKeyed<f16_t>; };

// ---- queue policies ---------------------------------------------------------------------------------------------------------
template <int SL, typename Rep> struct WaveQueue {  // every call is made by the whole wave with wave-uniform arguments
  using E = typename Rep::E;
  Rep r[SL];
  uint8_t *sel;  // LDS scratch of this wave, 2 * seln bytes: see partition()
  int seln;      // queue length n (the two rank -> position tables hold at most n entries each)
  __device__ __forceinline__ static bool gt(const E &x, const E &y) { return Rep::gt(x, y); }
  __device__ __forceinline__ E get(int i) const {
    Rep x = r[0].lane(i & 63);
#pragma unroll
    for (int s = 1; s < SL; ++s)
      if ((i >> 6) == s) x = r[s].lane(i & 63);  // uniform
    return x.unpack();
  }
  __device__ __forceinline__ void set(int i, const E &e) {
    const bool me = (int)(threadIdx.x & 63) == (i & 63);
    const Rep x = Rep::pack(e);
#pragma unroll
    for (int s = 0; s < SL; ++s)
      if ((i >> 6) == s) r[s] = Rep::pick(me, x, r[s]);  // uniform
  }
  // smallest position i' in [i, last) whose element is gt(., top), or -1: one ballot per slot instead of a scan (heap_select)
  __device__ __forceinline__ int next_gt(int i, int last, const E &top) const {
    const int lane = (int)(threadIdx.x & 63);
    int found = -1;
#pragma unroll
    for (int s = SL - 1; s >= 0; --s) {
      const int pos = lane + 64 * s;
      const unsigned long long m = __ballot(pos >= i && pos < last && Rep::gt(r[s].unpack(), top));
      if (m != 0ull) found = 64 * s + (int)__builtin_ctzll(m);
    }
    return found;
  }
  // libstdc++'s __unguarded_partition(first + 1, last, pivot = *first), all positions at once.  The sequential loop stops its left
  // cursor at the successive positions A = {p : !gt(q[p], pivot)} in ascending order and its right cursor at B = {p : !gt(pivot, q[p])}
  // in descending order, swapping the i-th of one with the i-th of the other while a_i < b_i; a cursor only ever examines positions
  // no swap has touched, or -- as its guard -- the most recently swapped one.  So: both sets by ballot, the rank of every member by
  // popcount, the partner through a rank -> position table in LDS, s = #{i : a_i < b_i} swaps done in one exchange, and the returned
  // cut is min(a_s, b_{s-1}) (the next untouched stop of the left cursor, or the guard the last swap left behind) -- the members of
  // rank s / s - 1, found by ballot again.
  __device__ __forceinline__ int partition(int first, int last, const E &pivot) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long mA[SL], mB[SL];
    bool a[SL], b[SL];
    int nA = 0, nB = 0;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const int pos = lane + 64 * s;
      const bool inr = pos > first && pos < last;
      const E e = r[s].unpack();
      a[s] = inr && !Rep::gt(e, pivot);
      b[s] = inr && !Rep::gt(pivot, e);
      mA[s] = __ballot(a[s]);
      mB[s] = __ballot(b[s]);
      nA += __popcll(mA[s]);
      nB += __popcll(mB[s]);
    }
    uint8_t *selA = sel, *selB = sel + seln;
    int rA[SL], rB[SL];  // rank of this position among A from the left / among B from the right
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      int ra = __popcll(mA[s] & lt), rbl = __popcll(mB[s] & lt);
#pragma unroll
      for (int s2 = 0; s2 < s; ++s2) {
        ra += __popcll(mA[s2]);
        rbl += __popcll(mB[s2]);
      }
      rA[s] = ra;
      rB[s] = nB - 1 - rbl;
      if (a[s]) selA[ra] = (uint8_t)(lane + 64 * s);
      if (b[s]) selB[rB[s]] = (uint8_t)(lane + 64 * s);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int pp[SL];  // partner position, or -1
    int nsw = 0;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const int pos = lane + 64 * s;
      pp[s] = -1;
      bool swa = false;
      if (a[s] && rA[s] < nB) {
        const int pb = selB[rA[s]];
        if (pos < pb) { pp[s] = pb; swa = true; }
      }
      if (b[s] && rB[s] < nA) {
        const int pa = selA[rB[s]];
        if (pa < pos) pp[s] = pa;
      }
      nsw += __popcll(__ballot(swa));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the tables are rewritten by the next call)
    // the cut: the A member of rank nsw, or the B member of rank nsw - 1, whichever is further left
    int cut = 0x7fffffff;
#pragma unroll
    for (int s = SL - 1; s >= 0; --s) {
      const unsigned long long ma = __ballot(a[s] && rA[s] == nsw);
      if (ma != 0ull) cut = 64 * s + (int)__builtin_ctzll(ma);
    }
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const unsigned long long mb = __ballot(b[s] && rB[s] == nsw - 1);
      if (mb != 0ull) { const int g = 64 * s + (int)__builtin_ctzll(mb); cut = g < cut ? g : cut; }
    }
    // the exchange: every new value is read from the OLD registers first
    Rep nr[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      nr[s] = r[s];
#pragma unroll
      for (int s2 = 0; s2 < SL; ++s2) {
        const Rep f = r[s2].from(pp[s] & 63);
        nr[s] = Rep::pick(pp[s] >= 0 && (pp[s] >> 6) == s2, f, nr[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < SL; ++s) r[s] = nr[s];
    return cut;
  }
};

template <typename CT, typename IT> struct LdsQueue {
  using E = TkElem<CT>;
  const CT *val;  // [n] the row (read only)
  IT *p;          // [n] the queue: expert ids
  __device__ __forceinline__ static bool gt(const E &x, const E &y) { return tk_gt(x, y); }
  __device__ __forceinline__ E get(int i) const {
    E e;
    e.id = p[i];
    e.v = val[e.id];
    return e;
  }
  __device__ __forceinline__ void set(int i, const E &e) { p[i] = (IT)e.id; }
  __device__ __forceinline__ int next_gt(int i, int last, const E &top) const {
    for (; i < last; ++i)
      if (tk_gt(get(i), top)) return i;
    return -1;
  }
  __device__ __forceinline__ int partition(int first, int last, const E &pivot) {  // __unguarded_partition(first + 1, last, first)
    int f = first + 1, l = last;
    for (;;) {
      while (tk_gt(get(f), pivot)) ++f;
      --l;
      while (tk_gt(pivot, get(l))) --l;
      if (!(f < l)) return f;
      const E x = get(f), y = get(l);
      set(f, y);
      set(l, x);
      ++f;
    }
  }
};

// ---- the library routines ---------------------------------------------------------------------------------------------------
template <typename Q> struct AtenTopk {
  using E = typename Q::E;
  Q &q;
  __device__ __forceinline__ explicit AtenTopk(Q &queue) : q(queue) {}
  __device__ __forceinline__ static bool gt(const E &x, const E &y) { return Q::gt(x, y); }

  // <bits/stl_heap.h>
  __device__ __forceinline__ void push_heap(int first, int hole, int top, const E &value) {
    int parent = (hole - 1) / 2;
    while (hole > top) {
      const E pe = q.get(first + parent);
      if (!gt(pe, value)) break;
      q.set(first + hole, pe);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    q.set(first + hole, value);
  }
  __device__ __forceinline__ void adjust_heap(int first, int hole, int len, const E &value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      E c = q.get(first + child);
      const E c1 = q.get(first + child - 1);
      if (gt(c, c1)) { child--; c = c1; }
      q.set(first + hole, c);
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      q.set(first + hole, q.get(first + child - 1));
      hole = child - 1;
    }
    push_heap(first, hole, top, value);
  }
  __device__ __forceinline__ void make_heap(int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    for (;;) {
      adjust_heap(first, parent, len, q.get(first + parent));
      if (parent == 0) return;
      parent--;
    }
  }
  __device__ __forceinline__ void pop_heap(int first, int last, int result) {
    const E value = q.get(result);
    q.set(result, q.get(first));
    adjust_heap(first, 0, last - first, value);
  }
  __device__ __forceinline__ void heap_select(int first, int middle, int last) {
    make_heap(first, middle);
    // for (i = middle; i < last; ++i) if (gt(q[i], q[first])) pop_heap(first, middle, i) -- positions >= i are untouched originals
    for (int i = middle;;) {
      i = q.next_gt(i, last, q.get(first));
      if (i < 0) break;
      pop_heap(first, middle, i);
      ++i;
    }
  }
  __device__ __forceinline__ void partial_sort(int first, int middle, int last) {
    heap_select(first, middle, last);
    while (middle - first > 1) {  // __sort_heap
      --middle;
      pop_heap(first, middle, middle);
    }
  }
  // <bits/stl_algo.h>
  __device__ __forceinline__ void swap(int a, int b) {
    const E x = q.get(a), y = q.get(b);
    q.set(a, y);
    q.set(b, x);
  }
  __device__ __forceinline__ void insertion_sort(int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      const E v = q.get(i);
      if (gt(v, q.get(first))) {
        for (int j = i; j != first; --j) q.set(j, q.get(j - 1));
        q.set(first, v);
      } else {  // __unguarded_linear_insert
        int l = i, nx = i - 1;
        for (;;) {
          const E ne = q.get(nx);
          if (!gt(v, ne)) break;
          q.set(l, ne);
          l = nx;
          --nx;
        }
        q.set(l, v);
      }
    }
  }
  __device__ __forceinline__ int partition_pivot(int first, int last) {  // __unguarded_partition_pivot
    const int mid = first + (last - first) / 2;
    // __move_median_to_first(first, first + 1, mid, last - 1), then the partition around *first (which the partition never moves)
    const int a = first + 1, b = mid, c = last - 1;
    const E ea = q.get(a), eb = q.get(b), ec = q.get(c), ef = q.get(first);
    int s;
    if (gt(ea, eb)) s = gt(eb, ec) ? b : (gt(ea, ec) ? c : a);
    else s = gt(ea, ec) ? a : (gt(eb, ec) ? c : b);
    const E es = s == a ? ea : (s == b ? eb : ec);
    q.set(first, es);
    q.set(s, ef);
    return q.partition(first, last, es);
  }
  __device__ __forceinline__ void nth_element(int nth, int n) {  // std::nth_element(q, q + nth, q + n)
    if (n == 0 || nth == n) return;
    int first = 0, last = n, depth = 0;
    for (int m = n; m > 1; m >>= 1) ++depth;
    depth *= 2;
    while (last - first > 3) {
      if (depth == 0) {
        heap_select(first, nth + 1, last);
        swap(first, nth);
        return;
      }
      --depth;
      const int cut = partition_pivot(first, last);
      if (cut <= nth) first = cut;
      else last = cut;
    }
    insertion_sort(first, last);
  }
  // ATen topk_impl_loop, largest = sorted = true; on return queue positions [0, k) hold the chosen experts in torch.topk's order
  __device__ __forceinline__ void run(int n, int k) {
    if (k * 64 <= n) {
      partial_sort(0, k, n);
    } else {
      nth_element(k - 1, n);
      insertion_sort(0, k - 1);  // std::sort of <= 16 elements
    }
  }
};
