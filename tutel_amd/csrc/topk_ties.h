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
// wave arg-max for the rows where the answer is unique and hand only the rows that carry a NaN or a tie among their k + 1 largest scores
// (one extra arg-max round detects them) to ONE lane, which replays the two library routines below over the row in LDS: the queue is
// kept as expert ids (`p`), values are looked up in the read-only row (`val`).  Statement-by-statement restatements of
// <bits/stl_algo.h> / <bits/stl_heap.h> (GCC 11..14); the tests compare the kernels with live torch.topk on the CPU, element for
// element, on tie-heavy rows (tests/test_ops_gpu.py).  k <= 16 here, so std::sort(q, q + k - 1) is its insertion-sort tail only (threshold 16).
#pragma once
#include "common.h"

template <typename CT, typename IT> struct AtenTopk {
  const CT *val;  // [n] the row
  IT *p;          // [n] the queue: expert ids

  __device__ __forceinline__ bool gt(int a, int b) const {  // a, b: expert ids
    const CT x = val[a], y = val[b];
    return ((x != x) && !(y != y)) || x > y;
  }
  // ---- <bits/stl_heap.h>
  __device__ void push_heap(int first, int hole, int top, int value) {
    int parent = (hole - 1) / 2;
    while (hole > top && gt(p[first + parent], value)) {
      p[first + hole] = p[first + parent];
      hole = parent;
      parent = (hole - 1) / 2;
    }
    p[first + hole] = (IT)value;
  }
  __device__ void adjust_heap(int first, int hole, int len, int value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (gt(p[first + child], p[first + child - 1])) child--;
      p[first + hole] = p[first + child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      p[first + hole] = p[first + child - 1];
      hole = child - 1;
    }
    push_heap(first, hole, top, value);
  }
  __device__ void make_heap(int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    int parent = (len - 2) / 2;
    for (;;) {
      adjust_heap(first, parent, len, p[first + parent]);
      if (parent == 0) return;
      parent--;
    }
  }
  __device__ void pop_heap(int first, int last, int result) {
    const int value = p[result];
    p[result] = p[first];
    adjust_heap(first, 0, last - first, value);
  }
  __device__ void heap_select(int first, int middle, int last) {
    make_heap(first, middle);
    for (int i = middle; i < last; ++i)
      if (gt(p[i], p[first])) pop_heap(first, middle, i);
  }
  __device__ void partial_sort(int first, int middle, int last) {
    heap_select(first, middle, last);
    while (middle - first > 1) {  // __sort_heap
      --middle;
      pop_heap(first, middle, middle);
    }
  }
  // ---- <bits/stl_algo.h>
  __device__ __forceinline__ void swap(int a, int b) {
    const IT t = p[a];
    p[a] = p[b];
    p[b] = t;
  }
  __device__ void insertion_sort(int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      const int v = p[i];
      if (gt(v, p[first])) {
        for (int j = i; j != first; --j) p[j] = p[j - 1];
        p[first] = (IT)v;
      } else {  // __unguarded_linear_insert
        int l = i, nx = i - 1;
        while (gt(v, p[nx])) {
          p[l] = p[nx];
          l = nx;
          --nx;
        }
        p[l] = (IT)v;
      }
    }
  }
  __device__ int partition_pivot(int first, int last) {  // __unguarded_partition_pivot
    const int mid = first + (last - first) / 2;
    {  // __move_median_to_first(first, first + 1, mid, last - 1)
      const int a = first + 1, b = mid, c = last - 1;
      if (gt(p[a], p[b])) {
        if (gt(p[b], p[c])) swap(first, b);
        else if (gt(p[a], p[c])) swap(first, c);
        else swap(first, a);
      } else if (gt(p[a], p[c])) swap(first, a);
      else if (gt(p[b], p[c])) swap(first, c);
      else swap(first, b);
    }
    int f = first + 1, l = last;
    const int pivot = p[first];  // (the pivot element itself is never moved by the loop below)
    for (;;) {
      while (gt(p[f], pivot)) ++f;
      --l;
      while (gt(pivot, p[l])) --l;
      if (!(f < l)) return f;
      swap(f, l);
      ++f;
    }
  }
  __device__ void nth_element(int nth, int n) {  // std::nth_element(q, q + nth, q + n)
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
  // ATen topk_impl_loop, largest = sorted = true; on return p[0 .. k) are the chosen expert ids in torch.topk's order
  __device__ __noinline__ void run(int n, int k) {
    if (k * 64 <= n) {
      partial_sort(0, k, n);
    } else {
      nth_element(k - 1, n);
      insertion_sort(0, k - 1);  // std::sort of <= 16 elements
    }
  }
};
