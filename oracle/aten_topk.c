/*
 * aten_topk.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or executed by the product).
 *
 * Which expert ids `torch.topk(scores, k, dim=1)` returns ON THE CPU -- including its order among EXACTLY equal scores, which
 * the reference inherits (tutel/impls/fast_dispatch.py:146-148) and which SURVEY.md section 7 hard part 1 / VERDICT r5 item 1
 * make part of "bit-exact token-to-expert index assignment".
 *
 * The algorithm is not in /root/reference: it lives in the reference's dependency PyTorch (pinned by this image:
 * torch 2.10.0+rocm7.0, the version the reference runs on here) and, below that, in libstdc++.  Restated from their published
 * sources:
 *   ATen  aten/src/ATen/native/cpu/TopKImpl.h  `topk_impl_loop`:  per row, queue = [(value, index)] in index order;
 *         comparator gt(x, y) = (isnan(x) && !isnan(y)) || x > y  -- on the VALUE only, the index never breaks a tie;
 *         if k * 64 <= n:  std::partial_sort(queue, queue + k, end, gt)
 *         else:            std::nth_element(queue, queue + k - 1, end, gt);  then (sorted=True) std::sort(queue, queue + k - 1, gt)
 *         result j = queue[j].second.       (bf16 rows are compared as float, fp16 as Half -> float: the same order.)
 *   libstdc++  <bits/stl_algo.h>, <bits/stl_heap.h> (GCC 11..14, unchanged in the parts used): __introselect with
 *         median-of-three __unguarded_partition_pivot, depth limit 2 * floor(log2 n) then __heap_select; __insertion_sort for
 *         <= 3 elements; std::sort = __introsort_loop (threshold 16) + __final_insertion_sort; __partial_sort = __heap_select +
 *         __sort_heap; the heap primitives __push_heap / __adjust_heap / __make_heap / __pop_heap.
 * Every step below follows those functions statement by statement (they are deterministic: which of two equal scores ends up
 * first is a pure function of the row).
 *
 * Pinned by tests/test_oracle_vs_reference.py::test_aten_topk_restatement_equals_torch_topk (live torch.topk in this container on
 * tie-heavy rows for E = 1..300, k = 1..16, both branches) and by the reference-written fixtures tests/golden/headline_gate_*.npz
 * (the reference's own idx at T = 4096, E = 64, k = 2 with 16-bit gates, 89 / 15 rows of which carry exact ties).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { double v; int64_t i; } tk_elem;   /* float -> double is exact and order-preserving (NaN stays NaN) */

static int tk_gt(const tk_elem *x, const tk_elem *y) {
  return (isnan(x->v) && !isnan(y->v)) || (x->v > y->v);
}
static void tk_swap(tk_elem *a, tk_elem *b) { tk_elem t = *a; *a = *b; *b = t; }
static int tk_lg(long n) { int l = 0; while (n > 1) { n >>= 1; ++l; } return l; }

/* ---- <bits/stl_heap.h> ------------------------------------------------------------------------------------------------ */
static void tk_push_heap(tk_elem *first, long hole, long top, tk_elem value) {
  long parent = (hole - 1) / 2;
  while (hole > top && tk_gt(first + parent, &value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void tk_adjust_heap(tk_elem *first, long hole, long len, tk_elem value) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (tk_gt(first + child, first + (child - 1))) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  tk_push_heap(first, hole, top, value);
}
static void tk_make_heap(tk_elem *first, tk_elem *last) {
  const long len = last - first;
  if (len < 2) return;
  long parent = (len - 2) / 2;
  for (;;) {
    tk_elem value = first[parent];
    tk_adjust_heap(first, parent, len, value);
    if (parent == 0) return;
    parent--;
  }
}
static void tk_pop_heap(tk_elem *first, tk_elem *last, tk_elem *result) {
  tk_elem value = *result;
  *result = *first;
  tk_adjust_heap(first, 0, last - first, value);
}
static void tk_heap_select(tk_elem *first, tk_elem *middle, tk_elem *last) {
  tk_make_heap(first, middle);
  for (tk_elem *i = middle; i < last; ++i)
    if (tk_gt(i, first)) tk_pop_heap(first, middle, i);
}
static void tk_sort_heap(tk_elem *first, tk_elem *last) {
  while (last - first > 1) {
    --last;
    tk_pop_heap(first, last, last);
  }
}
static void tk_partial_sort(tk_elem *first, tk_elem *middle, tk_elem *last) {
  tk_heap_select(first, middle, last);
  tk_sort_heap(first, middle);
}

/* ---- <bits/stl_algo.h> ------------------------------------------------------------------------------------------------ */
static void tk_move_median_to_first(tk_elem *result, tk_elem *a, tk_elem *b, tk_elem *c) {
  if (tk_gt(a, b)) {
    if (tk_gt(b, c)) tk_swap(result, b);
    else if (tk_gt(a, c)) tk_swap(result, c);
    else tk_swap(result, a);
  } else if (tk_gt(a, c)) tk_swap(result, a);
  else if (tk_gt(b, c)) tk_swap(result, c);
  else tk_swap(result, b);
}
static tk_elem *tk_unguarded_partition(tk_elem *first, tk_elem *last, tk_elem *pivot) {
  for (;;) {
    while (tk_gt(first, pivot)) ++first;
    --last;
    while (tk_gt(pivot, last)) --last;
    if (!(first < last)) return first;
    tk_swap(first, last);
    ++first;
  }
}
static tk_elem *tk_unguarded_partition_pivot(tk_elem *first, tk_elem *last) {
  tk_elem *mid = first + (last - first) / 2;
  tk_move_median_to_first(first, first + 1, mid, last - 1);
  return tk_unguarded_partition(first + 1, last, first);
}
static void tk_unguarded_linear_insert(tk_elem *last) {
  tk_elem val = *last;
  tk_elem *next = last - 1;
  while (tk_gt(&val, next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}
static void tk_insertion_sort(tk_elem *first, tk_elem *last) {
  if (first == last) return;
  for (tk_elem *i = first + 1; i != last; ++i) {
    if (tk_gt(i, first)) {
      tk_elem val = *i;
      for (tk_elem *p = i; p != first; --p) *p = *(p - 1);   /* std::move_backward(first, i, i + 1) */
      *first = val;
    } else {
      tk_unguarded_linear_insert(i);
    }
  }
}
static void tk_introselect(tk_elem *first, tk_elem *nth, tk_elem *last, int depth_limit) {
  while (last - first > 3) {
    if (depth_limit == 0) {
      tk_heap_select(first, nth + 1, last);
      tk_swap(first, nth);
      return;
    }
    --depth_limit;
    tk_elem *cut = tk_unguarded_partition_pivot(first, last);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  tk_insertion_sort(first, last);
}
static void tk_nth_element(tk_elem *first, tk_elem *nth, tk_elem *last) {
  if (first == last || nth == last) return;
  tk_introselect(first, nth, last, tk_lg(last - first) * 2);
}
static void tk_introsort_loop(tk_elem *first, tk_elem *last, int depth_limit) {
  while (last - first > 16) {
    if (depth_limit == 0) {
      tk_partial_sort(first, last, last);
      return;
    }
    --depth_limit;
    tk_elem *cut = tk_unguarded_partition_pivot(first, last);
    tk_introsort_loop(cut, last, depth_limit);
    last = cut;
  }
}
static void tk_sort(tk_elem *first, tk_elem *last) {
  if (first == last) return;
  tk_introsort_loop(first, last, tk_lg(last - first) * 2);
  if (last - first > 16) {
    tk_insertion_sort(first, first + 16);
    for (tk_elem *i = first + 16; i != last; ++i) tk_unguarded_linear_insert(i);
  } else {
    tk_insertion_sort(first, last);
  }
}

/* ---- ATen topk_impl_loop (largest = true, sorted = true) -------------------------------------------------------------- */
static void tk_row(tk_elem *queue, int n, int k) {
  if (k * 64 <= n) {
    tk_partial_sort(queue, queue + k, queue + n);
  } else {
    tk_nth_element(queue, queue + k - 1, queue + n);
    tk_sort(queue, queue + k - 1);
  }
}

/* idx layout [k][T] int32, as orc_topk_* (moe_oracle.c).  Returns 0, or -1 when out of memory. */
#define DEFINE_ATEN_TOPK(NAME, TYPE)                                                        \
  int NAME(const TYPE *scores, int T, int E, int k, int32_t *idx) {                         \
    if (k <= 0 || E <= 0) return 0;                                                         \
    tk_elem *queue = (tk_elem *)malloc(sizeof(tk_elem) * (size_t)E);                        \
    if (queue == NULL) return -1;                                                           \
    for (int t = 0; t < T; ++t) {                                                           \
      for (int e = 0; e < E; ++e) { queue[e].v = (double)scores[(size_t)t * E + e]; queue[e].i = e; } \
      tk_row(queue, E, k);                                                                  \
      for (int j = 0; j < k; ++j) idx[(size_t)j * T + t] = (int32_t)queue[j].i;             \
    }                                                                                       \
    free(queue);                                                                            \
    return 0;                                                                               \
  }
DEFINE_ATEN_TOPK(orc_aten_topk_f32, float)
DEFINE_ATEN_TOPK(orc_aten_topk_f64, double)
