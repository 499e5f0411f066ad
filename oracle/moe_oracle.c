/*
 * moe_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or executed by the product).
 *
 * Plain-C CPU restatement of the integer/index half and the scatter/gather half of the
 * reference's MoE forward hot path (microsoft/tutel @ 2025-02-04).  Every function cites the
 * reference lines it follows (paths relative to /root/reference).  The floating-point GEMM /
 * softmax half of the path is restated in oracle/moe_oracle.py on top of torch-CPU ATen ops,
 * which is what the reference itself calls on its CPU path.
 *
 * Parity pinning: tests/test_oracle_vs_reference.py checks these functions against the
 * reference itself (python package imported from /root/reference + its own C++ CPU kernels
 * compiled into oracle/_ref/ by oracle/Makefile) and tests/golden/ holds fixtures produced by
 * that reference (generator: tests/golden/make_golden.py).
 *
 * Build: make -C oracle   ->  oracle/libmoe_oracle.so   (gcc -ffp-contract=off: no FMA fusion,
 * so a*b+c rounds twice exactly like the reference's scalar C++ loops compiled without -mfma).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * top-k expert selection, "lowest index" tie rule (the product's TUTEL_OPT_TIE_RULE = 0; oracle tie_rule="lowest").
 * Reference: tutel/impls/fast_dispatch.py:146-148  (torch.topk(scores, k, dim=1).indices, then
 * one index vector per choice).  On tie-free rows this IS torch.topk; among EXACTLY equal scores
 * torch.topk's CPU kernel returns what libstdc++'s nth_element / partial_sort leave there -- that
 * order, the default of both the oracle and the product since round 6, is restated in
 * oracle/aten_topk.c.  Here: descending score, ties broken towards the LOWEST expert index.
 * idx layout: [k][T] int32 (one contiguous vector per choice, as the reference's indices_s).
 * NaN scores: a NaN never compares greater, so NaNs are selected last.
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_TOPK(NAME, TYPE)                                                              \
  void NAME(const TYPE *scores, int T, int E, int k, int32_t *idx) {                         \
    for (int t = 0; t < T; ++t) {                                                            \
      const TYPE *row = scores + (size_t)t * E;                                              \
      for (int j = 0; j < k; ++j) {                                                          \
        int best = -1;                                                                       \
        for (int e = 0; e < E; ++e) {                                                        \
          int taken = 0;                                                                     \
          for (int p = 0; p < j; ++p) taken |= (idx[(size_t)p * T + t] == e);                \
          if (taken) continue;                                                               \
          if (best < 0 || row[e] > row[best]) best = e;                                      \
        }                                                                                    \
        idx[(size_t)j * T + t] = best;                                                       \
      }                                                                                      \
    }                                                                                        \
  }
DEFINE_TOPK(orc_topk_f32, float)
DEFINE_TOPK(orc_topk_f64, double)

/* ------------------------------------------------------------------------------------------
 * fast_cumsum_sub_one: per-column inclusive cumsum minus one of an int [T,E] mask.
 * Reference: tutel/jit_kernels/gating.py:13-15,19-24 (CPU: torch.cumsum(mask,0)-1) and the GPU
 * kernel tutel/custom/custom_kernel.cpp:829-868 (same values, int32).
 * ---------------------------------------------------------------------------------------- */
void orc_cumsum_sub_one_i32(const int32_t *mask, int T, int E, int32_t *out) {
  for (int e = 0; e < E; ++e) {
    int32_t run = 0;
    for (int t = 0; t < T; ++t) {
      run += mask[(size_t)t * E + e];
      out[(size_t)t * E + e] = run - 1;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * compute_location (non batch-prioritised).
 * Reference: tutel/impls/fast_dispatch.py:150,159-171,177-178.
 *   masks_se[k]   = one_hot(idx_k)                       (:150, losses.py:7-10)
 *   locations1    = cumsum(mask_0,0)-1                   (:159)
 *   loc_0[t]      = sum_e locations1[t,e]*mask_0[t,e]    (:161)  = rank of t among tokens with
 *                                                                   the same first choice
 *   acc_base      = sum_{k'<k} colsum(mask_k')           (:166)
 *   loc_k[t]      = (cumsum(mask_k,0)-1+acc_base)[t,idx_k[t]]    (:167-169)
 *   dispatch_count= locations_{k-1}[-1] + 1              (:171,177-178)  = per-expert totals
 * idx entries outside [0,E) are treated as "no expert": loc = 0 and not counted (the reference
 * never produces them from extract_critical; fast_dispatcher.update callers may, see :25,59).
 * ---------------------------------------------------------------------------------------- */
void orc_compute_locations(const int32_t *idx, int T, int E, int k, int32_t *loc,
                           int32_t *dispatch_count) {
  int32_t *run = (int32_t *)calloc((size_t)E, sizeof(int32_t));
  for (int j = 0; j < k; ++j) {
    /* run[e] already holds acc_base[e] = number of tokens placed by choices < j */
    for (int t = 0; t < T; ++t) {
      int32_t e = idx[(size_t)j * T + t];
      if (e < 0 || e >= E) { loc[(size_t)j * T + t] = 0; continue; }
      loc[(size_t)j * T + t] = run[e];
      run[e] += 1;
    }
  }
  memcpy(dispatch_count, run, (size_t)E * sizeof(int32_t));
  free(run);
}

/* ------------------------------------------------------------------------------------------
 * Dispatch ("fast_encode") -- one call per top-k choice, accumulating into a zeroed buffer.
 * Reference: tutel/custom/custom_kernel.cpp:293-300 (invoke_cpu kernel_type 0), driven by
 * GatingEncoder.forward, tutel/impls/fast_dispatch.py:26-28.
 *   out[(idx[t]*C + loc[t])*M + j] += gate[t] * x[t*M + j]   iff loc[t] < C && idx[t] >= 0
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_ENCODE(NAME, TYPE)                                                            \
  void NAME(const TYPE *gate, const int32_t *idx, const int32_t *loc, const TYPE *x,         \
            TYPE *out, int T, int M, int C) {                                                \
    for (int t = 0; t < T; ++t) {                                                            \
      if (loc[t] < C && idx[t] >= 0) {                                                       \
        TYPE *dst = out + ((size_t)idx[t] * C + loc[t]) * M;                                 \
        const TYPE *src = x + (size_t)t * M;                                                 \
        for (int j = 0; j < M; ++j) dst[j] += gate[t] * src[j];                              \
      }                                                                                      \
    }                                                                                        \
  }
DEFINE_ENCODE(orc_encode_f32, float)
DEFINE_ENCODE(orc_encode_f64, double)

/* ------------------------------------------------------------------------------------------
 * Combine ("fast_decode") for ONE top-k choice: writes a full [T,M] temp.
 * Reference: tutel/custom/custom_kernel.cpp:301-312 (invoke_cpu kernel_type 1), driven by
 * GatingDecoder.forward, tutel/impls/fast_dispatch.py:61-66 which then sums the k temps
 * left-to-right (last_result + single_output).
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_DECODE(NAME, TYPE)                                                            \
  void NAME(const TYPE *gate, const int32_t *idx, const int32_t *loc, TYPE *y,               \
            const TYPE *buf, int T, int M, int C) {                                          \
    for (int t = 0; t < T; ++t) {                                                            \
      TYPE *dst = y + (size_t)t * M;                                                         \
      if (loc[t] < C && idx[t] >= 0) {                                                       \
        const TYPE *src = buf + ((size_t)idx[t] * C + loc[t]) * M;                           \
        for (int j = 0; j < M; ++j) dst[j] = gate[t] * src[j];                               \
      } else {                                                                               \
        for (int j = 0; j < M; ++j) dst[j] = 0;                                              \
      }                                                                                      \
    }                                                                                        \
  }
DEFINE_DECODE(orc_decode_f32, float)
DEFINE_DECODE(orc_decode_f64, double)

/* ------------------------------------------------------------------------------------------
 * Gate gradient (backward only).
 * Reference: tutel/custom/custom_kernel.cpp:313-322 (invoke_cpu kernel_type 2).
 *   ggate[t] = sum_j buf[(idx[t]*C+loc[t])*M + j] * x[t*M + j]   (0 when dropped)
 * Accumulated left to right in TYPE, exactly as the reference's scalar loop.
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_GATEGRAD(NAME, TYPE)                                                          \
  void NAME(TYPE *ggate, const int32_t *idx, const int32_t *loc, const TYPE *x,              \
            const TYPE *buf, int T, int M, int C) {                                          \
    for (int t = 0; t < T; ++t) {                                                            \
      ggate[t] = 0;                                                                          \
      if (loc[t] >= C || idx[t] < 0) continue;                                               \
      const TYPE *src = buf + ((size_t)idx[t] * C + loc[t]) * M;                             \
      const TYPE *xr = x + (size_t)t * M;                                                    \
      for (int j = 0; j < M; ++j) ggate[t] += src[j] * xr[j];                                \
    }                                                                                        \
  }
DEFINE_GATEGRAD(orc_gate_grad_f32, float)
DEFINE_GATEGRAD(orc_gate_grad_f64, double)

/* ------------------------------------------------------------------------------------------
 * Expert-parallel all-to-all layout, simulated for W ranks inside one process.
 * Reference: tutel/impls/communicate.py:181-192 (all_to_all_single, equal dim-0 chunks),
 * :447-503 (transform input_dim=1 -> output_dim=0) and :606-613 (pre_expert_permute).
 *   send[r]  : [E = W*E_loc, C, M] on rank r (viewed [W(dst), E_loc, C, M])
 *   recv[d]  : [E_loc, W(src)*C, M] on rank d, rows source-rank-major
 * elem = bytes per element (pure byte movement).
 * ---------------------------------------------------------------------------------------- */
void orc_a2a_dispatch_layout(const unsigned char *send, unsigned char *recv, int W, int E_loc,
                             int C, int M, int elem) {
  size_t row = (size_t)M * elem;
  size_t per_rank = (size_t)W * E_loc * C * row;
  for (int src = 0; src < W; ++src)
    for (int dst = 0; dst < W; ++dst)
      for (int e = 0; e < E_loc; ++e)
        for (int c = 0; c < C; ++c) {
          const unsigned char *s =
              send + src * per_rank + (((size_t)dst * E_loc + e) * C + c) * row;
          unsigned char *d =
              recv + dst * per_rank + (((size_t)e * W + src) * C + c) * row;
          memcpy(d, s, row);
        }
}
