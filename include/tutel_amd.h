/*
 * tutel_amd.h -- C ABI of libtutel_amd.so: the MI355X (gfx950) implementation of Tutel's MoE
 * forward hot path.  This is the drop-in boundary: plain pointers + sizes + a HIP stream, no
 * torch types.  Every entry point names the reference interface it replaces (paths relative to
 * microsoft/tutel @ 2025-02-04).  The reference reaches its native code through
 *   - tutel_custom_kernel.invoke(list[Tensor], list[int], blocks, fd)        custom_kernel.cpp:255-275
 *   - tutel_custom_kernel.invoke_cpu_fp32/fp64(list[Tensor], list[int], kt) custom_kernel.cpp:280-323
 *   - torch.ops.tutel_ops.cumsum / sparse_bmm_infer                         custom_kernel.cpp:822-894
 * and through ATen (topk / one_hot / cumsum / bmm) for the rest of the path; INTEGRATION.md
 * shows the ctypes stub a maintainer would drop into tutel/impls/jit_compiler.py.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers (HBM), caller-allocated, contiguous unless strides
 *     are part of the signature; outputs are written in place, nothing is returned but status;
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); every kernel is enqueued on
 *     it and the call returns without synchronising (the reference launches on the *default*
 *     stream, custom_kernel.cpp:268-274 -- taking the caller's current stream is deliberate);
 *   - return value 0 = success; non-zero = error, text via tutel_amd_last_error().  Argument
 *     errors are reported BEFORE anything is enqueued (mirrors CHECK_* -> c10::Error ->
 *     RuntimeError in the reference, custom_kernel.cpp:37-42);
 *   - not thread-safe per stream beyond what HIP guarantees; one process per GPU, like the
 *     reference (custom_kernel.cpp:172,327-338).
 *   - index layout: the k per-choice vectors of the reference's `indices_s / locations_s /
 *     gates_s` lists (fast_dispatch.py:148,161-175) are rows of ONE [k, T] array.
 */
#ifndef TUTEL_AMD_H
#define TUTEL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TUTEL_AMD_ABI_VERSION 1

typedef void *tutel_stream_t; /* hipStream_t */

/* element types of token / weight / score arrays; TUTEL_F64 is accepted by tutel_amd_gate_topk only
 * (scores / gates of an fp64 gate -- the dispatch itself runs in fp32 in the reference as well,
 * fast_dispatch.py:94-96) */
enum { TUTEL_F32 = 0, TUTEL_F16 = 1, TUTEL_BF16 = 2, TUTEL_F64 = 3 };
/* fused activation of the expert GEMM epilogue (experts/ffn.py:19-24,117) */
enum { TUTEL_ACT_NONE = 0, TUTEL_ACT_RELU = 1, TUTEL_ACT_GELU = 2, TUTEL_ACT_SILU = 3 };

/* ---- library info ----------------------------------------------------------------------- */
int tutel_amd_abi_version(void);
const char *tutel_amd_target_arch(void); /* "gfx950" */
const char *tutel_amd_last_error(void);  /* thread-local text of the last failure */

/* ---- routing -------------------------------------------------------
 * Replaces the ATen op chain of extract_critical(), fast_dispatch.py:143-178:
 *   softmax (moe_layer.py:290) -> torch.topk (:146) -> one_hot (:150) -> gates (:151,173-175)
 *   -> k x fast_cumsum_sub_one (:159-171; GPU kernel custom_kernel.cpp:829-868)
 *   -> dispatch_count (:177-178) -> gshard_loss inputs (losses.py:12-19).
 *
 * Scratch: `ws` must hold tutel_amd_routing_workspace_bytes(T, E, k) bytes; gate_topk fills it
 * (per-tile expert histograms + per-tile score column sums), compute_location consumes it.
 */
size_t tutel_amd_routing_workspace_bytes(int T, int E, int k);

/* in[T,E] (dtype): scores, or logits when apply_softmax != 0 (softmax over E in fp32, rounded
 * to dtype, as F.softmax on a `dtype` tensor does).  Outputs:
 *   scores_out [T,E] dtype  (optional, may be NULL; only meaningful with apply_softmax)
 *   idx   [k,T] int32  descending score; EXACTLY equal scores in the order the reference's CPU torch.topk returns them
 *                      (TUTEL_OPT_TIE_RULE below; 0 there = lowest expert index first)
 *   gates [k,T] dtype  scores[t, idx_k[t]], divided by clamp(sum_k, eps(dtype)) when
 *                      normalize_gate != 0 and k > 1, each step rounded in `dtype` exactly as
 *                      fast_dispatch.py:151,173-175 does.
 * Limits: 1 <= k <= min(E, 16), E <= 4096, k * E <= 8192 (per-tile expert histograms live in LDS).
 * clear_map / clear_n (optional, NULL / 0): an int32 array this launch also fills with -1 -- pass
 * the slot_map that the following tutel_amd_compute_location builds (slot_map_cleared = 1) to
 * save the separate fill launch. */
int tutel_amd_gate_topk(const void *in, int dtype, int apply_softmax, int T, int E, int k,
                        int normalize_gate, void *scores_out, int32_t *idx, void *gates, void *ws,
                        size_t ws_bytes, int32_t *clear_map, int clear_n, tutel_stream_t stream);

/* The gate projection of a 16-bit linear gate, logits = x @ wg^T (x [T, M], wg [E, M] = nn.Linear's weight; replaces
 * `F.linear(x, self.wg.weight)`, tutel/gates/top.py:20-22), as a split-K MFMA kernel that leaves `splits` fp32 partial sums
 * partials[s][t][e]; tutel_amd_gate_topk_partials adds them in split order, rounds once to `dtype` (= what a 16-bit
 * F.linear with fp32 accumulation returns; optionally stored to logits_out [T, E]) and continues exactly as
 * tutel_amd_gate_topk with apply_softmax = 1 on those logits.  No atomics: the logits are reproducible bit for bit.
 * tutel_amd_gate_proj_splits: the split count for a shape -- a pure function of (T, M, E, dtype); 0 = shape not covered
 * (dtype not fp16 / bf16, E > 128, E % 4 != 0, M % 64 != 0): project with a library GEMM and call tutel_amd_gate_topk.
 * partials must hold splits * T * E floats.  tutel_amd_gate_proj returns TUTEL_AMD_ENOTSUP for a shape that is not covered. */
int tutel_amd_gate_proj_splits(int T, int M, int E, int dtype);
int tutel_amd_gate_proj(const void *x, const void *wg, int dtype, int T, int M, int E, float *partials,
                        size_t partial_bytes, tutel_stream_t stream);
int tutel_amd_gate_topk_partials(const float *partials, int splits, int dtype, int T, int E, int k,
                                 int normalize_gate, void *logits_out, void *scores_out, int32_t *idx, void *gates,
                                 void *ws, size_t ws_bytes, int32_t *clear_map, int clear_n, tutel_stream_t stream);

/* Warm the memory-side cache (256 MiB Infinity Cache) with n_chunks byte ranges of chunk_bytes each, stride_bytes apart: plain
 * loads, nothing written (sink4: 4 writable bytes or NULL; practically never written).  For a caller that has a second stream
 * idle while the routing kernels run -- they are latency chains that leave HBM idle, and the first expert GEMM's weights depend
 * on nothing (tools/r5_headline_ab.py holds the measurement).  blocks < n_chunks: 256. */
int tutel_amd_cache_warm(const void *p, size_t chunk_bytes, int n_chunks, size_t stride_bytes, int blocks, void *sink4,
                         tutel_stream_t stream);

/* idx[k,T] -> loc[k,T] (stable rank of token t among tokens with the same k-th choice, queued
 * after ALL tokens' earlier choices -- fast_dispatch.py:159-171), dispatch_count[E] (:177-178),
 * stats[0] = max_e dispatch_count[e] (the dropless capacity before the all-reduce, :192),
 * l_aux[0] = gshard loss (losses.py:12-19; NULL to skip), computed in fp32 and stored as one
 * element of l_aux_dtype (the reference returns it in the scores dtype).
 * hist_ready != 0: `ws` was filled by tutel_amd_gate_topk for the same (T,E,k) problem;
 * hist_ready == 0: idx comes from elsewhere, the histograms are rebuilt here first (l_aux is
 *                  then unavailable and must be NULL).
 * capacity > 0 additionally builds slot_map[E*capacity] (see tutel_amd_slot_map); pass
 * capacity <= 0 / slot_map NULL when the capacity is not known yet (capacity_factor <= 0).
 * slot_map_cleared != 0: the caller already filled slot_map with -1 (see tutel_amd_gate_topk). */
int tutel_amd_compute_location(const int32_t *idx, int T, int E, int k, int hist_ready, void *ws,
                               size_t ws_bytes, int32_t *loc, int32_t *dispatch_count,
                               int32_t *stats, void *l_aux, int l_aux_dtype, int capacity,
                               int32_t *slot_map, int slot_map_cleared, tutel_stream_t stream);

/* slot_map[E*C]: for bucket row (e*C + c) the flat (choice,token) index j*T + t routed there,
 * or -1 for an empty row.  Inverse of (idx, loc) restricted to loc < C && 0 <= idx < E -- the
 * condition of the reference dispatch kernels, sparse.py:28-33 / custom_kernel.cpp:294.
 * Needed because fast_encode here is bucket-major (one pass, writes the zero rows itself)
 * instead of memset + k token-major launches (fast_dispatch.py:26-28). */
int tutel_amd_slot_map(const int32_t *idx, const int32_t *loc, int T, int E, int k, int capacity,
                       int32_t *slot_map, tutel_stream_t stream);

/* Replaces torch.ops.tutel_ops.cumsum (custom_kernel.cpp:822-872) behind
 * tutel.jit_kernels.gating.fast_cumsum_sub_one (gating.py:19-24): per-column inclusive
 * cumsum - 1 of an int32 [T,E] array. */
int tutel_amd_cumsum_sub_one(const int32_t *mask, int32_t *out, int T, int E,
                             tutel_stream_t stream);

/* ---- dispatch / combine --------------------------------------------
 * fast_encode.  Replaces GatingEncoder.forward (fast_dispatch.py:18-29): torch.zeros +
 * k launches of the `forward` kernel (sparse.py:21-35 | custom_kernel.cpp:293-300):
 *   out[e*C + c, :] = g * x[t, :]  for the (j,t) in slot_map, g = gates[j,t] or 1 (gates NULL,
 *   i.e. is_postscore=True); every other row = 0.  Math in fp32, one rounding to `dtype`
 *   (fast_dispatch.py:95-96,126).  x [T,M], out [n_slots,M] in `dtype`; gates [k,T] gate_dtype.
 * Bucket order of `out` (the same three layouts tutel_amd_fast_decode reads; slot_map itself is
 * always in plain order e*C + l): chunk_rows = expert_slice = 0 -> plain [E, C, M] (capacity,
 * num_experts, ep_world ignored); chunk_rows = c > 0 -> chunk-major [C/c, E, c, M]; expert_slice =
 * s > 0 -> expert-sliced [E_loc/s, W, s, C, M] (W = ep_world).  For the last two n_slots must equal
 * num_experts * capacity. */
int tutel_amd_fast_encode(const void *x, int dtype, const int32_t *slot_map, const void *gates,
                          int gate_dtype, int T, int M, int n_slots, int capacity, int num_experts,
                          int chunk_rows, int expert_slice, int ep_world, void *out,
                          tutel_stream_t stream);

/* fast_decode.  Replaces GatingDecoder.forward (fast_dispatch.py:52-66): k launches of the
 * `backward_data` kernel (sparse.py:42-64 | custom_kernel.cpp:301-312) each writing a [T,M]
 * fp32 temp, summed left to right, then cast (:132):
 *   out[t,:] = sum_j g_j[t] * buf[idx_j[t]*C + loc_j[t], :]   over j with loc < C && idx >= 0
 * products and sums in fp32 in the reference's order (no FMA contraction), one rounding.
 * gates NULL = all ones (is_postscore=False). Also serves GatingEncoder.backward (:31-38).
 * chunk_rows = 0: buf is the plain [E, C, M] bucket array (row e*C + l).
 * chunk_rows = c > 0 (c divides C): buf is CHUNK-MAJOR [C/c, E, c, M] -- the layout in which the
 *   overlapped all-to-all (overlap.py: capacity split into a2a_ffn_overlap_degree chunks) delivers
 *   the expert outputs, so no torch.cat copy is needed: row ((l/c)*num_experts + e)*c + l%c.
 * expert_slice = s > 0 (chunk_rows must be 0; s divides num_experts / ep_world): buf is
 *   EXPERT-SLICED [E_loc/s, W, s, C, M] with W = ep_world, E_loc = num_experts / W -- the layout of
 *   the overlapped all-to-all that pipelines over groups of s local experts (each expert's weights
 *   are then streamed once and its GEMM sees all W*C rows): global expert e = w*E_loc + el lives at
 *   bucket ((el/s)*W + w)*s + el%s.  expert_slice = 0: ep_world is ignored. */
int tutel_amd_fast_decode(const void *buf, int dtype, const int32_t *idx, const int32_t *loc,
                          const void *gates, int gate_dtype, int T, int M, int k, int capacity,
                          int num_experts, int chunk_rows, int expert_slice, int ep_world, void *out,
                          tutel_stream_t stream);

/* Gate gradient (backward only).  Replaces the `backward_gate` kernel
 * (sparse.py:71-133 | custom_kernel.cpp:313-322):
 *   ggate[j,t] = sum_m buf[idx_j[t]*C + loc_j[t], m] * x[t, m]   (0 when dropped), fp32 out. */
int tutel_amd_gate_grad(const void *x, const void *buf, int dtype, const int32_t *idx,
                        const int32_t *loc, int T, int M, int k, int capacity, float *ggate,
                        tutel_stream_t stream);

/* ---- expert FFN grouped GEMM -------------------------------------------
 * One launch computes, for every local expert e and row r < R:
 *     D[e, r, :] = act( A[e, r, :] @ op(W[e]) + bias[e, :] )         (fp32 accumulate on MFMA)
 * Replaces each torch.matmul (+ torch.add bias, + activation) of FusedExpertsNetwork.forward,
 * experts/ffn.py:114-120, and -- with row_counts -- torch.ops.tutel_ops.sparse_bmm_infer
 * (custom_kernel.cpp:874-889, ffn.py:70-81) without its host loop / .cpu() sync.
 *
 *   w_kmajor = 1: W[e] is [N, K] row-major (batched_fc1_w [E_loc,H,M], ffn.py:26):  x @ W^T
 *   w_kmajor = 0: W[e] is [K, N] row-major (batched_fc2_w [E_loc,H,M_out], :27):    x @ W
 *
 * Row addressing folds the expert-parallel permutes (communicate.py:606-622) into the GEMM:
 * row r of expert e lives at
 *     A + e*a_stride_e + (r / a_rows_per_w)*a_stride_w + (r % a_rows_per_w)*lda      (elements)
 * so A may be the raw all-to-all output [W, E_loc, C, M] (a_rows_per_w = C) and D the raw
 * all-to-all input, with no .contiguous() copy in between.  For a plain [E_loc, R, K] array use
 * a_rows_per_w = R, a_stride_w = 0.
 *
 * row_counts (device int32[E_loc], may be NULL): rows >= ceil(count/row_align)*row_align of
 * expert e are skipped and left unwritten, exactly the rows sparse_bmm_infer leaves
 * uninitialised (custom_kernel.cpp:878-886).
 *
 * dtype: TUTEL_BF16 or TUTEL_F16 (fp32/fp64 experts stay on the ATen/rocBLAS bmm).
 * Requirements: K % 64 == 0, N % 8 == 0, lda/ldw/ldd and strides % 8 == 0 (16-byte rows). */
int tutel_amd_expert_gemm(const void *A, int64_t a_stride_e, int64_t a_stride_w, int a_rows_per_w,
                          int lda, const void *W, int w_kmajor, int64_t w_stride_e, int ldw,
                          const void *bias, int64_t bias_stride_e, void *D, int64_t d_stride_e,
                          int64_t d_stride_w, int d_rows_per_w, int ldd, int E_loc, int R, int N,
                          int K, int dtype, int act, const int32_t *row_counts, int row_align,
                          tutel_stream_t stream);

/* fast_encode fused into the first expert GEMM (is_postscore=True, single rank): row r of expert e
 * is gathered from the TOKEN array instead of a materialised bucket array,
 *     A[e, r, :] = X[slot_map[e*R + r] % T, :]      (slot_map < 0: the all-zero row `zero_row`, >= K elements)
 * which is exactly what fast_encode would have written (GatingEncoder.forward with unit gates,
 * fast_dispatch.py:18-29): same values, minus the 2 x E*C*M bytes of writing and re-reading the
 * buckets.  Other arguments as tutel_amd_expert_gemm (D is a plain [E_loc, R, N] array). */
int tutel_amd_expert_gemm_gather(const void *X, int ldx, const int32_t *slot_map, int T,
                                 const void *zero_row, const void *W, int w_kmajor,
                                 int64_t w_stride_e, int ldw, const void *bias,
                                 int64_t bias_stride_e, void *D, int64_t d_stride_e, int ldd,
                                 int E_loc, int R, int N, int K, int dtype, int act,
                                 const int32_t *row_counts, int row_align, tutel_stream_t stream);

/* The whole expert FFN in ONE persistent launch (round 6; csrc/expert_ffn.hip) -- OPT-IN: TUTEL_OPT_FFN_FUSED = 1.
 *     hid[e] = act(A[e] @ W1[e]^T + b1[e]),   D[e] = hid[e] @ W2[e]^T + b2[e]        e < E_loc, R rows per expert
 * -- what FusedExpertsNetwork.forward computes with two batched matmuls, two bias adds and the activation
 * (tutel/experts/ffn.py:114-120), and what two calls of tutel_amd_expert_gemm compute here: the same tiles in the same k order
 * (bit-identical results), handed out by a device-side ticket to one resident workgroup per CU, an expert's fc2 tiles waiting for
 * its fc1 tiles through a per-expert counter.  Measured at the headline shape: 224 us against 209.8 us for the two launches (the
 * regime is bound by HBM + Infinity Cache bandwidth, which the two launches already saturate; the file comment has the numbers), so
 * with the option at its default EVERY call answers TUTEL_AMD_ENOTSUP and the callers run the two launches.
 *   A    [E_loc, R, M] (x_stride_e elements between experts, rows ldx apart) -- or, with slot_map != NULL, the TOKEN array X [T, M]
 *        gathered through slot_map [E_loc * R] as in tutel_amd_expert_gemm_gather (fast_encode fused; zero_row: >= M zeros)
 *   W1   [E_loc, H, M] k-major (batched_fc1_w as stored), b1 [E_loc, H] or NULL
 *   hid  [E_loc, R, H] scratch (written and read by the launch)
 *   W2   [E_loc, M_out, H] k-major (the eval-mode copy of batched_fc2_w), b2 [E_loc, M_out] or NULL
 *   D    [E_loc, R, M_out]
 * Covered: R <= 128 rows per expert (every row of an expert in one M-tile: the weight-streaming regime), H and M_out >= 256,
 * M and H multiples of 64, enough tiles to cover the chip (E_loc * ceil(N / 256) >= 256 for both GEMMs), 16-byte aligned rows.
 * Anything else returns TUTEL_AMD_ENOTSUP with NOTHING launched -- the caller then issues the two tutel_amd_expert_gemm launches.
 * The first call on a stream allocates a few hundred bytes of control words (not allowed while that stream is being captured:
 * ENOTSUP then; run one eager call first). */
int tutel_amd_expert_ffn(const void *A, int64_t x_stride_e, int ldx, const int32_t *slot_map, int T, const void *zero_row,
                         const void *W1, int64_t w1_stride_e, int ldw1, const void *b1, int64_t b1_stride_e, void *hid,
                         int64_t hid_stride_e, int ldh, const void *W2, int64_t w2_stride_e, int ldw2, const void *b2,
                         int64_t b2_stride_e, void *D, int64_t d_stride_e, int ldd, int E_loc, int R, int M, int H, int M_out,
                         int dtype, int act, tutel_stream_t stream);

/* Gated (GLU) form: D = act(A @ op(W) + bias) * G, elementwise, rounded once to `dtype`.
 * Replaces the matmul + elementwise product of the SwiGLU expert,
 * `y = activation_fn(y1) * y2` with y2 = x @ W_fc2 (experts/llama_ffn.py:38-40): the caller first
 * runs tutel_amd_expert_gemm with act = silu on W_fc1 to get G = act(y1), then this entry point on
 * W_fc2 with act = none.  G has D's layout (same strides); G == D (in place) is allowed.
 * All other arguments as tutel_amd_expert_gemm. */
int tutel_amd_expert_gemm_glu(const void *A, int64_t a_stride_e, int64_t a_stride_w,
                              int a_rows_per_w, int lda, const void *W, int w_kmajor,
                              int64_t w_stride_e, int ldw, const void *bias, int64_t bias_stride_e,
                              const void *G, void *D, int64_t d_stride_e, int64_t d_stride_w,
                              int d_rows_per_w, int ldd, int E_loc, int R, int N, int K, int dtype,
                              int act, const int32_t *row_counts, int row_align,
                              tutel_stream_t stream);

/* ---- expert-parallel pipeline ----------------------------------------
 * Replaces the reference's native all-to-all layer: the private NCCL communicator
 * (get_nccl_unique_id / init_nccl, custom_kernel.cpp:341-365), its stream + event table (:327-338,
 * :433-461) and the asynchronous all-to-all scatter / gather the overlap path is built from
 * (:520-654; driven from tutel/impls/overlap.py:8-67 one chunk at a time), plus the two
 * permute+contiguous copies around the experts (communicate.py:606-622), which disappear here.
 *
 * A communicator owns an RCCL comm (resolved with dlopen from the RCCL already in the process), one
 * side stream and a table of events.  Bootstrap like the reference: one rank calls
 * tutel_amd_ep_unique_id, the host code broadcasts the bytes (torch.distributed), every rank calls
 * tutel_amd_ep_comm_create with its device current.  One process per GPU. */
typedef struct tutel_amd_ep_comm tutel_amd_ep_comm_t;
#define TUTEL_AMD_EP_ID_BYTES 128

int tutel_amd_ep_load_rccl(const char *path_hint /* may be NULL */);
int tutel_amd_ep_unique_id(void *out, size_t bytes /* >= TUTEL_AMD_EP_ID_BYTES */);
int tutel_amd_ep_comm_create(const void *id, size_t bytes, int world, int rank, tutel_amd_ep_comm_t **out);
int tutel_amd_ep_comm_destroy(tutel_amd_ep_comm_t *comm);
/* Bring-up / test communicator: the equal-split exchange is performed by a HOST callback instead of RCCL (e.g. staged
 * through host memory over a gloo process group, which is how several ranks can share one GPU).  The callback is invoked
 * synchronously from tutel_amd_ep_forward / tutel_amd_ep_all_to_all on the calling thread; it must order itself after the
 * work already enqueued on the caller's current stream and return only when `recv` may be read by work enqueued after it.
 * Everything else -- stage layouts, buffers, both streams, events -- is the production path. */
typedef int (*tutel_amd_exchange_fn)(void *user, const void *send, void *recv, size_t bytes_per_peer, int world);
int tutel_amd_ep_comm_create_hosted(int world, int rank, tutel_amd_exchange_fn fn, void *user, tutel_amd_ep_comm_t **out);
int tutel_amd_ep_comm_info(const tutel_amd_ep_comm_t *comm, int *world, int *rank);

/* ---- IPC transport: the exchange as peer stores over xGMI, no collective on the hot path ----------------------------------
 * Replaces the exchange kernels of the reference's asynchronous all-to-all (custom_kernel.cpp:520-654: one ncclSend / ncclRecv
 * pair per peer and chunk inside ncclGroupStart / End, :559-579, :627-648) and its event hand-offs (:553, :648).  The GPUs of a
 * node address each other's HBM (hipIpc*), so the kernels that PRODUCE the exchanged rows store them where the all-to-all would
 * have put them: fast_encode into the receive array of the rank that owns the expert, the second expert GEMM into the return
 * array of the rank the row came from.  One 32-bit flag per (direction, stage, peer), written by a one-workgroup kernel after
 * the producer and polled by a one-workgroup kernel before the consumer, is the only synchronisation; epochs are counted in
 * device memory, so a captured forward replays from a HIP graph.  Works between processes that share ONE device as well
 * (how the tests run it with 2 and 4 ranks on a 1-GPU box).
 *
 * A segment is device memory of this rank that every peer maps: allocate, all-gather the TUTEL_AMD_IPC_HANDLE_BYTES-byte
 * handles with the host's process group (like the RCCL id), open.  flag_memory != 0: uncached / fine-grained memory for the
 * communicator's flag words (tutel_amd_ep_flag_bytes() of them, zeroed).  tutel_amd_ep_segment_ptr: base of `peer`'s segment in
 * this process (peer < 0: the local base). */
typedef struct tutel_amd_ep_segment tutel_amd_ep_segment_t;
#define TUTEL_AMD_IPC_HANDLE_BYTES 64
int tutel_amd_ep_segment_alloc(size_t bytes, int flag_memory, tutel_amd_ep_segment_t **out, void *handle_out, size_t handle_bytes);
int tutel_amd_ep_segment_open(tutel_amd_ep_segment_t *seg, int world, int rank, const void *handles, size_t handle_bytes);
void *tutel_amd_ep_segment_ptr(const tutel_amd_ep_segment_t *seg, int peer);
/* bytes [off, off + bytes) of the local segment -> dst (device memory), enqueued on `stream` */
int tutel_amd_ep_segment_read(const tutel_amd_ep_segment_t *seg, size_t off, void *dst, size_t bytes, tutel_stream_t stream);
int tutel_amd_ep_segment_free(tutel_amd_ep_segment_t *seg);
size_t tutel_amd_ep_flag_bytes(void);
/* a communicator whose only exchange is the IPC transport (no RCCL, no callback): at most 16 ranks of one node */
int tutel_amd_ep_comm_create_ipc(int world, int rank, tutel_amd_ep_comm_t **out);
/* gives any communicator (RCCL-backed, hosted or IPC-only) the IPC transport: `flags` = an opened flag segment of this
 * communicator's ranks, owned by the caller and kept alive as long as the communicator; timeout_ms bounds every wait for a
 * peer (<= 0: 120 s, the order of a collective watchdog) -- a wait that times out records which peer and stage never arrived, the forward's output is then
 * garbage and the NEXT call on the communicator fails with that text (tutel_amd_ep_ipc_status reads it without a call). */
int tutel_amd_ep_comm_attach_ipc(tutel_amd_ep_comm_t *comm, tutel_amd_ep_segment_t *flags, int timeout_ms);
int tutel_amd_ep_comm_has_ipc(const tutel_amd_ep_comm_t *comm);
int tutel_amd_ep_ipc_set_timeout(tutel_amd_ep_comm_t *comm, int timeout_ms); /* for the waits enqueued from now on */
int tutel_amd_ep_ipc_status(tutel_amd_ep_comm_t *comm);
/* all_to_all_single with equal splits over the IPC transport: block r of `send` (any device buffer, bytes_per_peer bytes,
 * multiple of 16) lands at byte offset recv_off + <my rank> * bytes_per_peer of rank r's `seg`; when the call's work on
 * `stream` completes, this rank's world blocks have arrived.  Two exchanges into the same offset must be kept apart by the
 * caller (there are no credits). */
int tutel_amd_ep_ipc_exchange(tutel_amd_ep_comm_t *comm, tutel_amd_ep_segment_t *seg, const void *send, size_t bytes_per_peer,
                              size_t recv_off, tutel_stream_t stream);
/* Payload-sized self-check of the transport, device-paced (no host synchronisation between its passes).  Every pass: a writer
 * kernel stores a tagged pattern of bytes_per_peer bytes into block <my rank> of every rank's `seg` (flavour 0 = plain stores,
 * what the pipeline uses; 1 = sc1, 2 = sc0 sc1 write-through, 3 = non-temporal: for probing), signal, wait (+ epoch canaries), a
 * reader kernel compares every 16-byte vector of this rank's `world` blocks and counts mismatches, then an acknowledgement in the
 * other direction lets the peers overwrite the blocks in the next pass.  side_stream != 0: wait + reader run on a side stream
 * of the communicator, as the stage GEMMs of the overlapped pipeline do.  mismatch: DEVICE array of two uint64 -- [0] vectors
 * that differed over all passes, [1] the first offender (source rank << 40 | vector index; all ones = none); read it after
 * synchronising `stream`.  Collective: every rank calls it with the same arguments.  Replaces nothing in the reference (its
 * exchange is NCCL's); it is the evidence the peer-store transport asks for before it is trusted on a node. */
int tutel_amd_ep_ipc_selfcheck(tutel_amd_ep_comm_t *comm, tutel_amd_ep_segment_t *seg, size_t bytes_per_peer, int passes, int flavour,
                               int side_stream, unsigned long long *mismatch, tutel_stream_t stream);

/* all_to_all_single with equal splits (simple_all_to_all, communicate.py:181-192): block r of `send`
 * (bytes_per_peer bytes) goes to rank r and lands as block <my rank> of its `recv`; enqueued on `stream`. */
int tutel_amd_ep_all_to_all(tutel_amd_ep_comm_t *comm, const void *send, void *recv, size_t bytes_per_peer,
                            tutel_stream_t stream);

/* Variable-size exchanges on the same communicator (tutel.net.batch_all_to_all_v / batch_all_gather_v; reference:
 * custom_kernel.cpp:463-491 and :493-518, one grouped ncclSend / ncclRecv loop per tensor on the shared communicator).
 *   all_to_all_v: send_bytes[r] bytes, taken from `send` at the running offset, go to rank r; recv_bytes[r] bytes from rank
 *                 r land in `recv` at the running offset.  The caller has exchanged the sizes (the one host sync the
 *                 reference API implies, communicate.py:225-241).
 *   all_gather_v: my recv_bytes[<my rank>] bytes at `send` go to every rank; rank r's recv_bytes[r] bytes land in `recv`
 *                 at the running offset (rank order).
 * Byte counts are host arrays of `world` entries; zero-byte pairs are skipped on both sides.  Enqueued on `stream`.
 * A hosted communicator performs them with the callback registered by tutel_amd_ep_comm_set_hosted_v. */
typedef int (*tutel_amd_exchange_v_fn)(void *user, const void *send, void *recv, const uint64_t *send_bytes,
                                       const uint64_t *send_offsets, const uint64_t *recv_bytes, int world);
int tutel_amd_ep_comm_set_hosted_v(tutel_amd_ep_comm_t *comm, tutel_amd_exchange_v_fn fn);
int tutel_amd_ep_all_to_all_v(tutel_amd_ep_comm_t *comm, const void *send, void *recv, const uint64_t *send_bytes,
                              const uint64_t *recv_bytes, tutel_stream_t stream);
int tutel_amd_ep_all_gather_v(tutel_amd_ep_comm_t *comm, const void *send, void *recv, const uint64_t *recv_bytes,
                              tutel_stream_t stream);

/* Stage layouts of the pipeline (identical to tutel_amd/impls/overlap.py::OverlapPlan; a CPU test pins it):
 *   sliced  (allow_sliced && E_loc % degree == 0): stages = groups of E_loc/degree local experts, buckets
 *           laid out [degree, W, s, C]; every expert's weights are streamed once per forward;
 *   chunked (otherwise; the reference's scheme, overlap.py:21-24): stages = capacity chunks of C/degree rows,
 *           buckets laid out [degree, E, c].
 * Stage i's message is [W, rows] bucket rows (dim 0 = peer); the GEMMs address the received rows as
 * experts_per_stage experts x gemm_rows source-rank-major rows: (stride_e, stride_w, rows_per_w, ld) =
 * (chunk*ld, rows*ld, chunk, ld). */
typedef struct {
  int sliced;            /* 1 expert-sliced, 0 capacity-chunked */
  int experts_per_stage; /* s */
  int chunk;             /* c: capacity rows per expert and stage */
  int rows;              /* s*c: bucket rows per (stage, rank) block */
  int gemm_rows;         /* W*c: GEMM rows per expert and stage */
} tutel_amd_ep_plan_t;
int tutel_amd_ep_plan(int num_experts, int world, int capacity, int degree, int allow_sliced, tutel_amd_ep_plan_t *out);

/* One call = fast_encode -> all-to-all -> expert FFN -> all-to-all -> fast_decode for one batch of tokens
 * whose routing is known (tutel_amd_gate_topk + tutel_amd_compute_location): what MOELayer.forward does
 * between extract_critical and the final reshape (moe_layer.py:327-361), with a2a_ffn_overlap_degree stages
 * pipelined over the caller's stream (RCCL all-to-alls, encode, decode) and the communicator's side stream (the
 * stage GEMMs); capturable in a HIP graph with the caller's stream as origin.  Returns after enqueueing;
 * when it returns, the caller's stream is ordered after every operation of the call (buffers may be reused or
 * freed in stream order).  comm == NULL: world must be 1, the exchange is a copy; with fuse_encode and
 * is_postscore the first GEMM then gathers its rows from the tokens and enc / recv / back are not touched. */
typedef struct {
  /* sizes */
  int T, M, H, M_out;            /* tokens of this rank, model dim, hidden size per expert, output dim */
  int num_experts, world, k;     /* GLOBAL experts (E_loc = num_experts / world local ones), ranks, top-k */
  int capacity;                  /* C, already aligned to the degree (moe_layer.py:298-301) */
  int degree;                    /* a2a_ffn_overlap_degree, 1..32 */
  int allow_sliced;              /* see tutel_amd_ep_plan */
  int dtype, gate_dtype, act;    /* TUTEL_BF16 | TUTEL_F16; dtype of `gates`; TUTEL_ACT_* fused into fc1 */
  int is_postscore;              /* gates applied in decode (1) or encode (0), fast_dispatch.py:125,131 */
  int w2_kmajor;                 /* w2 is [E_loc, M_out, H] (1) or the checkpoint layout [E_loc, H, M_out] (0) */
  int fuse_encode;               /* single rank: gather fc1's rows from the tokens (needs zero_row) */
  /* routing of this batch (device) */
  const void *x;                 /* [T, M] */
  const int32_t *slot_map;       /* [num_experts * C] (tutel_amd_compute_location) */
  const int32_t *idx, *loc;      /* [k, T] */
  const void *gates;             /* [k, T] gate_dtype */
  /* local experts (device): batched_fc1_w [E_loc,H,M], bias [E_loc,H]; batched_fc2_w, bias [E_loc,M_out] (NULL = none) */
  const void *w1, *b1, *w2, *b2;
  /* workspace (device, dtype): enc / recv [num_experts*C, M]; hid [E_loc*world*C, H]; send / back [num_experts*C, M_out] */
  void *enc, *recv, *hid, *send, *back;
  const void *zero_row;          /* >= M zero elements (fuse_encode only) */
  const int32_t *row_counts;     /* dropless / megablocks (moe_layer.py:278-280: single rank only): per-expert row counts, */
  int row_align;                 /*   rows >= ceil(count/row_align)*row_align are skipped by both GEMMs; NULL / 1 = all rows */
  void *y;                       /* out: [T, M_out] */
  /* IPC transport (see tutel_amd_ep_segment_alloc): the opened segment that holds `recv` and `back` at the same offsets on
   * every rank.  Then fast_encode stores into the peers' `recv`, the second GEMM into the peers' `back`, `enc` / `send` are
   * not touched and no collective is enqueued.  NULL: the exchange is the communicator's (RCCL / host callback / copy). */
  const struct tutel_amd_ep_segment *peer_seg;
} tutel_amd_ep_args_t;
int tutel_amd_ep_forward(tutel_amd_ep_comm_t *comm, const tutel_amd_ep_args_t *args, tutel_stream_t stream);

/* Routing + the pipeline above in ONE call: what MOELayer.forward does after the gate projection (moe_layer.py:290-361)
 * for the common inference configuration -- softmax + top-k + locations + gshard loss (tutel_amd_gate_topk,
 * tutel_amd_compute_location with the capacity known up front, capacity_factor > 0) and then tutel_amd_ep_forward on
 * the routing just computed.  ep.slot_map / idx / loc / gates are OUTPUT buffers here ([E*C], [k,T], [k,T], [k,T] in
 * the logits dtype); ep.gate_dtype is ignored (= logits_dtype).  Dropless routing: see the last fields. */
typedef struct {
  tutel_amd_ep_args_t ep;
  const void *logits;        /* [T, num_experts] gate logits */
  int logits_dtype;          /* TUTEL_F32 | TUTEL_F16 | TUTEL_BF16 */
  int normalize_gate;
  void *ws;                  /* tutel_amd_routing_workspace_bytes(T, E, k) bytes */
  size_t ws_bytes;
  int32_t *dispatch_count;   /* out [num_experts] */
  int32_t *stats;            /* out [1] max expert load (may be NULL; required for dropless) */
  void *l_aux;               /* out [1], logits dtype (NULL to skip the loss) */
  /* dropless routing (capacity_factor <= 0, fast_dispatch.py:191-199), single rank only: set ep.capacity = 0.  The call then
   * reads the maximum expert load back (the ONE host synchronisation the reference's API implies, `int(capacity)`), clamps it
   * to capacity_limit (> 0: k * int(-capacity_factor * samples_per_expert); 0: none), rounds it up to `alignment`, and runs
   * the rest with that capacity.  The workspace must hold max_capacity rows per expert; if the capacity exceeds it nothing
   * further is enqueued and the call returns TUTEL_AMD_EAGAIN with the needed value in *capacity_out (grow and call again). */
  int capacity_limit, alignment, max_capacity;
  int *capacity_out;         /* host pointer, out: the capacity used (may be NULL when ep.capacity > 0); dropless: the read-back
                              * lands here (pinned memory keeps the copy asynchronous) */
  /* the gate projection inside the call (round 5): logits == NULL and gate_w != NULL -> logits = ep.x @ gate_w^T through
   * tutel_amd_gate_proj + tutel_amd_gate_topk_partials (ep.dtype must equal logits_dtype; the shape must be covered, see
   * tutel_amd_gate_proj_splits -- otherwise the call fails before anything is enqueued). */
  const void *gate_w;        /* [num_experts, M] in ep.dtype (nn.Linear weight of the gate), or NULL */
  float *gate_partials;      /* >= splits * T * num_experts floats */
  size_t gate_partial_bytes;
  void *logits_out;          /* optional out [T, num_experts], logits dtype: the projected logits (NULL to skip) */
  /* fused location (round 5; TUTEL_OPT_FUSED_LOCATION): scratch of >= round_up(k * T, 16) bytes.  With it, on one rank with
   * capacity > 0, tutel_amd_compute_location is not launched where the shape allows: the first expert GEMM ranks its expert's
   * (choice, token) entries itself and fills ep.loc / ep.slot_map; dispatch_count, stats and l_aux come out of an extra block of the
   * decode launch.  Every output keeps its bits.  NULL: never. */
  void *fl_ws;
  size_t fl_ws_bytes;
} tutel_amd_moe_args_t;
#define TUTEL_AMD_EAGAIN 1000
#define TUTEL_AMD_ENOTSUP 1001 /* reserved: "this entry point does not take the shape, nothing was launched" */
int tutel_amd_moe_forward(tutel_amd_ep_comm_t *comm, const tutel_amd_moe_args_t *args, tutel_stream_t stream);

/* stage markers: roctx ranges (rocprofv3 --marker-trace); the pipeline above emits tutel_amd.fast_encode /
 * all_to_all / expert_fc1 / expert_fc2 / fast_decode itself.  No-ops when libroctx64 is not in the process
 * (set TUTEL_AMD_ROCTX=1 to load it).  The reference's only tracing is system.record_time (system.py:73-79). */
int tutel_amd_range_push(const char *name);
int tutel_amd_range_pop(void);

/* ---- per-stage timing (measurement only; bench.py) ------------------------------------------------
 * tutel_amd_stage_timing(1): from now on every entry point above brackets its kernel launch with a pair of HIP
 * timing events on the launch stream (skipped while that stream is being captured); (2): only the two expert GEMMs
 * (the dominant kernels: what a timed region can carry without being perturbed); (0): off.  tutel_amd_stage_report
 * waits for the recorded events, returns per-stage totals (microseconds) and launch counts, and clears the
 * records.  The reference's counterpart is system.record_time around whole steps (system.py:73-79). */
enum {
  TUTEL_STAGE_GATE_TOPK = 0, TUTEL_STAGE_LOCATION = 1, TUTEL_STAGE_ENCODE = 2, TUTEL_STAGE_FC1 = 3, TUTEL_STAGE_FC2 = 4,
  TUTEL_STAGE_DECODE = 5, TUTEL_STAGE_A2A_DISPATCH = 6, TUTEL_STAGE_A2A_COMBINE = 7, TUTEL_STAGE_OTHER = 8, TUTEL_STAGE_GATE_PROJ = 9,
  TUTEL_STAGE_COUNT = 10
};
int tutel_amd_stage_timing(int enable);
int tutel_amd_stage_report(double *total_us, int *counts, int n_stages /* >= TUTEL_STAGE_COUNT */);
/* step marks: tutel_amd_mark records one event on `stream` (device-scope release: a default HIP event writes the L2 back to
 * system scope before it takes its timestamp, ~5 us of idle GPU per record); tutel_amd_marks_report waits, returns the
 * deltas between consecutive marks in microseconds (at most n; return value = how many) and clears the list;
 * tutel_amd_marks_reserve(n) creates the events ahead of a timed region. */
int tutel_amd_mark(tutel_stream_t stream);
int tutel_amd_marks_reserve(int n);
int tutel_amd_marks_report(double *delta_us, int n);

/* ---- tuning knobs (A/B measurements and tests; never needed for correctness) ---------------------
 * value -1 = automatic (default; the environment variables TUTEL_AMD_GEMM_IMPL / TUTEL_AMD_GEMM_BIG / TUTEL_AMD_DECODE /
 * TUTEL_AMD_EP_STAGE_GRID / TUTEL_AMD_GEMM_PERSIST / TUTEL_AMD_EP_STREAMS / TUTEL_AMD_EP_CANARY / TUTEL_AMD_GEMM_SPLITK / TUTEL_AMD_GEMM_GATHER / TUTEL_AMD_FUSED_LOCATION / TUTEL_AMD_GEMM_STORE / TUTEL_AMD_TIE_RULE / TUTEL_AMD_FFN_FUSED seed it once), >= 0 = force.  Every choice computes bit-identical results (excepted: TUTEL_OPT_EP_CANARY = 2 provokes the error it tests; TUTEL_OPT_GEMM_SPLITK changes the fp32 summation order; TUTEL_OPT_TIE_RULE = 0 changes which of two EXACTLY equal scores is chosen).
 *   TUTEL_OPT_GEMM_IMPL  kernels of the <= 128-rows-per-expert regime: 0 register-staged 128 x 128, 1 LDS-DMA 128 x 128, 4 the
 *                        128 x 256 tile on a three-slot LDS-DMA ring (k-major weights; automatic when its grid covers the chip)
 *   TUTEL_OPT_GEMM_TILE  0 never use the 256-row tiles, 1 always the plain 256 x 256 kernel, 2 / 3 always 256 x 128
 *                        with a two- / three-slot LDS ring (k-major weights), 4 always the 256 x 256 ping-pong kernel
 *                        (automatic: > 128 rows per expert and enough tiles to cover the chip)
 *   TUTEL_OPT_DECODE     fast_decode launch shape: bit 0 = two waves per token, bit 1 = non-temporal stores of the output
 *                        (automatic: 2)
 *   TUTEL_OPT_EP_STAGE_GRID  stage GEMMs of an overlapped pipeline (degree > 1): 1 / automatic = the half-chip 256 x 256 grid where the
 *                        full one would be one workgroup per CU (two stages run side by side on the two side streams), 0 = the
 *                        grid the shape would take alone
 *   TUTEL_OPT_GEMM_PERSIST  256 x 256 ping-pong kernel, bias operand: 0 = fetched after the K loop, 1 / automatic = before it
 *                        (32 more live registers, its L2 round trip hidden behind the loop)
 *   TUTEL_OPT_EP_STREAMS overlapped pipeline: 1 = every stage's GEMMs on ONE side stream, 2 / automatic = stages alternate between
 *                        two side streams (the half-chip GEMM grids of two stages run side by side)
 *   TUTEL_OPT_GEMM_SPLITK  256 x 256 ping-pong kernel on launches of 96 .. 191 tiles (half the chip: a pipeline stage of an 8-way rank):
 *                        1 = two workgroups per tile, each over half of K, partials handed over through memory (the ONE choice that
 *                        changes result bits: two half sums added instead of one running sum; deterministic).  Measured slower than
 *                        the grids it was to replace (round 5), so 0 / automatic = never
 *   TUTEL_OPT_GEMM_GATHER  fused fast_encode of the ring kernels: 1 / automatic = the slot-map entries come through the scalar cache
 *                        and are waited for only after the first weight pieces have been issued; 0 = vector loads in front of them
 *                        (rounds 1-4)
 *   TUTEL_OPT_FUSED_LOCATION  tutel_amd_moe_forward on one rank: 1 / automatic = the locations are computed inside the first expert
 *                        GEMM (every block ranks its expert's entries itself, no tutel_amd_compute_location launch; dispatch_count and
 *                        the loss in an extra block of the decode launch) where the shape takes that kernel; 0 = never
 *   TUTEL_OPT_GEMM_STORE  output stores of the grouped GEMMs whose epilogue goes through LDS (the ring and ping-pong kernels): 1 / automatic =
 *                        write-through (sc0 sc1: the tile streams out while the kernel runs and stays valid in L2 for the next kernel),
 *                        0 = plain stores (the tile stays dirty in the XCD's L2 and what is left is written back when the kernel ends,
 *                        after the last wave: rounds 1-4), 2 = non-temporal; buffer stores through a descriptor, same values and addresses
 *   TUTEL_OPT_TIE_RULE   top-k among EXACTLY equal scores: 1 / automatic = the expert ids the reference's CPU path gets from torch.topk
 *                        (tutel/impls/fast_dispatch.py:146-148; ATen's nth_element / partial_sort over (value, index) pairs replayed per
 *                        tied row, csrc/topk_ties.h; every expert count the routing kernels take), 0 = descending score,
 *                        lowest expert index first (rounds 1-5)
 *   TUTEL_OPT_FFN_FUSED  the expert FFN where tutel_amd_expert_ffn covers the shape (one rank, <= 128 rows per expert, k-major fc2): 0 /
 *                        automatic = the two launches of rounds 1-5 (measured faster: 209.8 vs 224 us at the headline shape), 1 = one
 *                        persistent launch for fc1 -> activation -> fc2 (work queues per hardware XCC id), 2 = the same with queues by
 *                        blockIdx & 7, 3 = the same with the next ticket fetched after each item instead of inside it (A/B).  Same bits
 *   TUTEL_OPT_EP_CANARY  IPC transport: epoch canaries behind every exchanged block (1 / automatic = written by the producers and
 *                        checked by the wait kernels; 0 = off; 2 = TEST INJECTION: this rank publishes the previous epoch, as if
 *                        its rows had not landed when its flag did -- the peers must report it)
 */
#define TUTEL_OPT_GEMM_IMPL 0
#define TUTEL_OPT_GEMM_TILE 1
#define TUTEL_OPT_DECODE 2
#define TUTEL_OPT_EP_STAGE_GRID 3
#define TUTEL_OPT_GEMM_PERSIST 4
#define TUTEL_OPT_EP_STREAMS 5
#define TUTEL_OPT_EP_CANARY 6
#define TUTEL_OPT_GEMM_SPLITK 7
#define TUTEL_OPT_GEMM_GATHER 8
#define TUTEL_OPT_FUSED_LOCATION 9
#define TUTEL_OPT_GEMM_STORE 10
#define TUTEL_OPT_TIE_RULE 11
#define TUTEL_OPT_FFN_FUSED 12
#define TUTEL_OPT_COUNT 13
int tutel_amd_set_option(int key, int value);

/* ---- self-test helpers (used by tests / smoke only) ---------------------------------------
 * Dumps the lane->element permutation of ds_read_b64_tr_b16 (the transposing LDS read the
 * [K,N]-weight GEMM relies on): out[64*4] uint16, LDS pre-filled with lds[i] = i, lane l
 * reading 8 bytes at byte address l*8. */
int tutel_amd_probe_tr16(uint16_t *out, tutel_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TUTEL_AMD_H */
