// ep.hip -- the expert-parallel pipeline of the MoE forward behind ONE native call.
//
//   fast_encode -> all-to-all -> expert FFN (2 grouped GEMMs) -> all-to-all -> fast_decode
//
// The reference drives its overlapped exchange from C++ on a PRIVATE NCCL communicator with an event
// table and a pooled communication stream (tutel/custom/custom_kernel.cpp:341-365 unique id + init,
// :433-461 streams / events, :520-654 async scatter / gather, one Python call per chunk).  Round 1 of
// this repo drove the same pipeline from Python through torch.distributed: ~20 enqueues per forward
// and 0.32-0.46 ms of host time against 0.2 ms of GPU work per rank.  Here:
//   * an RCCL communicator of our own (ncclCommInitRank from an id the host code broadcasts), resolved
//     with dlopen from the RCCL already in the process (torch ships one): no link-time dependency;
//   * one call enqueues the whole pipeline on two HIP streams -- ncclAllToAll on the caller's stream, the
//     GEMMs of the overlapped stages on a side stream owned by the communicator -- with events from a
//     table created once; the call returns after enqueueing (no Python in between) and the whole call can
//     be captured in a HIP graph (the caller's stream is the capture origin, where RCCL can be captured);
//   * the stage layouts are those of tutel_amd/impls/overlap.py::OverlapPlan (expert-sliced when
//     a2a_ffn_overlap_degree divides the local expert count, capacity-chunked otherwise); the GEMMs
//     address the raw exchange buffers, so there are no permute copies (communicate.py:606-622) and no
//     torch.cat; tutel_amd_ep_plan() exposes the arithmetic so a CPU test pins it to the Python plan;
//   * degree 1 is the same plan with one stage, issued on the caller's stream alone;
//   * world size 1 without a communicator: the exchange is the identity (stage buffers alias), and
//     with is_postscore the first GEMM gathers its rows from the tokens (fused fast_encode).
// Every step is one of the C-ABI entry points of this library; this file adds orchestration only.
#include <dlfcn.h>
#include <stdlib.h>

#include <rccl/rccl.h>

#include "common.h"

// ---- RCCL, resolved at run time -------------------------------------------------------------
struct RcclApi {
  void *handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllToAll)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
};
static RcclApi g_rccl = {};

extern "C" int tutel_amd_ep_load_rccl(const char *path) {
  if (g_rccl.handle != nullptr) return 0;
  void *h = nullptr;
  // the copy already mapped into the process first (same HIP runtime as the caller's tensors), then the hint, then the loader path
  const char *sonames[] = {"librccl.so.1", "librccl.so"};
  for (const char *n : sonames)
    if (h == nullptr) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr && path != nullptr && path[0] != 0) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  for (const char *n : sonames)
    if (h == nullptr) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  TUTEL_REQUIRE(h != nullptr, "tutel_amd_ep_load_rccl: cannot load librccl (%s)", dlerror());
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.AllToAll = (decltype(a.AllToAll))dlsym(h, "ncclAllToAll");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
  a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
  a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
  TUTEL_REQUIRE(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllToAll && a.GetErrorString && a.GroupStart && a.GroupEnd && a.Send && a.Recv,
                "tutel_amd_ep_load_rccl: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllToAll / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
  g_rccl = a;
  return 0;
}

#define RCCL_CHECK(call, what)                                                          \
  do {                                                                                  \
    ncclResult_t r_ = (call);                                                           \
    if (r_ != ncclSuccess) {                                                            \
      tutel_set_error("%s: RCCL error %d (%s)", what, (int)r_, g_rccl.GetErrorString(r_)); \
      return (int)r_ ? (int)r_ : -1;                                                    \
    }                                                                                   \
  } while (0)
#define HIP_CHECK(call, what)                                                           \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      tutel_set_error("%s: %s", what, hipGetErrorString(e_));                           \
      return (int)e_;                                                                   \
    }                                                                                   \
  } while (0)

// stage markers: every C-ABI entry point opens a roctx range named after its stage (StageScope, api.hip), so the
// calls below show up as tutel_amd.fast_encode / all_to_all_* / expert_fc1 / expert_fc2 / fast_decode in a marker trace
struct Range {
  explicit Range(const char *name) { tutel_amd_range_push(name); }
  ~Range() { tutel_amd_range_pop(); }
};

// ---- communicator: RCCL comm + the communication stream + the event table -------------------------------
#define EP_MAX_SPLIT 32  // AllToAllStatus.max_num_split of the reference (custom_kernel.cpp:328)
struct tutel_amd_ep_comm {
  ncclComm_t comm;
  int world, rank, device;
  tutel_amd_exchange_fn hosted;  // bring-up / test communicator: the exchange is a host callback (comm == nullptr then)
  tutel_amd_exchange_v_fn hosted_v;
  void *hosted_user;

  // The GEMMs of the overlapped pipeline run on a side stream (the collectives on the caller's).  HIP multiplexes streams onto a
  // few hardware queues PER PRIORITY LEVEL, and two streams that share a queue run strictly one after the other: every fourth
  // normal-priority stream created in a process lands on the default stream's queue and would then never overlap with it
  // (measured on MI355X, tools/stream_concurrency_check.py, profiles/r03_stream_queues.txt).  Streams of different priority never
  // share a queue, so the communicator owns THREE side streams -- highest, lowest and normal priority -- and a call uses the two
  // whose priority differs from its caller's stream: stage i of the pipeline runs on side stream i % 2, so that the GEMMs of two
  // stages (each a half-chip grid of 128 workgroups at the expert-parallel shapes) run side by side.
  hipStream_t side_stream;      // highest priority
  hipStream_t side_stream_low;  // lowest priority
  hipStream_t side_stream_mid;  // normal priority (used only when the caller's stream is one of the other two)
  hipEvent_t recv_ev[EP_MAX_SPLIT], done_ev[EP_MAX_SPLIT];
};

static bool create_side_streams(tutel_amd_ep_comm *c) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
  return hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, greatest) == hipSuccess &&
         hipStreamCreateWithPriority(&c->side_stream_low, hipStreamNonBlocking, least) == hipSuccess &&
         hipStreamCreateWithPriority(&c->side_stream_mid, hipStreamNonBlocking, (least + greatest) / 2) == hipSuccess;
}

// the two side streams whose priority differs from the caller's (see struct tutel_amd_ep_comm); TUTEL_OPT_EP_STREAMS = 1: one
static void side_streams_for(tutel_amd_ep_comm *c, hipStream_t caller, hipStream_t out[2]) {
  int least = 0, greatest = 0, pr = 0;
  out[0] = c->side_stream;
  out[1] = c->side_stream_low;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && hipStreamGetPriority(caller, &pr) == hipSuccess && least != greatest) {
    if (pr == greatest) out[0] = c->side_stream_mid;
    else if (pr == least) out[1] = c->side_stream_mid;
  }
  if (tutel_get_option(TUTEL_OPT_EP_STREAMS) == 1) out[1] = out[0];
}

extern "C" int tutel_amd_ep_unique_id(void *out, size_t bytes) {
  TUTEL_REQUIRE(out != nullptr && bytes >= sizeof(ncclUniqueId), "tutel_amd_ep_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
  if (tutel_amd_ep_load_rccl(nullptr) != 0) return -1;
  ncclUniqueId id;
  RCCL_CHECK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, &id, sizeof(id));
  return 0;
}

extern "C" int tutel_amd_ep_comm_create(const void *id, size_t bytes, int world, int rank, tutel_amd_ep_comm_t **out) {
  TUTEL_REQUIRE(out != nullptr && world >= 1 && rank >= 0 && rank < world, "tutel_amd_ep_comm_create: bad world / rank %d / %d", world, rank);
  TUTEL_REQUIRE(id != nullptr && bytes >= sizeof(ncclUniqueId), "tutel_amd_ep_comm_create: need the %zu-byte id of tutel_amd_ep_unique_id", sizeof(ncclUniqueId));
  if (tutel_amd_ep_load_rccl(nullptr) != 0) return -1;
  tutel_amd_ep_comm *c = (tutel_amd_ep_comm *)calloc(1, sizeof(tutel_amd_ep_comm));
  TUTEL_REQUIRE(c != nullptr, "tutel_amd_ep_comm_create: out of memory");
  c->world = world;
  c->rank = rank;
  // every failure below releases what was created so far (tutel_amd_ep_comm_destroy tolerates the zeroed fields)
  auto fail = [&](int rc) {
    (void)tutel_amd_ep_comm_destroy(c);
    return rc;
  };
  if (hipGetDevice(&c->device) != hipSuccess) {
    tutel_set_error("tutel_amd_ep_comm_create: hipGetDevice failed");
    return fail(-1);
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t nr = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (nr != ncclSuccess) {
    c->comm = nullptr;
    tutel_set_error("ncclCommInitRank: RCCL error %d (%s)", (int)nr, g_rccl.GetErrorString(nr));
    return fail((int)nr);
  }
  bool ok = create_side_streams(c);
  for (int i = 0; ok && i < EP_MAX_SPLIT; ++i)
    ok = hipEventCreateWithFlags(&c->recv_ev[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->done_ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    tutel_set_error("tutel_amd_ep_comm_create: cannot create the side stream / event table: %s", hipGetErrorString(hipGetLastError()));
    return fail(-1);
  }
  *out = c;
  return 0;
}

extern "C" int tutel_amd_ep_comm_create_hosted(int world, int rank, tutel_amd_exchange_fn fn, void *user, tutel_amd_ep_comm_t **out) {
  TUTEL_REQUIRE(out != nullptr && fn != nullptr && world >= 1 && rank >= 0 && rank < world, "tutel_amd_ep_comm_create_hosted: bad arguments");
  tutel_amd_ep_comm *c = (tutel_amd_ep_comm *)calloc(1, sizeof(tutel_amd_ep_comm));
  TUTEL_REQUIRE(c != nullptr, "tutel_amd_ep_comm_create_hosted: out of memory");
  c->world = world;
  c->rank = rank;
  c->hosted = fn;
  c->hosted_user = user;
  bool ok = hipGetDevice(&c->device) == hipSuccess && create_side_streams(c);
  for (int i = 0; ok && i < EP_MAX_SPLIT; ++i)
    ok = hipEventCreateWithFlags(&c->recv_ev[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->done_ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    tutel_set_error("tutel_amd_ep_comm_create_hosted: cannot create the side stream / event table");
    (void)tutel_amd_ep_comm_destroy(c);
    return -1;
  }
  *out = c;
  return 0;
}

extern "C" int tutel_amd_ep_comm_destroy(tutel_amd_ep_comm_t *c) {
  if (c == nullptr) return 0;
  if (c->side_stream != nullptr) (void)hipStreamSynchronize(c->side_stream);
  if (c->side_stream_low != nullptr) (void)hipStreamSynchronize(c->side_stream_low);
  if (c->side_stream_mid != nullptr) (void)hipStreamSynchronize(c->side_stream_mid);
  if (c->comm != nullptr && g_rccl.CommDestroy != nullptr) (void)g_rccl.CommDestroy(c->comm);
  if (c->side_stream != nullptr) (void)hipStreamDestroy(c->side_stream);
  if (c->side_stream_low != nullptr) (void)hipStreamDestroy(c->side_stream_low);
  if (c->side_stream_mid != nullptr) (void)hipStreamDestroy(c->side_stream_mid);
  for (int i = 0; i < EP_MAX_SPLIT; ++i) {
    if (c->recv_ev[i] != nullptr) (void)hipEventDestroy(c->recv_ev[i]);
    if (c->done_ev[i] != nullptr) (void)hipEventDestroy(c->done_ev[i]);
  }
  free(c);
  return 0;
}

extern "C" int tutel_amd_ep_comm_info(const tutel_amd_ep_comm_t *c, int *world, int *rank) {
  TUTEL_REQUIRE(c != nullptr, "tutel_amd_ep_comm_info: null communicator");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return 0;
}

// equal-split all-to-all of `bytes_per_peer` bytes per rank pair on `stream` (all_to_all_single semantics,
// communicate.py:181-192): block r of `send` goes to rank r and lands as block (my rank) of its `recv`
static int exchange(tutel_amd_ep_comm *c, const void *send, void *recv, size_t bytes_per_peer, int world, hipStream_t st,
                    int stage = TUTEL_STAGE_OTHER) {
  if (bytes_per_peer == 0) return 0;
  // single rank without a communicator: the exchange is the identity -- nothing to do when the caller aliased the stage
  // buffers (impls/ep_native.py does), a copy otherwise
  if (c == nullptr && send == recv) return 0;
  StageScope scope(stage, st);
  if (c == nullptr) {
    HIP_CHECK(hipMemcpyAsync(recv, send, bytes_per_peer * (size_t)world, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
    return 0;
  }
  if (c->hosted != nullptr) {  // bring-up / test communicator
    const int rc = c->hosted(c->hosted_user, send, recv, bytes_per_peer, world);
    TUTEL_REQUIRE(rc == 0, "tutel_amd_ep_forward: the host exchange callback failed (%d)", rc);
    return 0;
  }
  if ((bytes_per_peer & 1) == 0)
    RCCL_CHECK(g_rccl.AllToAll(send, recv, bytes_per_peer / 2, ncclFloat16, c->comm, st), "ncclAllToAll");
  else
    RCCL_CHECK(g_rccl.AllToAll(send, recv, bytes_per_peer, ncclInt8, c->comm, st), "ncclAllToAll");
  return 0;
}

extern "C" int tutel_amd_ep_all_to_all(tutel_amd_ep_comm_t *c, const void *send, void *recv, size_t bytes_per_peer,
                                       tutel_stream_t stream) {
  TUTEL_REQUIRE(c != nullptr && send != nullptr && recv != nullptr && send != recv, "tutel_amd_ep_all_to_all: need a communicator and two distinct buffers");
  return exchange(c, send, recv, bytes_per_peer, c->world, (hipStream_t)stream);
}

// ---- variable-size exchanges (tutel.net.batch_all_to_all_v / batch_all_gather_v; custom_kernel.cpp:463-518) -----------
extern "C" int tutel_amd_ep_comm_set_hosted_v(tutel_amd_ep_comm_t *c, tutel_amd_exchange_v_fn fn) {
  TUTEL_REQUIRE(c != nullptr && c->hosted != nullptr, "tutel_amd_ep_comm_set_hosted_v: need a hosted communicator");
  c->hosted_v = fn;
  return 0;
}

// one grouped send / recv loop: to rank r `sb[r]` bytes from send + so[r], from rank r `rb[r]` bytes into the running offset
static int exchange_v(tutel_amd_ep_comm *c, const void *send, void *recv, const uint64_t *sb, const uint64_t *so, const uint64_t *rb,
                      hipStream_t st, const char *what) {
  StageScope scope(TUTEL_STAGE_OTHER, st);
  if (c->hosted != nullptr) {
    TUTEL_REQUIRE(c->hosted_v != nullptr, "%s: the hosted communicator has no variable-size callback (tutel_amd_ep_comm_set_hosted_v)", what);
    const int rc = c->hosted_v(c->hosted_user, send, recv, sb, so, rb, c->world);
    TUTEL_REQUIRE(rc == 0, "%s: the host exchange callback failed (%d)", what, rc);
    return 0;
  }
  RCCL_CHECK(g_rccl.GroupStart(), "ncclGroupStart");
  uint64_t ro = 0;
  ncclResult_t bad = ncclSuccess;
  for (int r = 0; r < c->world && bad == ncclSuccess; ++r) {
    if (sb[r]) bad = g_rccl.Send((const char *)send + so[r], (size_t)sb[r], ncclInt8, r, c->comm, st);
    if (rb[r] && bad == ncclSuccess) bad = g_rccl.Recv((char *)recv + ro, (size_t)rb[r], ncclInt8, r, c->comm, st);
    ro += rb[r];
  }
  const ncclResult_t end = g_rccl.GroupEnd();  // always closed, also after a failed send / recv
  RCCL_CHECK(bad, what);
  RCCL_CHECK(end, "ncclGroupEnd");
  return 0;
}

extern "C" int tutel_amd_ep_all_to_all_v(tutel_amd_ep_comm_t *c, const void *send, void *recv, const uint64_t *send_bytes,
                                         const uint64_t *recv_bytes, tutel_stream_t stream) {
  TUTEL_REQUIRE(c != nullptr && send_bytes != nullptr && recv_bytes != nullptr, "tutel_amd_ep_all_to_all_v: need a communicator and both size arrays");
  TUTEL_REQUIRE(c->world <= 4096, "tutel_amd_ep_all_to_all_v: world size %d", c->world);
  uint64_t so[4096], tot_s = 0, tot_r = 0;
  for (int r = 0; r < c->world; ++r) {
    so[r] = tot_s;
    tot_s += send_bytes[r];
    tot_r += recv_bytes[r];
  }
  TUTEL_REQUIRE((send != nullptr || tot_s == 0) && (recv != nullptr || tot_r == 0) && (send != recv || tot_s + tot_r == 0),
                "tutel_amd_ep_all_to_all_v: null or aliased buffers");
  return exchange_v(c, send, recv, send_bytes, so, recv_bytes, (hipStream_t)stream, "tutel_amd_ep_all_to_all_v");
}

extern "C" int tutel_amd_ep_all_gather_v(tutel_amd_ep_comm_t *c, const void *send, void *recv, const uint64_t *recv_bytes,
                                         tutel_stream_t stream) {
  TUTEL_REQUIRE(c != nullptr && recv_bytes != nullptr, "tutel_amd_ep_all_gather_v: need a communicator and the size array");
  TUTEL_REQUIRE(c->world <= 4096, "tutel_amd_ep_all_gather_v: world size %d", c->world);
  uint64_t sb[4096], so[4096], tot_r = 0;
  for (int r = 0; r < c->world; ++r) {
    sb[r] = recv_bytes[c->rank];  // the same bytes to everybody
    so[r] = 0;
    tot_r += recv_bytes[r];
  }
  TUTEL_REQUIRE((send != nullptr || recv_bytes[c->rank] == 0) && (recv != nullptr || tot_r == 0) && (send != recv || tot_r == 0),
                "tutel_amd_ep_all_gather_v: null or aliased buffers");
  return exchange_v(c, send, recv, sb, so, recv_bytes, (hipStream_t)stream, "tutel_amd_ep_all_gather_v");
}

// ---- stage layouts (== tutel_amd/impls/overlap.py::OverlapPlan) ------------------------------------------------
extern "C" int tutel_amd_ep_plan(int E, int W, int capacity, int degree, int allow_sliced, tutel_amd_ep_plan_t *out) {
  TUTEL_REQUIRE(out != nullptr && E >= 1 && W >= 1 && E % W == 0 && degree >= 1 && degree <= EP_MAX_SPLIT && capacity >= 0,
                "tutel_amd_ep_plan: bad sizes E=%d W=%d C=%d degree=%d", E, W, capacity, degree);
  const int E_loc = E / W;
  out->sliced = (allow_sliced && E_loc >= degree && E_loc % degree == 0) ? 1 : 0;
  if (out->sliced) {
    out->experts_per_stage = E_loc / degree;
    out->chunk = capacity;
  } else {
    TUTEL_REQUIRE(capacity % degree == 0, "tutel_amd_ep_plan: capacity %d is not a multiple of a2a_ffn_overlap_degree %d", capacity, degree);
    out->experts_per_stage = E_loc;
    out->chunk = capacity / degree;
  }
  out->rows = out->experts_per_stage * out->chunk;  // bucket rows per (stage, rank) block
  out->gemm_rows = W * out->chunk;                  // GEMM rows per expert and stage
  return 0;
}

// ---- the pipeline ---------------------------------------------------------------------------------------------
extern "C" int tutel_amd_ep_forward(tutel_amd_ep_comm_t *c, const tutel_amd_ep_args_t *a, tutel_stream_t stream) {
  TUTEL_REQUIRE(a != nullptr, "tutel_amd_ep_forward: null arguments");
  const int W = a->world, E = a->num_experts, C = a->capacity, T = a->T, M = a->M, H = a->H, Mo = a->M_out, k = a->k;
  TUTEL_REQUIRE(W >= 1 && E >= 1 && E % W == 0 && T >= 0 && M >= 1 && H >= 1 && Mo >= 1 && k >= 1 && C >= 0,
                "tutel_amd_ep_forward: bad sizes");
  TUTEL_REQUIRE(c == nullptr ? W == 1 : c->world == W, "tutel_amd_ep_forward: communicator world size does not match (%d)", W);
  TUTEL_REQUIRE(a->dtype == TUTEL_BF16 || a->dtype == TUTEL_F16, "tutel_amd_ep_forward: bf16 / fp16 experts only (got dtype %d)", a->dtype);
  TUTEL_REQUIRE((T == 0 || (a->x && a->idx && a->loc && a->y)) && (a->slot_map || C == 0) && a->w1 && a->w2, "tutel_amd_ep_forward: null pointer");
  // a rank without tokens still owes its peers every collective of the pipeline (they block in ncclAllToAll otherwise):
  // only a single rank without a communicator may leave here.  With a communicator the empty rank encodes all-zero
  // buckets, runs its experts on what the others send and skips nothing but its own (empty) decode.
  if (T == 0 && c == nullptr) return 0;
  const int E_loc = E / W, es = 2;
  hipStream_t cur = (hipStream_t)stream;
  const void *enc_gates = a->is_postscore ? nullptr : a->gates;  // fast_dispatch.py:125,131: gates on one side only
  const void *dec_gates = a->is_postscore ? a->gates : nullptr;
  TUTEL_REQUIRE(a->gates != nullptr || T == 0, "tutel_amd_ep_forward: null gates");
  TUTEL_REQUIRE(a->row_counts == nullptr || (W == 1 && c == nullptr && a->degree <= 1), "tutel_amd_ep_forward: row counts (megablocks) need a single rank");
  const int32_t *rcnt = a->row_counts;
  const int ralign = a->row_counts != nullptr && a->row_align >= 1 ? a->row_align : 1;

  if (C == 0) {  // nothing is dispatched (on any rank: the capacity is agreed): every token's output is the zero vector
    if (T > 0) HIP_CHECK(hipMemsetAsync(a->y, 0, (size_t)T * Mo * es, cur), "hipMemsetAsync");
    return 0;
  }

  // single rank, no communicator, pure-copy encode: fc1 gathers its rows from the tokens (no bucket array at all)
  if (c == nullptr && a->degree <= 1 && a->is_postscore && a->fuse_encode) {
    TUTEL_REQUIRE(a->hid && a->send && a->zero_row, "tutel_amd_ep_forward: null workspace");
    int rc;
    {
      Range r("tutel_amd.expert_fc1");
      rc = tutel_amd_expert_gemm_gather(a->x, M, a->slot_map, T, a->zero_row, a->w1, 1, (int64_t)H * M, M, a->b1, H, a->hid,
                                        (int64_t)C * H, H, E_loc, C, H, M, a->dtype, a->act, rcnt, ralign, cur);
      if (rc) return rc;
    }
    {
      Range r("tutel_amd.expert_fc2");
      rc = tutel_amd_expert_gemm(a->hid, (int64_t)C * H, 0, C, H, a->w2, a->w2_kmajor, (int64_t)H * Mo, a->w2_kmajor ? H : Mo, a->b2,
                                 Mo, a->send, (int64_t)C * Mo, 0, C, Mo, E_loc, C, Mo, H, a->dtype, TUTEL_ACT_NONE, rcnt, ralign, cur);
      if (rc) return rc;
    }
    Range r("tutel_amd.fast_decode");
    return tutel_amd_fast_decode(a->send, a->dtype, a->idx, a->loc, dec_gates, a->gate_dtype, T, Mo, k, C, E, 0, 0, 1, a->y, cur);
  }

  TUTEL_REQUIRE(a->enc && a->recv && a->hid && a->send && a->back, "tutel_amd_ep_forward: null workspace");
  TUTEL_REQUIRE(rcnt == nullptr, "tutel_amd_ep_forward: row counts need the fused-encode single-rank route (is_postscore, fuse_encode)");
  const int degree = a->degree < 1 ? 1 : a->degree;
  tutel_amd_ep_plan_t pl;
  if (tutel_amd_ep_plan(E, W, C, degree, a->allow_sliced, &pl) != 0) return -1;
  const int s = pl.experts_per_stage, cc = pl.chunk, rows = pl.rows, R = pl.gemm_rows;
  const int chunk_rows = pl.sliced ? 0 : cc, expert_slice = pl.sliced ? s : 0;
  int rc;

  {
    Range r("tutel_amd.fast_encode");
    rc = tutel_amd_fast_encode(a->x, a->dtype, a->slot_map, enc_gates, a->gate_dtype, T, M, E * C, C, E, chunk_rows, expert_slice, W, a->enc, cur);
    if (rc) return rc;
  }
  const size_t msg_in = (size_t)W * rows * M * es, msg_out = (size_t)W * rows * Mo * es;  // bytes per stage
  const size_t hid_stage = (size_t)s * R * H * es;
  auto stage_gemms = [&](int i, hipStream_t st) -> int {
    // GEMM rows addressed in the raw exchange buffer: expert el of the stage, source rank w, row l -> ((w*s + el)*cc + l)
    const char *recv_i = (const char *)a->recv + (size_t)i * msg_in;
    char *send_i = (char *)a->send + (size_t)i * msg_out;
    char *hid_i = (char *)a->hid + (size_t)i * hid_stage;
    const int e0 = pl.sliced ? i * s : 0;  // first local expert of the stage
    const char *w1 = (const char *)a->w1 + (size_t)e0 * H * M * es, *w2 = (const char *)a->w2 + (size_t)e0 * H * Mo * es;
    const char *b1 = a->b1 ? (const char *)a->b1 + (size_t)e0 * H * es : nullptr, *b2 = a->b2 ? (const char *)a->b2 + (size_t)e0 * Mo * es : nullptr;
    int r1;
    {
      Range r("tutel_amd.expert_fc1");
      tutel_stage_hint(TUTEL_STAGE_FC1);
      r1 = tutel_amd_expert_gemm(recv_i, (int64_t)cc * M, (int64_t)rows * M, cc, M, w1, 1, (int64_t)H * M, M, b1, H, hid_i,
                                 (int64_t)R * H, 0, R, H, s, R, H, M, a->dtype, a->act, nullptr, 1, st);
      tutel_stage_hint(-1);
      if (r1) return r1;
    }
    Range r("tutel_amd.expert_fc2");
    return tutel_amd_expert_gemm(hid_i, (int64_t)R * H, 0, R, H, w2, a->w2_kmajor, (int64_t)H * Mo, a->w2_kmajor ? H : Mo, b2, Mo, send_i,
                                 (int64_t)cc * Mo, (int64_t)rows * Mo, cc, Mo, s, R, Mo, H, a->dtype, TUTEL_ACT_NONE, nullptr, 1, st);
  };

  if (degree == 1 || c == nullptr) {
    // one stream: exchange, GEMMs, exchange per stage, in order (degree 1; or a single rank whose exchange is a copy)
    for (int i = 0; i < degree; ++i) {
      {
        Range r("tutel_amd.all_to_all");
        rc = exchange(c, (const char *)a->enc + (size_t)i * msg_in, (char *)a->recv + (size_t)i * msg_in, (size_t)rows * M * es, W, cur, TUTEL_STAGE_A2A_DISPATCH);
        if (rc) return rc;
      }
      rc = stage_gemms(i, cur);
      if (rc) return rc;
      Range r("tutel_amd.all_to_all");
      rc = exchange(c, (const char *)a->send + (size_t)i * msg_out, (char *)a->back + (size_t)i * msg_out, (size_t)rows * Mo * es, W, cur, TUTEL_STAGE_A2A_COMBINE);
      if (rc) return rc;
    }
  } else {
    // 3-stage pipeline over two streams: stage i+1 is on the links while stage i is in the GEMMs and stage i-1 travels
    // back.  The COLLECTIVES stay on the caller's stream and the GEMMs go to the communicator's side stream: the
    // caller's stream is the origin of a HIP-graph capture, and RCCL can be captured there but not on a stream that
    // joined the capture through an event (segfault inside the library, tools/graph_rccl_probe.py) -- plain kernel
    // launches are fine on either.  Eager and captured execution take this one path.
    hipStream_t kss[2];
    side_streams_for(c, cur, kss);
    {
      Range r("tutel_amd.all_to_all(dispatch)");
      for (int i = 0; i < degree; ++i) {
        rc = exchange(c, (const char *)a->enc + (size_t)i * msg_in, (char *)a->recv + (size_t)i * msg_in, (size_t)rows * M * es, W, cur, TUTEL_STAGE_A2A_DISPATCH);
        if (rc) return rc;
        HIP_CHECK(hipEventRecord(c->recv_ev[i], cur), "hipEventRecord");
      }
    }
    // Stage i's GEMMs go to side stream i % 2: fc1 of stage i + 1 does not depend on fc2 of stage i, and at the expert-parallel
    // shapes a stage launch is a half-chip grid (128 workgroups of the 256 x 256 kernel), so two stages side by side fill the
    // GPU: 4 launches take ~3 launch times instead of 4 (degree 2).  The ENQUEUE order stays stage by stage -- GEMMs of stage i,
    // join, return exchange of stage i -- so that every RCCL call is captured with all forked streams joined back (RCCL cannot
    // be captured next to an open fork, tools/graph_rccl_probe.py); the device runs by dependencies, not by enqueue order:
    // stage i + 1 on the other stream only waits for its own recv event.
    for (int i = 0; i < degree; ++i) {
      hipStream_t ks = kss[i & 1];
      HIP_CHECK(hipStreamWaitEvent(ks, c->recv_ev[i], 0), "hipStreamWaitEvent");  // the side stream forks from the caller's
      tutel_gemm_corun_hint(c->comm != nullptr);  // a real collective runs beside these GEMMs (the hosted test exchange is synchronous)
      rc = stage_gemms(i, ks);
      tutel_gemm_corun_hint(0);
      if (rc) return rc;
      HIP_CHECK(hipEventRecord(c->done_ev[i], ks), "hipEventRecord");
      HIP_CHECK(hipStreamWaitEvent(cur, c->done_ev[i], 0), "hipStreamWaitEvent");  // and is joined back
      Range r("tutel_amd.all_to_all(combine)");
      rc = exchange(c, (const char *)a->send + (size_t)i * msg_out, (char *)a->back + (size_t)i * msg_out, (size_t)rows * Mo * es, W, cur, TUTEL_STAGE_A2A_COMBINE);
      if (rc) return rc;
    }
  }

  Range r("tutel_amd.fast_decode");
  return tutel_amd_fast_decode(a->back, a->dtype, a->idx, a->loc, dec_gates, a->gate_dtype, T, Mo, k, C, E, chunk_rows, expert_slice, W, a->y, cur);
}

// ---- routing + pipeline in one call ------------------------------------------------------------------------------
extern "C" int tutel_amd_moe_forward(tutel_amd_ep_comm_t *c, const tutel_amd_moe_args_t *m, tutel_stream_t stream) {
  TUTEL_REQUIRE(m != nullptr, "tutel_amd_moe_forward: null arguments");
  const tutel_amd_ep_args_t &a = m->ep;
  const int T = a.T, E = a.num_experts, k = a.k;
  TUTEL_REQUIRE((m->logits != nullptr || T == 0) && m->ws != nullptr && m->dispatch_count != nullptr, "tutel_amd_moe_forward: null pointer");
  TUTEL_REQUIRE(a.slot_map && (T == 0 || (a.idx && a.loc && a.gates)), "tutel_amd_moe_forward: null routing buffers");
  if (T == 0 && c == nullptr) return 0;  // (with a communicator an empty rank still takes part in every exchange, see tutel_amd_ep_forward)
  const bool dropless = a.capacity <= 0;
  TUTEL_REQUIRE(!dropless || (c == nullptr && a.world == 1 && m->stats != nullptr && m->capacity_out != nullptr && m->max_capacity >= 1),
                "tutel_amd_moe_forward: dropless routing needs a single rank, stats, capacity_out and max_capacity");
  int32_t *smap = const_cast<int32_t *>(a.slot_map);
  int rc = TUTEL_AMD_ENOTSUP;
  // top-k + locations in one launch (tutel_amd_route) only on request, TUTEL_OPT_ROUTING = 1: measured equal to the two launches
  // (17.8 vs 9.6 + 8.6 us, profiles/r03_routing_fused_and_decode_ab.txt) -- the default stays two launches
  if (T > 0 && m->route_sync != nullptr && tutel_get_option(TUTEL_OPT_ROUTING) == 1 && m->logits_dtype != TUTEL_F64)
    rc = tutel_amd_route(m->logits, m->logits_dtype, T, E, k, m->normalize_gate, const_cast<int32_t *>(a.idx), const_cast<void *>(a.gates),
                         m->ws, m->ws_bytes, const_cast<int32_t *>(a.loc), m->dispatch_count, m->stats, m->l_aux,
                         dropless ? 0 : a.capacity, dropless ? nullptr : smap, m->route_sync, stream);
  if (rc == TUTEL_AMD_ENOTSUP) {
    rc = tutel_amd_gate_topk(m->logits, m->logits_dtype, 1, T, E, k, m->normalize_gate, nullptr, const_cast<int32_t *>(a.idx),
                             const_cast<void *>(a.gates), m->ws, m->ws_bytes, dropless ? nullptr : smap, dropless ? 0 : E * a.capacity, stream);
    if (rc) return rc;
    rc = tutel_amd_compute_location(a.idx, T, E, k, 1, m->ws, m->ws_bytes, const_cast<int32_t *>(a.loc), m->dispatch_count, m->stats,
                                    m->l_aux, m->logits_dtype, dropless ? 0 : a.capacity, dropless ? nullptr : smap, dropless ? 0 : 1, stream);
  }
  if (rc) return rc;
  tutel_amd_ep_args_t e = a;
  e.gate_dtype = m->logits_dtype;
  if (dropless) {
    // the one host synchronisation of the dropless API (fast_dispatch.py:192-193), taken here so that nothing but this
    // function stands between the read-back and the next launch
    // the read-back lands in the caller's own slot (m->capacity_out: any host memory; pinned memory makes the copy
    // asynchronous, pageable memory makes the runtime stage it) -- no process-global state, so callers on different
    // threads / streams / devices cannot read each other's capacity
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipMemcpyAsync(m->capacity_out, m->stats, sizeof(int), hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    HIP_CHECK(hipStreamSynchronize(st), "hipStreamSynchronize");
    int cap = *m->capacity_out;
    if (m->capacity_limit > 0 && cap > m->capacity_limit) cap = m->capacity_limit;
    const int al = m->alignment >= 1 ? m->alignment : 1;
    cap = (cap + al - 1) / al * al;
    *m->capacity_out = cap;
    if (cap > m->max_capacity) return TUTEL_AMD_EAGAIN;
    if (cap == 0) return hipMemsetAsync(a.y, 0, (size_t)T * a.M_out * 2, st) == hipSuccess ? 0 : -1;
    rc = tutel_amd_slot_map(a.idx, a.loc, T, E, k, cap, smap, stream);
    if (rc) return rc;
    e.capacity = cap;
  } else if (m->capacity_out != nullptr) {
    *m->capacity_out = a.capacity;
  }
  return tutel_amd_ep_forward(c, &e, stream);
}
