"""All-to-all / expert-FFN overlap (reference: tutel/impls/overlap.py + the stream/event code of
custom_kernel.cpp:433-654).

The capacity dimension is cut into `degree` chunks; chunk i's dispatch all-to-all, expert FFN and
combine all-to-all form a 3-stage pipeline across two HIP streams: RCCL kernels run on a
dedicated communication stream, the grouped GEMMs on the caller's stream, HIP events hand chunks
over.  While chunk i is in the FFN, chunk i+1 is on the xGMI links and chunk i-1 is travelling
back.  Stream discipline: the generic routine (a2a_ffn_overlap_forward: custom experts, training fallback)
registers every tensor that crosses streams with the caching allocator (record_stream), as the reference
does with CUDACachingAllocator::recordStream (custom_kernel.cpp:536-550,609-618); the copy-free routine
(a2a_ffn_overlap_fused) allocates its stage buffers once on the caller's stream and keeps them referenced
until the communication stream has been joined back -- no record_stream bookkeeping.  Result is identical to
the non-overlapped path -- exactly what the reference asserts in tests/test_tutel.py:161-176.

Production path since round 2: the same pipeline behind ONE native call (impls/ep_native.py ->
tutel_amd_ep_forward, csrc/ep.hip) on the library's own RCCL communicator; the routines here remain as
the torch.distributed fallback (gloo rendezvous, custom expert modules, training)."""
import torch
import torch.distributed as dist

from . import communicate as C

MAX_NUM_SPLIT = 32
_comm_streams = {}


def _comm_stream(device):
    """the communication stream: HIGH priority, so that it never shares a hardware queue with a normal-priority caller stream
    -- HIP serialises streams that share a queue, and every fourth normal-priority stream lands on the default stream's
    (profiles/r03_stream_queues.txt).  A caller that is itself on a high-priority stream gets a low-priority one."""
    cur_prio = torch.cuda.current_stream(device).priority
    prio = -1 if cur_prio >= 0 else 1
    key = (device.type, device.index, prio)
    if key not in _comm_streams:
        try:
            _comm_streams[key] = torch.cuda.Stream(device=device, priority=prio)
        except (RuntimeError, ValueError):   # a build that knows two levels only
            _comm_streams[key] = torch.cuda.Stream(device=device, priority=0 if prio > 0 else prio)
    return _comm_streams[key]


_event_pool = {}


def _events(device, n):
    """n reusable events for one forward on `device` (re-recording an event is legal; every wait of the
    previous forward has been enqueued by then and captured the earlier record)."""
    key = (device.type, device.index)
    pool = _event_pool.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Event())
    return pool


def a2a_ffn_overlap_forward(input, expert_fn, a2a_ffn_overlap_degree, use_2dh, group, needs_autograd=None):
    """input [E, C, M] -> [E, C, M_out]; expert_fn maps [E_loc, W*c, M] -> [E_loc, W*c, M_out].

    needs_autograd: the caller's statement that something in the chain (the buckets OR the expert
    parameters behind expert_fn) can require grad.  The stream pipeline below exchanges through raw
    collectives into fresh buffers -- no autograd graph -- so it is taken only when nothing can; the
    default (None) is the safe reading `torch.is_grad_enabled()`.  (The reference wraps every step of
    its overlap path in autograd Functions, communicate.py:288-397.)"""
    degree = a2a_ffn_overlap_degree
    assert degree <= MAX_NUM_SPLIT, "Excepting a2a_ffn_overlap_degree (%d) <= AllToAllStatus.max_num_split (%d)." % (degree, MAX_NUM_SPLIT)
    assert input.shape[1] % degree == 0, "Excepting input.shape[%d] (%d) be multiple of a2a_ffn_overlap_degree (%d)." % (1, input.shape[1], degree)
    W = C.get_world_size(group)
    if W == 1:
        return expert_fn(input)
    if needs_autograd is None:
        needs_autograd = torch.is_grad_enabled()
    if (torch.is_grad_enabled() and (needs_autograd or input.requires_grad)) or not input.is_cuda:
        # training / CPU: same chunking, executed in order (autograd-safe); bitwise the same result
        outs = []
        for x in input.chunk(degree, dim=1):
            y = C.all_to_all(x.contiguous(), 1, 0, use_2dh=use_2dh, group=group)
            outs.append(C.all_to_all(expert_fn(y), 0, 1, use_2dh=use_2dh, group=group))
        return torch.cat(outs, dim=1)

    cur = torch.cuda.current_stream()
    comm = _comm_stream(input.device)
    chunks = [x.contiguous() for x in input.chunk(degree, dim=1)]  # [E, c, M] each
    ready = torch.cuda.Event()
    ready.record(cur)

    recv, recv_ev = [None] * degree, [None] * degree
    with torch.cuda.stream(comm):
        comm.wait_event(ready)
        for i, x in enumerate(chunks):
            x.record_stream(comm)
            buf = torch.empty_like(x)
            C.exchange_equal_split(buf, x, group)
            buf.record_stream(cur)
            recv[i], recv_ev[i] = buf, torch.cuda.Event()
            recv_ev[i].record(comm)

    outs = [None] * degree
    for i in range(degree):
        cur.wait_event(recv_ev[i])
        y = expert_fn(C.pre_expert_permute(recv[i], group=group))
        y = C.post_expert_permute(y, group=group).contiguous()
        done = torch.cuda.Event()
        done.record(cur)
        with torch.cuda.stream(comm):
            comm.wait_event(done)
            y.record_stream(comm)
            back = torch.empty_like(y)
            C.exchange_equal_split(back, y, group)
            back.record_stream(cur)
            outs[i] = back
    fin = torch.cuda.Event()
    fin.record(comm)
    cur.wait_event(fin)
    return torch.cat(outs, dim=1)


# ---------------------------------------------------------------------------------------------
# copy-free variant for the HIP expert path
# ---------------------------------------------------------------------------------------------
_FORCE_RCCL = False  # test hook: issue the collective even in a 1-rank group


def _exchange(dst, src, group):
    """equal-split all-to-all of dim-0 blocks (a plain copy when the group has one rank)."""
    if C.get_world_size(group) > 1 or (_FORCE_RCCL and dist.is_initialized()):
        C.exchange_equal_split(dst, src, group)
    else:
        dst.copy_(src)


class OverlapPlan:
    """Buffer layouts of the copy-free overlapped pipeline (pure index arithmetic; shared by
    a2a_ffn_overlap_fused and by tests/test_host_logic_cpu.py, which replays it for W > 1 on CPU).

    sliced:  stages = groups of s = E_loc/degree local experts; buckets laid out [degree, W, s, C]
    chunked: stages = capacity chunks of c = C/degree rows;     buckets laid out [degree, E, c]
    In both, stage i's message is [W, s, c] rows (dim 0 = destination rank), the GEMM addresses the
    received [W(src), s, c] rows as `s` experts x (W*c) source-rank-major rows, and decode finds
    bucket (e, l) through `decode_kwargs`."""

    def __init__(self, E, W, Cap, degree, allow_sliced=True):
        E_loc = E // W
        self.E, self.W, self.Cap, self.degree = E, W, Cap, degree
        self.sliced = bool(allow_sliced and E_loc >= degree and E_loc % degree == 0)
        if self.sliced:
            self.s, self.c = E_loc // degree, Cap
            self._view, self._perm = (W, degree, self.s, Cap), (1, 0, 2, 3)
            self.decode_kwargs = dict(num_experts=E, expert_slice=self.s, ep_world=W)
        else:
            self.s, self.c = E_loc, Cap // degree
            self._view, self._perm = (E, degree, self.c), (1, 0, 2)
            self.decode_kwargs = dict(num_experts=E, chunk_rows=self.c)
        self.rows = self.s * self.c   # bucket rows per (stage, rank) block
        self.R = W * self.c           # GEMM rows per expert and stage

    def permute_slots(self, per_slot):
        """[E*Cap] values in plain bucket order (e*Cap + l) -> the pipeline's bucket order."""
        return per_slot.view(*self._view).permute(*self._perm).contiguous().view(-1)

    def row_layout(self, ld):
        """(stride_e, stride_w, rows_per_w, ld) of a stage buffer with rows of `ld` elements."""
        return (self.c * ld, self.rows * ld, self.c, ld)

    def expert_range(self, i):
        return (i * self.s, (i + 1) * self.s) if self.sliced else None


def a2a_ffn_overlap_fused(layer, x, crit, degree, is_postscore):
    """Overlapped encode -> all-to-all -> expert FFN -> all-to-all -> decode with NO layout copies.

    reference path (overlap.py:8-67 + communicate.py:606-622) per chunk: split view -> contiguous
    copy -> a2a -> permute+contiguous -> FFN -> permute+contiguous -> a2a -> torch.cat, with the
    CAPACITY dimension cut into `degree` chunks.  Two pipelines here, both without copies:

    expert-sliced (E_loc % degree == 0): the pipeline runs over groups of s = E_loc/degree local
      experts.  fast_encode writes the buckets [degree, W, s, C, M] (only its slot map is permuted),
      so stage i is the contiguous message [W, s, C, M]; its GEMMs see ALL W*C rows of s experts:
      every expert's weights are streamed from HBM once per forward (capacity chunks stream all of
      them `degree` times -- 2x the dominant HBM traffic at degree 2) and the row count per launch
      stays at W*C (the 256 x 256-tile kernel's regime).  The return all-to-all lands in slice i of
      one [degree, W, s, C, M_out] buffer that fast_decode addresses through its expert_slice mode.
    capacity-chunked (otherwise): buckets CHUNK-MAJOR [degree, E, c, M]; the GEMMs read the raw
      received buffer [W, E_loc, c, M] through their row addressing (rows_per_w = c); the return
      all-to-all lands in slice i of [degree, E, c, M_out], decoded chunk-major.

    Stage i+1 is on the xGMI links (communication stream) while stage i is in the GEMMs (caller's
    stream); events hand the buffers over and every tensor that crosses streams is registered
    with the caching allocator.  x [T, M] -> [T, M_out]."""
    from .. import ops
    group, experts = layer.group, layer.experts
    W, E_loc = layer.world_size, layer.num_local_experts
    E, Cap = crit[0], crit[4]
    T, M = x.shape
    Mo = experts.output_dim
    dev = x.device
    plan = OverlapPlan(E, W, Cap, degree, allow_sliced=getattr(layer, "megablocks_size", 0) == 0)
    rows = plan.rows
    # Host cost matters here: with ~20 enqueues per forward the eager path is host-bound before it is
    # GPU-bound (measured 0.46 ms/step of Python at degree 2).  So: the bucket order is computed inside the
    # encode kernel (no permuted slot-map tensor), every buffer is allocated up front on the caller's stream
    # and kept referenced until the communication stream has been joined back (no record_stream
    # bookkeeping: the allocator only ever sees these blocks freed on a stream that already waited for all
    # their readers), events are reused, and streams are switched with the raw setter.
    enc = ops.fast_encode(x, crit.slot_map, None if is_postscore else crit.gates2d, E * Cap, capacity=Cap,
                          **plan.decode_kwargs).view(degree, W * rows, M)
    recv = torch.empty([degree, W * rows, M], dtype=x.dtype, device=dev)      # stage i: [W(src), s, c, M]
    send = torch.empty([degree, W * rows, Mo], dtype=x.dtype, device=dev)     # stage i: [W(dst), s, c, M_out]
    out_all = torch.empty([degree, W * rows, Mo], dtype=x.dtype, device=dev)

    cur = torch.cuda.current_stream()
    comm = _comm_stream(dev)
    evs = _events(dev, 2 * degree + 2)
    ready, fin = evs[0], evs[1]
    ready.record(cur)

    torch.cuda.set_stream(comm)
    try:
        comm.wait_event(ready)
        for i in range(degree):
            _exchange(recv[i], enc[i], group)
            evs[2 + i].record(comm)
    finally:
        torch.cuda.set_stream(cur)

    a_layout, d_layout = plan.row_layout(M), plan.row_layout(Mo)
    for i in range(degree):
        cur.wait_event(evs[2 + i])
        experts.forward_fused(recv[i], layer, a_layout=a_layout, R=plan.R, out=send[i], d_layout=d_layout,
                              expert_range=plan.expert_range(i))
        done = evs[2 + degree + i]
        done.record(cur)
        torch.cuda.set_stream(comm)
        try:
            comm.wait_event(done)
            _exchange(out_all[i], send[i], group)
        finally:
            torch.cuda.set_stream(cur)
    fin.record(comm)
    cur.wait_event(fin)

    layer.protected_shape = torch.Size([E_loc, W * Cap, Mo])
    y = ops.fast_decode(out_all.view(E * Cap, Mo), crit.idx2d, crit.loc2d, crit.gates2d if is_postscore else None,
                        Cap, **plan.decode_kwargs)
    del enc, recv, send  # released only now: `cur` has waited for every read the communication stream made
    return y
