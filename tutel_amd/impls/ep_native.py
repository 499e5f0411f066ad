"""The expert-parallel pipeline behind one native call (tutel_amd_ep_forward, csrc/ep.hip).

    encode -> all-to-all -> expert FFN -> all-to-all -> decode

One C call enqueues every kernel of the forward after routing and, with more than one rank, the
RCCL all-to-alls on the library's own communicator and communication stream (event table inside the
library) -- the structure of the reference's native overlap layer (custom_kernel.cpp:341-365,
433-461, 520-654) instead of ~20 Python-driven enqueues per forward.

This module owns the host side of that call:
  * one communicator per process group: rank 0 of the group draws the RCCL unique id, the bytes are
    broadcast with torch.distributed, every rank creates its communicator, then all ranks run a
    tagged all-to-all and agree (all-reduce) that it delivered the right blocks.  Any failure on any
    rank makes every rank fall back to the torch.distributed path (impls/overlap.py), loudly;
  * the workspace of the pipeline, cached per (layer, shape, stream): buffers live as long as the
    layer, calls on one stream are ordered, so nothing is allocated per forward except the output.
"""
import ctypes
import logging
import os

import torch
import torch.distributed as dist

from .. import _lib, ops

ENABLED = int(os.environ.get("TUTEL_AMD_NATIVE_EP", "1")) != 0
FAST_PATH = int(os.environ.get("TUTEL_AMD_FAST_PATH", "1")) != 0  # routing + pipeline in one call (tutel_amd_moe_forward)
_FUSED_LOCATION = int(os.environ.get("TUTEL_AMD_FUSED_LOCATION", "1")) != 0  # hand the one-call path its fused-location scratch (single rank)
HOSTED = int(os.environ.get("TUTEL_AMD_NATIVE_HOSTED", "0")) != 0  # bring-up / tests: native pipeline over a gloo group, exchange staged by the host
# How the bucket rows travel between the ranks of a node:
#   "auto"  RCCL's all-to-all on the library communicator (north_star's exchange: all_to_all_single on separate streams) for ranks on
#           different devices -- since round 6: the peer-store transport below has only ever run between rank
#           processes that SHARE one GPU, and the first multi-GPU run should not have to debug two transports at once.  Set
#           TUTEL_AMD_EP_IPC_VERIFIED=1 once tests/test_multi_gpu_rccl.py has passed on the hardware in question to make "auto" prefer
#           the peer stores again (attach + payload-sized self-check, RCCL when that fails)
#   "ipc"   peer stores over xGMI (IPC transport, csrc/ep.hip) when every rank can map every other rank's segment and the tagged
#           self-check passes on all of them; also for process groups that are not on the "nccl" backend (ranks that share one GPU: the tests)
#   "rccl"  never attach the IPC transport
TRANSPORT = os.environ.get("TUTEL_AMD_EP_TRANSPORT", "auto").lower()
IPC_VERIFIED = os.environ.get("TUTEL_AMD_EP_IPC_VERIFIED", "0") == "1"
# how long a wait kernel spins for a peer's flag before it gives up and the NEXT call reports which peer never arrived: the order of a
# collective watchdog (NCCL's default is 10 minutes; 2 minutes here) -- ranks may legitimately be seconds to minutes apart (data loading, a first-call
# code-object load with eight processes on one box)
IPC_TIMEOUT_MS = int(os.environ.get("TUTEL_AMD_EP_TIMEOUT_MS", "120000"))
_FORCE_COMM = False  # test hook: run a single rank through a real 1-rank RCCL communicator (staged pipeline, both streams)
_comms = {}      # (ranks of the group, device) -> EpComm | False (creation failed: do not retry)
_groups = {}
_zero_rows = {}


class Segment:
    """device memory of this rank that every peer has mapped (tutel_amd_ep_segment_*); `ptr` = local base address"""

    def __init__(self, handle, nbytes):
        self.handle, self.nbytes = handle, nbytes
        self.ptr = int(_lib.lib().tutel_amd_ep_segment_ptr(handle, -1) or 0)


class EpComm:
    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, world, rank
        self.ipc = False       # IPC transport attached (peer stores, no collective on the forward path)
        self.generic = True    # has an exchange for arbitrary buffers (RCCL or the host callback); False: IPC transport only
        self.segments = {}     # rank-invariant key -> Segment; kept for the communicator's lifetime (peers hold mappings)
        self.group = self.device = None

    def segment(self, key, nbytes):
        """the peer-mapped segment of `key`, at least nbytes large.  COLLECTIVE on a miss: the key and the size must be the same
        on every rank (they are functions of the model / expert / capacity sizes, never of a rank's own token count)"""
        sg = self.segments.get(key)
        if sg is None:
            sg = self.segments[key] = _open_segment(self, nbytes, False)
        return sg

    def all_to_all(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == inp.numel() and inp.numel() % self.world == 0
        per_peer = inp.numel() * inp.element_size() // self.world
        _lib.check(_lib.lib().tutel_amd_ep_all_to_all(self.handle, inp.data_ptr(), out.data_ptr(), per_peer, ops._stream()),
                   "tutel_amd_ep_all_to_all")

    def _scoped(self, *tensors):
        """hosted communicators resolve device pointers through registered tensors: register for one call"""
        reg = getattr(self, "register", None)
        if reg is None:
            return lambda: None
        n = len(tensors)
        for t in tensors:
            reg(t)
        return lambda: self.unregister(n)

    def all_to_all_v(self, inp, send_counts, recv_counts):
        """flat `inp` split by send_counts (elements per destination rank) -> flat tensor of sum(recv_counts) elements
        (tutel_amd_ep_all_to_all_v: one grouped ncclSend / ncclRecv loop on the library's communicator)"""
        assert inp.is_contiguous() and inp.dim() == 1 and len(send_counts) == self.world == len(recv_counts)
        es = inp.element_size()
        out = torch.empty([int(sum(recv_counts))], dtype=inp.dtype, device=inp.device)
        arr = ctypes.c_uint64 * self.world
        sb, rb = arr(*[int(c) * es for c in send_counts]), arr(*[int(c) * es for c in recv_counts])
        done = self._scoped(inp, out)
        try:
            _lib.check(_lib.lib().tutel_amd_ep_all_to_all_v(self.handle, inp.data_ptr() or None, out.data_ptr() or None, sb, rb, ops._stream()),
                       "tutel_amd_ep_all_to_all_v")
        finally:
            done()
        return out

    def all_gather_v(self, inp, counts):
        """flat `inp` (counts[rank] elements) -> the concatenation of every rank's tensor, in rank order"""
        assert inp.is_contiguous() and inp.dim() == 1 and len(counts) == self.world and int(counts[self.rank]) == inp.numel()
        es = inp.element_size()
        out = torch.empty([int(sum(counts))], dtype=inp.dtype, device=inp.device)
        rb = (ctypes.c_uint64 * self.world)(*[int(c) * es for c in counts])
        done = self._scoped(inp, out)
        try:
            _lib.check(_lib.lib().tutel_amd_ep_all_gather_v(self.handle, inp.data_ptr() or None, out.data_ptr() or None, rb, ops._stream()),
                       "tutel_amd_ep_all_gather_v")
        finally:
            done()
        return out


def _rccl_hint():
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()


def _agree(v, group, device):
    """min over the ranks of `group` (all ranks take the same branch afterwards)"""
    on_dev = dist.get_backend(group) == "nccl"
    f = torch.tensor([int(v)], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
    return int(f)


def _all_gather_bytes(raw, group, device):
    """every rank's `raw` (bytes, same length) in rank order"""
    on_dev = dist.get_backend(group) == "nccl"
    t = torch.tensor(list(raw), dtype=torch.uint8, device=device if on_dev else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return b"".join(bytes(o.cpu().tolist()) for o in out)


SEGMENT_ATTEMPTS = 3   # allocate / export / map tries per segment (collective: every rank retries when any rank failed)
_SEGMENT_FAULTS = set()   # tests only: (rank, attempt) pairs whose allocation is reported as failed (tests/test_ep_ipc_one_gpu.py)


def _open_segment(comm, nbytes, flag_memory):
    """allocate + exchange handles + map every peer's allocation.  Collective; raises on every rank if any rank failed.
    Several processes exporting and importing allocations of ONE device at the same moment have shown a transient "invalid argument"
    from the runtime (once in the round-6 runs of the three-rank self-check test, not reproducible on the next box): a failed attempt is
    undone on every rank and repeated, SEGMENT_ATTEMPTS times in all, before the communicator gives up on the transport."""
    import time
    L = _lib.lib()
    hb = _lib.IPC_HANDLE_BYTES
    for attempt in range(SEGMENT_ATTEMPTS):
        handle, raw, ok = ctypes.c_void_p(), (ctypes.c_ubyte * hb)(), 1
        last = attempt + 1 == SEGMENT_ATTEMPTS
        say = logging.warning if last else logging.info
        with torch.cuda.device(comm.device):
            rc = L.tutel_amd_ep_segment_alloc(int(nbytes), int(bool(flag_memory)), ctypes.byref(handle), raw, hb)
            if rc == 0 and (comm.rank, attempt) in _SEGMENT_FAULTS:
                L.tutel_amd_ep_segment_free(handle)
                handle, raw, rc = ctypes.c_void_p(), (ctypes.c_ubyte * hb)(), -1
            if rc != 0:
                say("tutel_amd: rank %d cannot allocate a %d-byte exchange segment (%s), attempt %d of %d", comm.rank, nbytes,
                    L.tutel_amd_last_error().decode(), attempt + 1, SEGMENT_ATTEMPTS)
                ok = 0
            handles = _all_gather_bytes(bytes(raw), comm.group, comm.device)
            all_exported = _agree(ok, comm.group, comm.device)   # (a rank without a handle contributed zeros: nobody maps those)
            if ok and all_exported and L.tutel_amd_ep_segment_open(handle, comm.world, comm.rank, handles, hb) != 0:
                say("tutel_amd: rank %d cannot map its peers' exchange segments (%s), attempt %d of %d", comm.rank,
                    L.tutel_amd_last_error().decode(), attempt + 1, SEGMENT_ATTEMPTS)
                ok = 0
        if _agree(ok and all_exported, comm.group, comm.device):
            return Segment(handle, int(nbytes))
        if handle:
            L.tutel_amd_ep_segment_free(handle)
        if not last:
            time.sleep(0.2 * (attempt + 1))
    raise _lib.TutelAmdError("tutel_amd: the exchange segment could not be opened on every rank")


def _node_identity():
    """64 bytes that are equal for processes of one node and (practically) nowhere else: the kernel's boot id, else the hostname"""
    import hashlib
    import socket
    try:
        ident = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        ident = ""
    return hashlib.sha512((ident + "|" + socket.gethostname()).encode()).digest()


# the payload-sized self-check: per-peer block = SELFCHECK_MB / world MiB (clamped to [1, 32] MiB) -- 64 MiB in all, twice the 32 MiB of
# L2 a producer's dirty lines can hide in; SELFCHECK_PASSES passes back to back, once consuming on the caller's stream and once on a
# side stream of the communicator (the two stream layouts of the pipeline: degree 1 and degree > 1)
SELFCHECK_MB = int(os.environ.get("TUTEL_AMD_EP_SELFCHECK_MB", "64"))
SELFCHECK_PASSES = int(os.environ.get("TUTEL_AMD_EP_SELFCHECK_PASSES", "4"))


def ipc_selfcheck(comm, seg, bytes_per_peer, passes=4, flavour=0, side_stream=False):
    """tutel_amd_ep_ipc_selfcheck on the current stream; returns (vectors that differed, first offender) after synchronising.
    Collective over the communicator's ranks."""
    L = _lib.lib()
    out = torch.zeros([2], dtype=torch.int64, device=comm.device)
    with torch.cuda.device(comm.device):
        _lib.check(L.tutel_amd_ep_ipc_selfcheck(comm.handle, seg.handle, int(bytes_per_peer), int(passes), int(flavour), int(bool(side_stream)),
                                                out.data_ptr(), ops._stream()), "tutel_amd_ep_ipc_selfcheck")
        torch.cuda.synchronize(comm.device)
        _lib.check(L.tutel_amd_ep_ipc_status(comm.handle), "tutel_amd_ep_ipc_status")
    bad, first = (int(v) for v in out.cpu())
    return bad, first


def _attach_ipc(comm, group, device):
    """give `comm` the IPC transport (collective).  True when every rank attached it and the payload-sized self-check (real
    kernels, real flags, epoch canaries, no host synchronisation between its passes) read back exactly what the peers stored, on
    every rank; False (on every rank) otherwise -- the communicator then keeps exchanging the way it did.  Every early exit frees
    what it opened and every decision is an agreement, so the ranks leave together."""
    L = _lib.lib()
    comm.group, comm.device = group, device
    if comm.world > 16:
        return False
    # one node only: hipIpcOpenMemHandle of a handle from another host must not even be attempted
    ids = _all_gather_bytes(_node_identity(), group, device)
    if len(set(ids[i:i + 64] for i in range(0, len(ids), 64))) != 1:
        if comm.rank == 0:
            logging.info("tutel_amd: the ranks of this group are not on one node: no IPC transport")
        return False
    try:
        flags = _open_segment(comm, int(L.tutel_amd_ep_flag_bytes()), True)
    except _lib.TutelAmdError:
        return False
    with torch.cuda.device(device):
        ok = int(L.tutel_amd_ep_comm_attach_ipc(comm.handle, flags.handle, IPC_TIMEOUT_MS) == 0)
    if not _agree(ok, group, device):
        # (a communicator that did attach keeps its pointer to the flag segment: it is never used again -- comm.ipc stays False --
        # but the segment must outlive it, so it is parked on the communicator and freed with it)
        comm._flags = flags
        return False
    comm._flags = flags
    W = comm.world
    per_peer = max(1 << 20, min(32 << 20, (SELFCHECK_MB << 20) // W)) // 256 * 256
    try:
        seg = _open_segment(comm, W * per_peer, False)
    except _lib.TutelAmdError:
        return False
    # the ranks are within milliseconds of each other here, so the waits of the check itself are bounded by 30 s, not by the
    # watchdog-scale default
    L.tutel_amd_ep_ipc_set_timeout(comm.handle, min(30000, IPC_TIMEOUT_MS))
    good, report = 1, []
    for side in (False, True):
        try:
            bad, first = ipc_selfcheck(comm, seg, per_peer, SELFCHECK_PASSES, 0, side)
            report.append(bad)
            if bad:
                raise _lib.TutelAmdError(f"{bad} of {SELFCHECK_PASSES * W * per_peer // 16} 16-byte vectors differed from what the peers stored "
                                         f"(first: from rank {first >> 40}, vector {first & ((1 << 40) - 1)}; consuming on "
                                         f"{'a side stream' if side else 'the calling stream'})")
        except Exception as ex:  # noqa: BLE001
            logging.warning("tutel_amd: the IPC transport failed its self-check on rank %d (%s)", comm.rank, ex)
            good = 0
        if not _agree(good, group, device):
            if comm.rank == 0:
                logging.warning("tutel_amd: IPC transport unavailable; the exchange stays on the communicator's all-to-all")
            L.tutel_amd_ep_segment_free(seg.handle)
            return False
    L.tutel_amd_ep_segment_free(seg.handle)   # (after the agreement: no rank touches it any more)
    L.tutel_amd_ep_ipc_set_timeout(comm.handle, IPC_TIMEOUT_MS)
    comm.selfcheck = {"bytes_per_peer": per_peer, "passes": SELFCHECK_PASSES, "stream_layouts": 2, "mismatches": report}
    comm.ipc = True
    return True


def _create_ipc_only(group, device):
    """communicator without RCCL and without a host callback: only the IPC transport can carry its exchange"""
    L = _lib.lib()
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    handle, ok = ctypes.c_void_p(), 1
    with torch.cuda.device(device):
        if L.tutel_amd_ep_comm_create_ipc(W, rank, ctypes.byref(handle)) != 0:
            logging.warning("tutel_amd: %s", L.tutel_amd_last_error().decode())
            ok = 0
    if not _agree(ok, group, device):
        if handle:
            L.tutel_amd_ep_comm_destroy(handle)
        return None
    c = EpComm(handle, W, rank)
    c.generic = False
    return c


def _create(group, device):
    """Collective over `group`: every rank must call it.  Returns EpComm or None (all ranks agree)."""
    L = _lib.lib()
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    ok, handle = 1, ctypes.c_void_p()
    idbuf = torch.zeros([_lib.EP_ID_BYTES], dtype=torch.uint8)
    try:
        if L.tutel_amd_ep_load_rccl(_rccl_hint()) != 0:
            raise _lib.TutelAmdError(L.tutel_amd_last_error().decode())
        if rank == 0:
            raw = (ctypes.c_ubyte * _lib.EP_ID_BYTES)()
            _lib.check(L.tutel_amd_ep_unique_id(raw, _lib.EP_ID_BYTES), "tutel_amd_ep_unique_id")
            idbuf = torch.tensor(list(raw), dtype=torch.uint8)
    except Exception as ex:  # keep going: the broadcast below is collective
        logging.warning("tutel_amd: native expert-parallel path unavailable on rank %d (%s)", rank, ex)
        ok = 0
    # the id travels through the existing process group (the reference broadcasts it the same way, communicate.py:150-163)
    on_dev = dist.get_backend(group) == "nccl"
    t = idbuf.to(device) if on_dev else idbuf
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    idbytes = bytes(t.cpu().tolist())

    def agree(v):  # all ranks continue only if every rank is fine (a rank that skipped a collective would hang the others)
        f = torch.tensor([v], dtype=torch.int32, device=device if on_dev else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        return int(f)
    comm = None
    ok = agree(ok)     # every rank resolved RCCL and rank 0 drew an id: only then is ncclCommInitRank (collective) entered
    if ok:
        try:
            with torch.cuda.device(device):
                _lib.check(L.tutel_amd_ep_comm_create(idbytes, len(idbytes), W, rank, ctypes.byref(handle)), "tutel_amd_ep_comm_create")
            comm = EpComm(handle, W, rank)
        except Exception as ex:
            logging.warning("tutel_amd: native expert-parallel communicator could not be created on rank %d (%s)", rank, ex)
            ok = 0

    ok = agree(ok)
    if ok:
        try:
            # self-check: block p of my send buffer carries (my rank, p); after the exchange block r must carry (r, my rank)
            n = 1024
            send = (torch.arange(W, device=device, dtype=torch.int32).view(W, 1) + 1000 * rank).repeat(1, n).contiguous()
            recv = torch.full_like(send, -1)
            comm.all_to_all(recv, send)
            torch.cuda.synchronize(device)
            want = (torch.arange(W, device=device, dtype=torch.int32).view(W, 1) * 1000 + rank).repeat(1, n)
            if not torch.equal(recv, want):
                raise _lib.TutelAmdError("tagged all-to-all returned wrong blocks")
        except Exception as ex:
            logging.warning("tutel_amd: native expert-parallel communicator failed its self-check on rank %d (%s)", rank, ex)
            ok = 0
        ok = agree(ok)
    if not ok:
        if handle:
            L.tutel_amd_ep_comm_destroy(handle)
        if rank == 0:
            logging.warning("tutel_amd: falling back to the torch.distributed all-to-all path (impls/overlap.py)")
        return None
    return comm


def _create_hosted(group, device):
    """communicator whose exchange is done by the host over `group` (gloo): the native pipeline with ranks that share a GPU"""
    from . import communicate as C
    L = _lib.lib()
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    bufs = []   # (base, nbytes, uint8 view) of the tensors exchanges may address -- registered by the workspaces

    def find(ptr, nbytes):
        for base, size, view in bufs:
            if base <= ptr and ptr + nbytes <= base + size:
                return view[ptr - base:ptr - base + nbytes]
        raise _lib.TutelAmdError("hosted exchange: pointer outside the registered workspace")

    def cb(_user, send, recv, per_peer, world):
        try:
            n = int(per_peer) * int(world)
            C.exchange_equal_split(find(int(recv), n), find(int(send), n), group)   # device -> host -> gloo -> device, on the current stream
            return 0
        except Exception as ex:  # noqa: BLE001
            logging.error("tutel_amd: hosted exchange failed: %s", ex)
            return 1
    def cb_v(_user, send, recv, sb, so, rb, world):
        try:
            sb, so, rb = ([int(a[i]) for i in range(world)] for a in (sb, so, rb))
            span = max([o + b for o, b in zip(so, sb)] + [0])
            src = find(int(send or 0), span) if span else None
            host = torch.cat([src[o:o + b] for o, b in zip(so, sb)]).cpu() if span else torch.empty([0], dtype=torch.uint8)
            out = torch.empty([sum(rb)], dtype=torch.uint8)
            dist.all_to_all_single(out, host, output_split_sizes=rb, input_split_sizes=sb, group=group)
            if sum(rb):
                find(int(recv), sum(rb)).copy_(out)
            return 0
        except Exception as ex:  # noqa: BLE001
            logging.error("tutel_amd: hosted variable-size exchange failed: %s", ex)
            return 1
    fn, fn_v = _lib.EXCHANGE_FN(cb), _lib.EXCHANGE_V_FN(cb_v)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        _lib.check(L.tutel_amd_ep_comm_create_hosted(W, rank, ctypes.cast(fn, ctypes.c_void_p), None, ctypes.byref(handle)),
                   "tutel_amd_ep_comm_create_hosted")
        _lib.check(L.tutel_amd_ep_comm_set_hosted_v(handle, ctypes.cast(fn_v, ctypes.c_void_p)), "tutel_amd_ep_comm_set_hosted_v")
    comm = EpComm(handle, W, rank)
    comm._keep = (fn, fn_v)
    comm.register = lambda t: bufs.append((t.data_ptr(), t.numel() * t.element_size(), t.view(-1).view(torch.uint8)))
    comm.unregister = lambda n: bufs.__delitem__(slice(len(bufs) - n, len(bufs)))
    return comm


def group_ok(group):
    """can the native pipeline exchange over `group`?  RCCL needs the "nccl" backend; the hosted test exchange and the IPC
    transport (TRANSPORT == "ipc") also run over a gloo rendezvous"""
    return dist.is_initialized() and (dist.get_backend(group) == "nccl" or HOSTED or TRANSPORT == "ipc")


def _group_key(group):
    """cache key of a process group: its member ranks (id(group) can be reused after garbage collection)"""
    if group is None:
        return "world"
    try:
        return tuple(dist.get_process_group_ranks(group))
    except Exception:  # noqa: BLE001
        return id(group)


def communicator(group, device):
    """EpComm of `group` (created on first use; collective), or None when the native path is unavailable."""
    key = (_group_key(group), str(device))
    ent = _comms.get(key)
    if ent is None:
        if dist.get_backend(group) != "nccl":
            c = _create_hosted(group, device) if HOSTED else (_create_ipc_only(group, device) if TRANSPORT == "ipc" else None)
            if c is not None and TRANSPORT == "ipc" and not _attach_ipc(c, group, device) and not HOSTED:
                c = None   # an IPC-only communicator without its transport cannot exchange anything
        else:
            c = _create(group, device)
            if TRANSPORT == "ipc" or (TRANSPORT == "auto" and (IPC_VERIFIED or c is None)):
                if c is None:
                    c = _create_ipc_only(group, device)
                    if c is not None and not _attach_ipc(c, group, device):
                        c = None
                else:
                    _attach_ipc(c, group, device)
        ent = _comms[key] = c or False
        _groups[key] = group   # held so that the group's identity cannot be recycled under the key
    return ent or None


_ALL = object()


def ipc_status(group=_ALL, device=None):
    """raise if an exchange of the IPC transport on `group`'s communicator gave up (a peer never arrived, or its rows were behind
    its flag).  The forward that gave up has returned already -- its output is poisoned with NaN -- so callers that want the error
    at a definite place (after a synchronize, at the end of a run) ask here; every forward asks on entry anyway."""
    for (gkey, dkey), c in list(_comms.items()):
        if c and c.ipc and (group is _ALL or gkey == _group_key(group)) and (device is None or dkey == str(device)):
            _lib.check(_lib.lib().tutel_amd_ep_ipc_status(c.handle), "tutel_amd_ep_ipc_status")


def destroy_all(check=False):
    """free every communicator and its segments (collective in effect: peers hold mappings).  check=True first raises a pending
    exchange error (after everything has been freed), so that the LAST forward of a run cannot fail silently."""
    L = _lib.lib()
    pending = None
    for c in _comms.values():
        if c:
            if c.ipc and pending is None and L.tutel_amd_ep_ipc_status(c.handle) != 0:
                pending = L.tutel_amd_last_error().decode("utf-8", "replace")
                logging.error("tutel_amd: %s", pending)
            L.tutel_amd_ep_comm_destroy(c.handle)
            for sg in list(c.segments.values()) + [getattr(c, "_flags", None)]:
                if sg is not None:
                    L.tutel_amd_ep_segment_free(sg.handle)
    _comms.clear()
    _groups.clear()
    if check and pending:
        raise _lib.TutelAmdError(pending)


def set_transport(name, hosted=None):
    """switch how the NEXT communicators exchange ("auto" | "ipc" | "rccl"): destroys the existing ones (every rank of every group
    must call this at the same point of its program) -- layers then re-create theirs on their next forward; their cached pipeline
    workspaces belong to the old communicator, so callers drop them with forget_workspaces(layer).  bench.py times the transports
    side by side with this."""
    global TRANSPORT, HOSTED
    assert name in ("auto", "ipc", "rccl"), name
    destroy_all()
    TRANSPORT = name
    if hosted is not None:
        HOSTED = bool(hosted)


def forget_workspaces(layer):
    layer.__dict__.pop("_ep_workspaces", None)


def plan(E, W, capacity, degree, allow_sliced=True):
    """tutel_amd_ep_plan as a dict (CPU-callable: pure arithmetic inside the library)."""
    p = _lib.EpPlan()
    _lib.check(_lib.lib().tutel_amd_ep_plan(E, W, capacity, degree, int(allow_sliced), ctypes.byref(p)), "tutel_amd_ep_plan")
    return {n: int(getattr(p, n)) for n, _ in _lib.EpPlan._fields_}


def usable(layer, x, crit, degree):
    """Can this forward run through tutel_amd_ep_forward?  (Everything else takes the Python-orchestrated paths.)"""
    if not ENABLED or crit.gates2d is None or crit[4] <= 0:
        return False
    W = layer.world_size
    if getattr(layer, "megablocks_size", 0) > 0 and (W > 1 or not layer.is_postscore):
        return False  # row counts ride on the single-rank fused-encode route only
    if W > 1:
        if not group_ok(layer.group):
            return False  # gloo rendezvous (ranks sharing a GPU in the tests): host-staged exchange in impls/overlap.py
    return crit[4] % max(degree, 1) == 0 and degree <= 32


FUSE_ENCODE = int(os.environ.get("TUTEL_AMD_FUSE_ENCODE", "1")) != 0   # A/B switch: single-rank fc1 gathers its rows from the tokens
WS_MAX = 4   # workspaces kept per layer (least recently used goes first)


def _bucket_tokens(T):
    """token-count bucket of a workspace: the next power of two (at least 256) -- growing batches reallocate O(log T) times
    (288 GB of HBM: a workspace up to twice the size needed costs nothing that matters)"""
    T = max(int(T), 256)
    return 1 << (T - 1).bit_length()


def _bucket_capacity(C):
    """rows per expert of a workspace: the next power of two (at least 32).  A function of the capacity ALONE (scaling
    it by T_cap / T blew up for ranks with few or no tokens under an agreed capacity -- inequivalent_tokens -- and for T < E), so
    every rank of a group derives the same value and a workspace is never more than twice the rows any call needs"""
    C = max(int(C), 32)
    return 1 << (C - 1).bit_length()


class _Workspace:
    """the buffers of one pipeline configuration + its argument struct.  Buffers are sized for a BUCKET (T_cap tokens,
    C_cap rows per expert): any call with T <= T_cap and capacity <= C_cap reuses them -- the strides the kernels use come
    from the call's own T / capacity in the argument struct, the allocation only has to be large enough."""

    def __init__(self, layer, x, E, C_cap, k, degree, comm, T_cap):
        ex = layer.experts
        W = layer.world_size
        M = x.shape[1]
        H, Mo = ex.batched_fc1_w.size(1), ex.output_dim
        dev, dt = x.device, x.dtype
        E_loc = E // W
        self.T_cap, self.C_cap = int(T_cap), int(C_cap)
        fuse = comm is None and degree <= 1 and layer.is_postscore and FUSE_ENCODE
        a = _lib.EpArgs()
        a.M, a.H, a.M_out, a.num_experts, a.world, a.k, a.degree = M, H, Mo, E, W, k, max(degree, 1)
        a.allow_sliced, a.dtype, a.act = 1, ops._DT[dt], ops.ACT_CODES[ex.fused_activation()]
        a.is_postscore, a.fuse_encode = int(bool(layer.is_postscore)), int(fuse)
        self.bufs = {}

        def buf(name, rows, cols, alias=None):
            t = self.bufs[alias] if alias is not None else torch.empty([rows, cols], dtype=dt, device=dev)
            self.bufs[name] = t
            setattr(a, name, t.data_ptr())
        buf("hid", E_loc * W * C_cap, H)
        self.segment = None
        if comm is not None and comm.ipc:
            # IPC transport: the receive / return arrays live in a peer-mapped segment (same offsets on every rank); the encode
            # and send staging arrays do not exist -- the producing kernels store into the peers' arrays directly
            es = x.element_size()
            recv_bytes = (E * C_cap * M * es + 255) // 256 * 256
            sg = comm.segment(("ep", M, Mo, str(dt), E, C_cap), recv_bytes + E * C_cap * Mo * es)
            a.recv, a.back, a.peer_seg = sg.ptr, sg.ptr + recv_bytes, sg.handle
            self.segment = sg
            self.args, self.comm = a, comm
            return
        buf("send", E * C_cap, Mo)
        if not fuse:
            buf("enc", E * C_cap, M)
            # without a communicator the "exchange" is the identity: the stage buffers alias and no copy is made
            buf("recv", E * C_cap, M, alias="enc" if comm is None else None)
            buf("back", E * C_cap, Mo, alias="send" if comm is None else None)
        else:
            z = _zero_rows.get((dev, dt))
            if z is None or z.numel() < M:
                z = _zero_rows[(dev, dt)] = torch.zeros([max(M, 8192)], dtype=dt, device=dev)
            self.bufs["zero_row"] = z
            a.zero_row = z.data_ptr()
        self.args, self.comm = a, comm
        if comm is not None and hasattr(comm, "register"):
            for t in self.bufs.values():
                comm.register(t)


def _workspace(layer, static_key, T, C, make, exact_capacity=False):
    """least-recently-used cache of at most WS_MAX workspaces per layer; a hit is ANY workspace of the same configuration
    that is large enough (variable token counts -- serving -- keep hitting the largest one allocated so far).
    exact_capacity (the IPC transport): the capacity bucket decides WHICH peer-mapped segment the kernels store into and where
    `back` starts inside it, so it must be the same on every rank for a given call -- a function of the call's (agreed) capacity
    alone, never of which workspaces this rank happens to have cached (its token count differs from its peers': ADVICE r4 high).
    A hit then needs C_cap == bucket(C); only the token bucket, which no peer ever sees, may be larger than needed."""
    import collections
    cache = layer.__dict__.get("_ep_workspaces")
    if not isinstance(cache, collections.OrderedDict):
        cache = layer.__dict__["_ep_workspaces"] = collections.OrderedDict()
    C_cap = _bucket_capacity(C)
    for key, ws in cache.items():
        if key[0] == static_key and ws.T_cap >= T and (ws.C_cap == C_cap if exact_capacity else ws.C_cap >= C):
            cache.move_to_end(key)
            return ws
    T_cap = _bucket_tokens(T)
    ws = make(T_cap, C_cap)
    cache[(static_key, T_cap, C_cap)] = ws
    layer.__dict__["_ep_workspace_allocations"] = layer.__dict__.get("_ep_workspace_allocations", 0) + 1
    while len(cache) > WS_MAX:
        cache.popitem(last=False)
    return ws


def forward(layer, x, crit, degree):
    """x [T, M] (contiguous, expert dtype) -> y [T, M_out].  The caller has checked usable() and experts.can_fuse()."""
    ex = layer.experts
    W = layer.world_size
    with_comm = W > 1 or (_FORCE_COMM and dist.is_initialized())
    comm = communicator(layer.group, x.device) if with_comm else None
    if with_comm and comm is None:
        return None
    if getattr(layer, "megablocks_size", 0) > 0 and with_comm:
        return None
    if not with_comm:
        degree = 1  # a single rank has nothing to overlap (the reference returns expert_fn(input) there, overlap.py:16-17)
    k = crit.idx2d.shape[0]
    key = ("ep", x.shape[1], x.dtype, x.device, crit[0], k, degree, bool(layer.is_postscore), ex.fused_activation(), ops._stream(), with_comm,
           bool(comm is not None and comm.ipc))
    ws = _workspace(layer, key, x.shape[0], crit[4], lambda Tc, Cc: _Workspace(layer, x, crit[0], Cc, k, degree, comm, Tc),
                    exact_capacity=bool(comm is not None and comm.ipc))
    a = ws.args
    a.T, a.capacity = x.shape[0], crit[4]
    w1, b1, w2, b2, kmajor = ex.fused_params(x.dtype)
    gates = crit.gates2d
    y = torch.empty([x.shape[0], ex.output_dim], dtype=x.dtype, device=x.device)
    a.w2_kmajor = int(kmajor)
    a.x, a.slot_map, a.idx, a.loc, a.gates = x.data_ptr(), crit.slot_map.data_ptr(), crit.idx2d.data_ptr(), crit.loc2d.data_ptr(), gates.data_ptr()
    a.gate_dtype = ops._DT[gates.dtype]
    a.w1, a.w2 = w1.data_ptr(), w2.data_ptr()
    a.b1 = b1.data_ptr() if b1 is not None else None
    a.b2 = b2.data_ptr() if b2 is not None else None
    a.y = y.data_ptr()
    mega = int(getattr(layer, "megablocks_size", 0))
    if mega > 0 and not with_comm:
        a.row_counts, a.row_align = layer.dispatch_count.data_ptr(), mega
    else:
        a.row_counts, a.row_align = None, 1
    _lib.check(_lib.lib().tutel_amd_ep_forward(comm.handle if comm is not None else None, ctypes.byref(a), ops._stream()),
               "tutel_amd_ep_forward")
    layer.protected_shape = torch.Size([layer.num_local_experts, W * crit[4], ex.output_dim])
    return y


# ---------------------------------------------------------------------------------------------
# routing + pipeline in one call (tutel_amd_moe_forward)
# ---------------------------------------------------------------------------------------------
class _MoeWorkspace(_Workspace):
    """_Workspace + the routing buffers (idx / loc / gates / slot map / per-tile histograms), all reused call after call"""

    def __init__(self, layer, x, logits, k, capacity, degree, comm, T_cap):
        E, T = logits.shape[1], int(T_cap)
        dev = x.device
        super().__init__(layer, x, E, capacity, k, degree, comm, T_cap)
        self.idx = torch.empty([k, T], dtype=torch.int32, device=dev)
        self.loc = torch.empty([k, T], dtype=torch.int32, device=dev)
        self.gates = torch.empty([k, T], dtype=logits.dtype, device=dev)
        self.slot_map = torch.empty([E * capacity], dtype=torch.int32, device=dev)
        # routing scratch for the largest tiling any T takes (<= 128 token tiles, csrc/routing.hip)
        self.ws = torch.empty([max(int(_lib.lib().tutel_amd_routing_workspace_bytes(64 * 128, E, k)), 4)], dtype=torch.uint8, device=dev)
        self.stats = torch.empty([1], dtype=torch.int32, device=dev)
        # the dropless capacity is read back into this workspace's own pinned word (no process-global slot: ADVICE r2)
        self.cap_host = torch.zeros([1], dtype=torch.int32).pin_memory()
        self.cap_c = ctypes.c_int.from_address(self.cap_host.data_ptr())
        m = _lib.MoeArgs()
        m.ep = self.args
        m.ep.slot_map, m.ep.idx, m.ep.loc, m.ep.gates = (self.slot_map.data_ptr(), self.idx.data_ptr(), self.loc.data_ptr(),
                                                          self.gates.data_ptr())
        m.logits_dtype = ops._DT[logits.dtype]
        m.ws, m.ws_bytes, m.stats = self.ws.data_ptr(), self.ws.numel(), self.stats.data_ptr()
        m.capacity_out = ctypes.pointer(self.cap_c)
        self.margs = m
        self.gate_partials, self.gate_partials_keep = None, []   # split-K partial sums of the in-call gate projection (on demand)
        # fused location (single rank; csrc/expert_gemm.hip FL kernels): the top-k kernel's byte copy of idx, k * T bytes padded to 16
        self.fl_ws = None
        if comm is None and _FUSED_LOCATION:
            self.fl_ws = torch.empty([(k * T + 15) // 16 * 16 + 16], dtype=torch.uint8, device=dev)
            m.fl_ws, m.fl_ws_bytes = self.fl_ws.data_ptr(), self.fl_ws.numel()


def forward_from_logits(layer, x, logits, k, capacity, degree, normalize_gate, want_loss, dropless=None, megablocks_size=0, gate_w=None):
    """x [T, M], logits [T, E] -> (y [T, M_out], l_aux | None, dispatch_count [E], capacity); None when the native path is
    unavailable.  One C call: softmax + top-k + locations + loss, encode, exchange(s), expert FFN, exchange(s), decode.
    dropless = (capacity_limit, alignment): capacity_factor <= 0 on a single rank -- the capacity is read back inside the call
    (`capacity` is then only the first guess for the workspace size)."""
    ex = layer.experts
    W = layer.world_size
    with_comm = W > 1 or (_FORCE_COMM and dist.is_initialized())
    if dropless is not None and with_comm:
        return None
    comm = communicator(layer.group, x.device) if with_comm else None
    if with_comm and comm is None:
        return None
    if not with_comm:
        degree = 1
    sizes = layer.__dict__.setdefault("_ep_dropless_cap", {})
    skey = (tuple(x.shape), logits.shape[1], k)
    if dropless is not None:
        capacity = max(capacity, sizes.get(skey, 0))
    for attempt in range(4):
        key = ("moe", x.shape[1], x.dtype, x.device, logits.shape[1], logits.dtype, k, degree, bool(layer.is_postscore),
               ex.fused_activation(), ops._stream(), with_comm, bool(comm is not None and comm.ipc))
        ws = _workspace(layer, key, x.shape[0], capacity, lambda Tc, Cc: _MoeWorkspace(layer, x, logits, k, Cc, degree, comm, Tc),
                        exact_capacity=bool(comm is not None and comm.ipc))
        m = ws.margs
        a = m.ep
        a.T = x.shape[0]
        w1, b1, w2, b2, kmajor = ex.fused_params(x.dtype)
        dev = x.device
        y = torch.empty([x.shape[0], ex.output_dim], dtype=x.dtype, device=dev)
        cnt = torch.empty([logits.shape[1]], dtype=torch.int32, device=dev)
        l_aux = torch.empty([1], dtype=logits.dtype, device=dev) if want_loss else None
        a.w2_kmajor = int(kmajor)
        a.x, a.w1, a.w2 = x.data_ptr(), w1.data_ptr(), w2.data_ptr()
        a.b1 = b1.data_ptr() if b1 is not None else None
        a.b2 = b2.data_ptr() if b2 is not None else None
        a.y = y.data_ptr()
        if megablocks_size > 0 and not with_comm:
            a.row_counts, a.row_align = cnt.data_ptr(), int(megablocks_size)
        else:
            a.row_counts, a.row_align = None, 1
        m.normalize_gate = int(bool(normalize_gate))
        if gate_w is None:
            m.logits, m.gate_w, m.logits_out = logits.data_ptr(), None, None
        else:   # `logits` is a meta tensor (shape / dtype only): the projection runs inside the call (csrc/gate_proj.hip)
            need = ops.gate_proj_splits(x.shape[0], x.shape[1], logits.shape[1], x.dtype) * x.shape[0] * logits.shape[1]
            if ws.gate_partials is None or ws.gate_partials.numel() < need:
                ws.gate_partials_keep.append(ws.gate_partials)   # a captured graph may still hold the old pointer
                ws.gate_partials = torch.empty([need], dtype=torch.float32, device=dev)
            m.logits, m.gate_w = None, gate_w.data_ptr()
            m.gate_partials, m.gate_partial_bytes = ws.gate_partials.data_ptr(), ws.gate_partials.numel() * 4
            layer.last_logits = torch.empty(list(logits.shape), dtype=x.dtype, device=dev) if getattr(layer, "_keep_routing", False) else None
            m.logits_out = layer.last_logits.data_ptr() if layer.last_logits is not None else None
        m.dispatch_count = cnt.data_ptr()
        m.l_aux = l_aux.data_ptr() if l_aux is not None else None
        cap_out = ws.cap_c
        if dropless is not None:
            a.capacity = 0
            m.capacity_limit, m.alignment, m.max_capacity = int(dropless[0]), int(dropless[1]), int(ws.C_cap)
        else:
            a.capacity = int(capacity)
            m.capacity_limit, m.alignment, m.max_capacity = 0, 1, int(capacity)
        rc = _lib.lib().tutel_amd_moe_forward(comm.handle if comm is not None else None, ctypes.byref(m), ops._stream())
        if rc == _lib.EAGAIN and dropless is not None:   # the batch needs more rows per expert than the workspace holds: grow, redo
            capacity = (int(cap_out.value) * 5 // 4 + 31) // 32 * 32
            sizes[skey] = capacity
            continue
        _lib.check(rc, "tutel_amd_moe_forward")
        used = int(cap_out.value) if dropless is not None else int(capacity)
        if getattr(layer, "_keep_routing", False):   # tests: [k, T] views of the workspace's routing arrays (rows are T apart)
            n = k * x.shape[0]
            layer.last_routing = (ws.idx.view(-1)[:n].view(k, -1).clone(), ws.loc.view(-1)[:n].view(k, -1).clone())
        layer.protected_shape = torch.Size([layer.num_local_experts, W * used, ex.output_dim])
        return y, (l_aux[0] if l_aux is not None else None), cnt, used
    raise _lib.TutelAmdError("tutel_amd_moe_forward: the dropless capacity kept growing")
