"""Process groups + collectives for the MoE hot path, over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU multi-process tests).

Mirrors the call surface of the reference's tutel/impls/communicate.py that the forward path and
its users touch:
  get_world_size / get_world_rank / barrier                      (communicate.py:20-36)
  create_groups_from_world / create_standalone_group             (:43-168)
  simple_all_reduce / simple_all_to_all / simple_split /
  simple_reduce_scatter / simple_all_gather                      (:173-224)
  all_to_all(input, input_dim, output_dim, group, background, use_2dh)   (:447-503)
  all_to_all_single, all_gather, zero_gather, zero_scatter, spatial_split,
  reduce_scatter, allreduce_forward, allreduce_backward          (:505-604,624-632)
  pre_expert_permute / post_expert_permute                       (:606-622)

MI355X notes: one process per GPU; within a node xGMI is a full mesh, so an 8-way all-to-all
maps one peer slice per link and hierarchical (2DH) exchange buys nothing -- `use_2dh` is
accepted and produces the identical result through the flat exchange (the reference itself
degenerates to a linear send/recv loop on one node, custom_kernel.cpp:681,722-737).
No private communicator: the overlap path (impls/overlap.py) drives the same RCCL communicator
from a dedicated HIP stream.
"""
import datetime
import logging
import os

import torch
import torch.distributed as dist

GLOBAL_TIMEOUT_SEC = int(os.environ.get("TUTEL_GLOBAL_TIMEOUT_SEC", 86400))
SKIP_A2A = int(os.environ.get("SKIP_A2A", 0)) > 0  # ablation switch kept from the reference (:40)

_GROUP_CACHE = {}


def get_world_size(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


def get_world_rank(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(group)


def barrier(group=None):
    if get_world_size(group) > 1:
        dist.barrier(group=group)


def create_standalone_group():
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist.new_group(ranks=[get_world_rank()])


class DistributedProperties:
    """Bag of rank/group facts returned by create_groups_from_world (same field names as the
    reference so user scripts keep working)."""

    def __repr__(self):
        return "DistributedProperties(%s)" % ", ".join(
            f"{k}={v}" for k, v in self.__dict__.items() if not callable(v) and "group" not in k)


def _timeout():
    return datetime.timedelta(seconds=GLOBAL_TIMEOUT_SEC)


def _init_default_group(backend):
    env = os.environ
    if "LOCAL_RANK" not in env and "OMPI_COMM_WORLD_SIZE" in env:  # mpiexec bootstrap
        dist.init_process_group(
            backend=backend, timeout=_timeout(),
            init_method="tcp://%s:%s" % (env["MASTER_ADDR"], env.get("MASTER_PORT", "23456")),
            rank=int(env["OMPI_COMM_WORLD_RANK"]), world_size=int(env["OMPI_COMM_WORLD_SIZE"]))
    else:  # torchrun env (RANK / WORLD_SIZE / MASTER_*)
        dist.init_process_group(backend=backend, timeout=_timeout())


def create_groups_from_world(group_count, include_init=None, parent_group=None):
    """Split the world into `group_count` data-parallel groups (negative: groups of that size).
    model_group = my contiguous block of ranks, data_group = ranks with my position in their block."""
    if 1 < get_world_size(parent_group) < get_world_size():
        raise Exception("Splitting nesting groups from a subgroup is yet not allowed, please report an issue for this requirement.")

    backend = _GROUP_CACHE.get("", include_init)
    if include_init:
        assert backend == include_init, "Only 1 backend type is allowed, get: %s v.s. %s" % (backend, include_init)
        _GROUP_CACHE[""] = backend
    if group_count in _GROUP_CACHE:
        return _GROUP_CACHE[group_count]

    distributed = True
    local_rank = 0
    if include_init and not dist.is_initialized():
        if "RANK" in os.environ or "OMPI_COMM_WORLD_SIZE" in os.environ:
            _init_default_group(backend)
        else:
            distributed = False
    elif not dist.is_initialized():
        distributed = False
    if distributed:
        world, rank = dist.get_world_size(), dist.get_rank()
        if "LOCAL_RANK" not in os.environ and "OMPI_COMM_WORLD_LOCAL_RANK" in os.environ:
            local_rank = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        else:
            local_rank = int(os.environ.get("LOCAL_RANK", 0))
            if torch.cuda.is_available():
                local_rank = min(local_rank, torch.cuda.device_count() - 1)
    else:
        world, rank = 1, 0

    requested = group_count
    n_groups = world // -group_count if group_count < 0 else group_count
    assert n_groups > 0 and world % n_groups == 0, \
        f"Expected to evenly divide devices into {n_groups} groups, while the world size of current sesion is {world}."
    block = world // n_groups  # ranks per model group

    model_group = data_group = global_group = (dist.group.WORLD if distributed else None)
    if distributed and block != world:
        for b in range(n_groups):  # every rank must create every group, in the same order
            ranks = list(range(b * block, (b + 1) * block))
            grp = dist.new_group(ranks=ranks, timeout=_timeout())
            if rank // block == b:
                model_group = grp
    if distributed and n_groups != world:
        for pos in range(block):
            ranks = list(range(pos, world, block))
            grp = dist.new_group(ranks=ranks, timeout=_timeout())
            if rank % block == pos:
                data_group = grp

    res = DistributedProperties()
    res.global_size, res.global_rank = world, rank
    res.group_count, res.data_rank = n_groups, rank // block
    res.model_size, res.model_rank = block, rank % block
    if backend == "nccl":
        res.local_device = torch.device("cuda", local_rank)
        torch.cuda.set_device(res.local_device)
    elif backend == "gloo":
        res.local_device = torch.device("cpu")
    elif backend is None:
        res.local_device = None
    else:
        raise Exception("Unsupported backend type: %s" % backend)
    res.data_group, res.model_group, res.global_group = data_group, model_group, global_group
    res.is_distributed = distributed
    res.dist_print = (lambda *a: print(*a) if rank == 0 else None) if distributed else print
    _GROUP_CACHE[requested] = res
    return res


# ---------------------------------------------------------------------------------------------
# collectives without autograd
# ---------------------------------------------------------------------------------------------
def simple_all_reduce(input, group=None, op=dist.ReduceOp.SUM, inplace=False):
    if get_world_size(group) == 1:
        return input
    out = input if inplace else input.clone(memory_format=torch.contiguous_format)
    dist.all_reduce(out, op=op, group=group)
    return out


def exchange_equal_split(out, input, group=None):
    """all_to_all_single with equal dim-0 splits on the CURRENT stream.  Device tensors on an RCCL
    ("nccl") group go straight to the library.  A gloo group has no device all-to-all: the message is
    staged through host memory (stream-synchronous), which keeps every expert-parallel code path runnable
    when several ranks share ONE GPU or the job was brought up with a CPU rendezvous -- that is how the
    multi-rank GPU tests exercise the W > 1 kernels' addressing on a single-GPU box."""
    backend = _backend_of(group)
    if input.is_cuda and backend == "gloo":
        host_in = input.cpu()
        host_out = torch.empty_like(host_in)
        dist.all_to_all_single(host_out, host_in, group=group)
        out.copy_(host_out)
        return
    if backend == "nccl" and _DIRECT_PG:
        # straight to the process group: dist.all_to_all_single re-validates its arguments and goes through the
        # c10d logging wrapper on every call (~10 us of the ~25 us a call costs the host); four calls per forward
        # on a path that is host-bound
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        pg.alltoall_base(out, input, [], []).wait()
        return
    dist.all_to_all_single(out, input, group=group)


_DIRECT_PG = hasattr(dist, "ProcessGroup") and hasattr(dist.ProcessGroup, "alltoall_base")


def _backend_of(group):
    return dist.get_backend(group)


def simple_all_to_all(input, group=None, background=False):
    """Equal-split exchange of dim-0 chunks: chunk r of my tensor goes to rank r."""
    input = input.contiguous()
    if get_world_size(group) == 1 or SKIP_A2A:
        return input if not background else (input, lambda *a: None)
    out = torch.empty_like(input)
    if background and not (input.is_cuda and _backend_of(group) == "gloo"):
        work = dist.all_to_all_single(out, input, group=group, async_op=True)
        return out, work.wait
    exchange_equal_split(out, input, group)
    return (out, lambda *a: None) if background else out


def simple_split(input, group=None):
    W = get_world_size(group)
    if W == 1:
        return input
    assert input.size(0) % W == 0, "Cannot evenly divide dim length %s into %s slices" % (input.size(0), W)
    return input.contiguous().chunk(W, dim=0)[get_world_rank(group)]


def simple_reduce_scatter(input, group=None, op=dist.ReduceOp.SUM):
    W = get_world_size(group)
    if W == 1:
        return input
    input = input.contiguous()
    assert input.size(0) % W == 0, "Cannot evenly divide dim length %s into %s slices" % (input.size(0), W)
    if not input.is_cuda:  # gloo has no reduce_scatter
        return simple_split(simple_all_reduce(input, group, op=op), group=group)
    out = torch.empty_like(input.chunk(W, dim=0)[0])
    dist.reduce_scatter_tensor(out, input, op=op, group=group)
    return out


def simple_all_gather(input, group=None):
    W = get_world_size(group)
    if W == 1:
        return input
    input = input.contiguous()
    out = torch.empty([W * input.numel()], device=input.device, dtype=input.dtype)
    dist.all_gather_into_tensor(out, input.view(-1), group=group)
    return out.view([-1] + list(input.shape[1:]))


def _staged(t, group):
    """device tensor on a gloo group -> host copy (see exchange_equal_split), else the tensor itself"""
    return t.cpu() if t.is_cuda and dist.get_backend(group) == "gloo" else t


def _native_comm(t, group):
    """the library's own communicator for device tensors (RCCL; or host-staged when the ranks share a GPU in tests):
    the variable-size collectives then run as ONE grouped ncclSend / ncclRecv loop per tensor inside libtutel_amd.so, as the
    reference's do inside its extension (custom_kernel.cpp:463-518).  None -> torch.distributed's all_to_all_single."""
    if not t.is_cuda:
        return None
    from . import ep_native
    if not ep_native.ENABLED or not (dist.get_backend(group) == "nccl" or ep_native.HOSTED):
        return None
    c = ep_native.communicator(group, t.device)   # collective on first use; every rank of the group is in this call
    return c if c is not None and c.generic else None


def batch_all_to_all_v(datas, partition_sizes, group=None):
    """Variable-size all-to-all of a batch of flat tensors sharing one split (reference:
    communicate.py:225-241 over custom_kernel.cpp:463-491, a grouped ncclSend/ncclRecv loop):
    partition_sizes[r] elements of every tensor go to rank r.  Returns (outputs, out_sizes) where
    out_sizes[r] = number of elements received from rank r.  Here: all_to_all_single with split lists."""
    assert type(datas) in (tuple, list), "data type for batch_all_to_all_v() is not a list of tensors"
    in_sizes = partition_sizes
    if not torch.is_tensor(in_sizes):
        in_sizes = torch.tensor(in_sizes, dtype=torch.int64, device=datas[0].device)
    else:
        in_sizes = in_sizes.to(torch.int64)
    world_size = get_world_size(group)
    assert in_sizes.numel() == world_size
    if world_size == 1:
        return list(datas), in_sizes
    out_sizes = simple_all_to_all(in_sizes, group=group)
    send, recv = [int(v) for v in in_sizes.tolist()], [int(v) for v in out_sizes.tolist()]  # the one sync the API implies
    comm = _native_comm(datas[0], group)
    outputs = []
    for data in datas:
        flat = data.contiguous().view(-1)
        assert flat.numel() == sum(send), "Tensor instances within batch_all_to_all_v are supposed to share same length."
        if comm is not None:
            outputs.append(comm.all_to_all_v(flat, send, recv))
            continue
        src = _staged(flat, group)
        out = torch.empty([sum(recv)], dtype=flat.dtype, device=src.device)
        dist.all_to_all_single(out, src, output_split_sizes=recv, input_split_sizes=send, group=group)
        outputs.append(out.to(flat.device))
    return outputs, out_sizes


def batch_all_gather_v(datas, group=None):
    """Variable-size all-gather of a batch of flat tensors (reference: communicate.py:243-255 over
    custom_kernel.cpp:493-518): every rank receives the concatenation, in rank order, of all ranks'
    tensors.  Returns (outputs, output_sizes)."""
    assert type(datas) in (tuple, list), "data type for batch_all_gather_v() is not a list of tensors"
    datas = [data.contiguous().view(-1) for data in datas]
    input_size = torch.tensor([int(datas[0].numel())], dtype=torch.int64, device=datas[0].device)
    world_size = get_world_size(group)
    if world_size == 1:
        return list(datas), input_size
    output_sizes = simple_all_gather(input_size, group=group)
    recv = [int(v) for v in output_sizes.tolist()]
    comm = _native_comm(datas[0], group)
    outputs = []
    for flat in datas:
        assert flat.numel() == int(input_size), "Tensor instances within batch_all_gather_v are supposed to share same length."
        if comm is not None:
            outputs.append(comm.all_gather_v(flat, recv))
            continue
        src = _staged(flat, group)
        # my tensor to every rank, every rank's tensor to me: an all-to-all whose input is W copies of the tensor
        out = torch.empty([sum(recv)], dtype=flat.dtype, device=src.device)
        dist.all_to_all_single(out, src.repeat(world_size), output_split_sizes=recv,
                               input_split_sizes=[flat.numel()] * world_size, group=group)
        outputs.append(out.to(flat.device))
    return outputs, output_sizes


# ---------------------------------------------------------------------------------------------
# collectives with autograd
# ---------------------------------------------------------------------------------------------
class _A2A(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, group):
        ctx.group = group
        return simple_all_to_all(input, group)

    @staticmethod
    def backward(ctx, grad):
        return _A2A.apply(grad, ctx.group), None


def all_to_all_single(input, group=None):
    return _A2A.apply(input, group)


def _a2a_gather_dim(x, dim, group):
    """a2a that concatenates the received chunks along `dim` and splits dim 0:
    [W*a, ..., d, ...] -> [a, ..., W*d, ...] (source-rank-major along dim)."""
    W = get_world_size(group)
    recv = _A2A.apply(x, group)                      # [W(src) * a, ...]
    recv = recv.view([W, -1] + list(recv.shape[1:]))  # [W, a, ...]
    order = list(range(1, dim + 1)) + [0] + list(range(dim + 1, recv.dim()))
    recv = recv.permute(order).contiguous()           # [a, ..., W, d, ...]
    shape = list(recv.shape)
    return recv.view(shape[:dim] + [-1] + shape[dim + 2:])


def _a2a_scatter_dim(x, dim, group):
    """inverse: [a, ..., W*d, ...] -> [W*a, ..., d, ...]."""
    W = get_world_size(group)
    shape = list(x.shape)
    x = x.view(shape[:dim] + [W, -1] + shape[dim + 1:])
    order = [dim] + list(range(dim)) + list(range(dim + 1, x.dim()))
    x = x.permute(order).contiguous()                 # [W(dst), a, ..., d, ...]
    recv = _A2A.apply(x, group)
    return recv.view([-1] + list(recv.shape[2:]))


class _A2A2DH(torch.autograd.Function):
    """Two-phase (intra-node then inter-node) exchange, reference communicate.py:412-430.  Only
    meaningful with LOCAL_SIZE < world size; on one xGMI node it is the flat exchange."""

    @staticmethod
    def forward(ctx, x, input_dim, output_dim):
        ctx.dims = (input_dim, output_dim)
        local = int(os.environ.get("LOCAL_SIZE", 1))
        world = get_world_size()
        if local <= 1 or local >= world:
            if local <= 1 and world > 1:
                logging.warning("LOCAL_SIZE (> 1) for AllToAll 2DH is not set; using the flat all-to-all (identical result).")
            return all_to_all(x, input_dim, output_dim)
        d = create_groups_from_world(-local)
        y = all_to_all(x, input_dim, output_dim, group=d.data_group)
        y = all_to_all(y, input_dim, output_dim, group=d.model_group)
        nm, nd = get_world_size(d.model_group), get_world_size(d.data_group)
        shp = y.shape
        y = y.view(list(shp[:input_dim]) + [nm, nd, -1] + list(shp[input_dim + 1:]))
        return y.swapaxes(input_dim, input_dim + 1).contiguous().view(shp)

    @staticmethod
    def backward(ctx, dy):
        return _A2A2DH.apply(dy, ctx.dims[1], ctx.dims[0]), None, None


def all_to_all(input, input_dim, output_dim, group=None, background=False, use_2dh=False):
    """Flexible all-to-all: the result has dim `input_dim` W times longer and dim `output_dim` W
    times shorter ("[HY] X LY Z -> [HX] HY LX LY Z" in the reference's words).
    all_to_all(y, 1, 0): [E, C, M] -> [E/W, W*C, M];  all_to_all(y, 0, 1): the inverse."""
    if use_2dh:
        assert not background, "Background mode for AllToAll 2DH is not implemented."
        return _A2A2DH.apply(input, input_dim, output_dim)
    W = get_world_size(group)
    if input_dim == output_dim or W == 1:
        return (lambda *a: input) if background else input

    def run():
        if output_dim == 0:
            return _a2a_gather_dim(input, input_dim, group)
        if input_dim == 0:
            return _a2a_scatter_dim(input, output_dim, group)
        x = input.swapaxes(0, output_dim)
        x = all_to_all(x, input_dim, 0, group=group)
        return x.swapaxes(0, output_dim).contiguous()

    if background:
        if input_dim != 0 and output_dim != 0:
            raise Exception("Unhandle async branch case for flexible all_to_all()")
        return run  # deferred: evaluated when the caller invokes it
    return run()


class _AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, fused, group):
        ctx.group, ctx.fused = group, fused
        return simple_all_gather(input, group)

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            return simple_reduce_scatter(g, ctx.group), None, None
        return simple_split(g, ctx.group), None, None


class _ReduceScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, op, group):
        ctx.group = group
        return simple_reduce_scatter(input, group, op=op)

    @staticmethod
    def backward(ctx, g):
        return simple_all_gather(g, ctx.group), None, None


class _Split(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, group):
        ctx.group = group
        return simple_split(input, group)

    @staticmethod
    def backward(ctx, g):
        return simple_all_gather(g, ctx.group), None


class _FwdAllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, op, group):
        return simple_all_reduce(input, group=group)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class _BwdAllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, op, group):
        ctx.group, ctx.op = group, op
        return input

    @staticmethod
    def backward(ctx, g):
        return simple_all_reduce(g, group=ctx.group, op=ctx.op), None, None


def _on_dim0(fn, input, dim):
    x = input if dim == 0 else input.swapaxes(0, dim)
    x = fn(x)
    return x if dim == 0 else x.swapaxes(0, dim)


def all_gather(input, dim, fused=False, group=None):
    return _on_dim0(lambda x: _AllGather.apply(x, fused, group), input, dim)


def reduce_scatter(input, dim, group=None):
    return _on_dim0(lambda x: _ReduceScatter.apply(x, dist.ReduceOp.SUM, group), input, dim)


def spatial_split(input, dim, group=None):
    return _on_dim0(lambda x: _Split.apply(x, group), input, dim)


def allreduce_forward(input, op=dist.ReduceOp.SUM, group=None):
    return _FwdAllReduce.apply(input, op, group)


def allreduce_backward(input, op=dist.ReduceOp.SUM, group=None):
    return _BwdAllReduce.apply(input, op, group)


def zero_gather(input, full_shape=None, group=None):
    """All-gather a dim-0 sharded parameter (ZeRO-style); backward = reduce-scatter."""
    if not full_shape:
        full_shape = list(input.shape)
        full_shape[0] *= get_world_size(group)
    numel = 1
    for s in full_shape:
        numel *= int(s)
    return _AllGather.apply(input, True, group).view(-1)[:numel].view(full_shape)


def zero_scatter(input, scatter_fn, group=None):
    W = get_world_size(group)
    n = input.numel()
    if n % W == 0:
        flat = input.reshape(-1)
    else:
        flat = torch.zeros([(n + W - 1) // W * W], device=input.device, dtype=input.dtype)
        flat[:n] = input.reshape(-1)
    return scatter_fn(flat, group=group), input.shape


# ---------------------------------------------------------------------------------------------
# expert-parallel row layout
# ---------------------------------------------------------------------------------------------
def pre_expert_permute(input, group=None):
    """raw a2a output [W*E_loc, c, M] -> expert input [E_loc, W*c, M] (source-rank-major rows)."""
    W = get_world_size(group)
    if W == 1:
        return input
    x = input.view([W, -1] + list(input.shape[1:]))                   # [W, E_loc, c, ...]
    x = x.transpose(0, 1).contiguous()                                # [E_loc, W, c, ...]
    return x.view([x.shape[0], -1] + list(x.shape[3:]))


def post_expert_permute(input, group=None):
    """expert output [E_loc, W*c, M] -> a2a input [W*E_loc, c, M]."""
    W = get_world_size(group)
    if W == 1:
        return input
    x = input.view([input.shape[0], W, -1] + list(input.shape[2:]))   # [E_loc, W, c, ...]
    x = x.transpose(0, 1).contiguous()                                # [W, E_loc, c, ...]
    return x.view([-1] + list(x.shape[2:]))
