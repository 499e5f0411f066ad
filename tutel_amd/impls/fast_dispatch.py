"""Top-k routing and the encode/decode dispatch ops behind Tutel's low-level API
(reference: tutel/impls/fast_dispatch.py).

    crit, l_aux = extract_critical(scores, top_k, ...)      # == tutel.moe.top_k_routing
    y = fast_encode(x, crit)          # [T,M] -> [E,C,M]
    o = fast_decode(y, crit)          # [E,C,M] -> [T,M]
    d = fast_dispatcher(E, C, M, dtype); d.update(idx, loc, gates, capacity); d.encode / d.decode

`crit` is the reference's 6-tuple (E, [idx_k], [loc_k], [gate_k], capacity, dispatch_count); the
per-choice vectors are rows of single [k,T] device arrays, which is what the HIP kernels take.
All device work goes through tutel_amd.ops (C ABI); nothing here computes on the CPU.
"""
import logging

import torch

from . import losses
from .communicate import get_world_rank, simple_all_reduce
from .. import ops


class RoutingPlan(tuple):
    """The reference's `critical data` tuple plus the packed [k,T] arrays it was cut from."""

    def __new__(cls, E, idx2d, loc2d, gates2d, capacity, dispatch_count, slot_map=None, gate_list=None):
        k = idx2d.shape[0]
        self = super().__new__(cls, (int(E), [idx2d[j] for j in range(k)], [loc2d[j] for j in range(k)],
                                     gate_list if gate_list is not None else [gates2d[j] for j in range(k)],
                                     int(capacity), dispatch_count))
        self.idx2d, self.loc2d, self.gates2d, self.slot_map = idx2d, loc2d, gates2d, slot_map
        return self


def _pack(rows, dtype=None):
    """list of k [T] vectors -> one contiguous [k,T] array (no copy when they already are rows of one)."""
    rows = [r.view(-1) for r in rows]
    if dtype is not None:
        rows = [r if r.dtype == dtype else r.to(dtype) for r in rows]
    base = rows[0]
    T = base.numel()
    if all(r.is_contiguous() and r.data_ptr() == base.data_ptr() + j * T * base.element_size()
           and r.dtype == base.dtype for j, r in enumerate(rows)) and not any(r.requires_grad for r in rows):
        try:
            return base.as_strided([len(rows), T], [T, 1])
        except RuntimeError:
            pass
    return torch.stack(rows).contiguous()


class _Encode(torch.autograd.Function):
    """fast_encode with autograd (reference GatingEncoder, fast_dispatch.py:16-47).
    backward(data) is a fast_decode of the incoming gradient, backward(gates) the gate-grad kernel."""

    @staticmethod
    def forward(ctx, cfg, x, gates2d):
        ctx.cfg = cfg
        ctx.save_for_backward(x, gates2d)
        return ops.fast_encode(x, cfg.slot_map, gates2d, cfg.E * cfg.capacity)

    @staticmethod
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        x, gates2d = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gx = ops.fast_decode(grad_out, cfg.idx2d, cfg.loc2d, gates2d, cfg.capacity) if ctx.needs_input_grad[1] else None
        gg = None
        if gates2d is not None and ctx.needs_input_grad[2]:
            gg = ops.gate_grad(x, grad_out, cfg.idx2d, cfg.loc2d, cfg.capacity).to(gates2d.dtype)
        return None, gx, gg


class _Decode(torch.autograd.Function):
    """fast_decode with autograd (reference GatingDecoder, fast_dispatch.py:50-82)."""

    @staticmethod
    def forward(ctx, cfg, y, gates2d):
        ctx.cfg = cfg
        ctx.save_for_backward(y, gates2d)
        return ops.fast_decode(y, cfg.idx2d, cfg.loc2d, gates2d, cfg.capacity)

    @staticmethod
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        y, gates2d = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gy = ops.fast_encode(grad_out, cfg.slot_map, gates2d, cfg.E * cfg.capacity) if ctx.needs_input_grad[1] else None
        gg = None
        if gates2d is not None and ctx.needs_input_grad[2]:
            gg = ops.gate_grad(grad_out, y, cfg.idx2d, cfg.loc2d, cfg.capacity).to(gates2d.dtype)
        return None, gy, gg


class TutelMoeFastDispatcher:
    """Holds one routing decision and applies it (reference fast_dispatch.py:85-134)."""

    def __init__(self, num_global_experts, capacity, model_dim, dispatch_dtype):
        self.E = self.num_global_experts = int(num_global_experts)
        self.capacity = int(capacity)
        self.model_dim = int(model_dim)
        self.original_dtype = dispatch_dtype
        # the kernels compute in fp32 internally for every storage dtype (the reference casts
        # to fp32 around its kernels on HIP, fast_dispatch.py:94-96); fp64 goes through fp32
        # exactly as it does there.
        self.dtype = dispatch_dtype if ops.supported_dtype(dispatch_dtype) else torch.float32

    def update(self, indices_, locations_, gates_, capacity=None, is_postscore=True, plan=None):
        if plan is not None and plan.idx2d is not None:
            self.idx2d, self.loc2d = plan.idx2d, plan.loc2d
            self.slot_map = plan.slot_map
        else:
            self.idx2d, self.loc2d = _pack(indices_, torch.int32), _pack(locations_, torch.int32)
            self.slot_map = None
        gdt = gates_[0].dtype if ops.supported_dtype(gates_[0].dtype) else torch.float32
        if plan is not None and plan.gates2d is not None and plan.gates2d.dtype == gdt and not any(g.requires_grad for g in gates_):
            self.gates2d = plan.gates2d
        else:
            self.gates2d = _pack(gates_, gdt)
        self.is_postscore = is_postscore
        self.sample_size = int(self.idx2d.shape[1])
        self.capacity = int(capacity) or self.capacity
        if self.slot_map is None or self.slot_map.numel() != self.E * self.capacity:
            self.slot_map = ops.slot_map(self.idx2d, self.loc2d, self.E, self.capacity)

    def _run(self, fn, data, with_gates):
        x = data if data.dtype == self.dtype else data.to(self.dtype)
        x = x if x.is_contiguous() else x.contiguous()
        gates = self.gates2d if with_gates else None
        if torch.is_grad_enabled() and (x.requires_grad or (gates is not None and gates.requires_grad)):
            out = fn.apply(self, x, gates)
        elif fn is _Encode:   # inference: straight to the kernel, no autograd node
            out = ops.fast_encode(x, self.slot_map, gates, self.E * self.capacity)
        else:
            out = ops.fast_decode(x, self.idx2d, self.loc2d, gates, self.capacity)
        return out if out.dtype == self.original_dtype else out.to(self.original_dtype)

    def encode(self, data):
        return self._run(_Encode, data, with_gates=not self.is_postscore)

    def decode(self, data):
        return self._run(_Decode, data.reshape(-1, data.size(-1)), with_gates=self.is_postscore)


fast_dispatcher = TutelMoeFastDispatcher


_aten_warned = set()


def _routing_limits_exceeded(E, k):
    """outside the routing kernels' shapes (csrc/routing.hip: per-tile expert histograms live in LDS)"""
    return E > 4096 or k > 16 or k * E > 8192


def _locations_aten(idx2d, E):
    """the reference's own op chain for the slots (fast_dispatch.py:150,159-171,177-178) -- one-hot masks, k column cumsums
    (tutel_amd_cumsum_sub_one takes any E), masked row sums -- for expert counts the routing kernels do not take"""
    k, T = idx2d.shape
    acc = torch.zeros([E], dtype=torch.int32, device=idx2d.device)
    locs = []
    for j in range(k):
        valid = idx2d[j] >= 0
        mask = torch.zeros([T, E], dtype=torch.int32, device=idx2d.device)
        mask.scatter_(1, idx2d[j].clamp(min=0).long().unsqueeze(-1), valid.to(torch.int32).unsqueeze(-1))
        loc = ops.cumsum_sub_one(mask) + acc.unsqueeze(0)
        locs.append((loc * mask).sum(dim=1).to(torch.int32))
        acc = acc + mask.sum(dim=0, dtype=torch.int32)
    stats = acc.max().reshape(1) if E > 0 else torch.zeros([1], dtype=torch.int32, device=idx2d.device)
    return torch.stack(locs).contiguous(), acc, stats.to(torch.int32)


def _topk_aten(work, k, apply_softmax, normalize_gate):
    """torch.topk + gather + normalise, op for op as fast_dispatch.py:146-151,173-175 (tie order: torch.topk's, like upstream)"""
    sc = torch.softmax(work, dim=1) if apply_softmax else work
    top = torch.topk(sc, k, dim=1).indices
    gl = [sc.gather(1, top[:, j:j + 1]).squeeze(-1) for j in range(k)]
    if k > 1 and normalize_gate:
        denom = torch.clamp(sum(gl), min=torch.finfo(gl[0].dtype).eps)
        gl = [g / denom for g in gl]
    return top.t().contiguous().to(torch.int32), torch.stack(gl).contiguous(), sc


def extract_critical(scores, top_k, loss_fn=losses.gshard_loss, capacity_factor=1.0,
                     batch_prioritized_routing=False, normalize_gate=True, alignment=1, group=None,
                     inequivalent_tokens=False, _logits=None):
    """Top-k routing (reference fast_dispatch.py:143-204).  scores [T,E] -> (crit, l_aux).

    `_logits` (internal): raw gate logits; when given, softmax is fused into the top-k kernel and
    `scores` may be None."""
    src = _logits if _logits is not None else scores
    T, E = int(src.size(0)), int(src.size(1))
    k_req, k = top_k, min(top_k, E)
    work = src if ops.routing_dtype(src.dtype) else src.float()
    needs_grad = torch.is_grad_enabled() and src.requires_grad

    fused_loss = loss_fn is losses.gshard_loss and not needs_grad

    # samples per expert from the GLOBAL maximum token count when ranks hold different numbers of tokens
    # (fast_dispatch.py:181-186): every rank must derive the same capacity in every branch below
    n = T
    if inequivalent_tokens:
        n = int(simple_all_reduce(torch.tensor(T, device=src.device), group=group, op=torch.distributed.ReduceOp.MAX))
    spe = (n + E - 1) // E
    if capacity_factor > 0:
        capacity = k * int(capacity_factor * spe)
        rem = capacity % alignment
        capacity += (alignment - rem) if rem > 0 else 0
    else:
        capacity = 0  # known only after the counts are

    # Shapes past the routing kernels' limits (E > 4096, k > 16, k * E > 8192): upstream's ATen chain takes any E and k
    # (fast_dispatch.py:145-148), so a drop-in must not raise there -- the same op chain runs on the device instead, loudly (once
    # per shape).  The C ABI itself still refuses such shapes with an error naming the limit.
    aten = _routing_limits_exceeded(E, k)
    if aten and (E, k) not in _aten_warned:
        _aten_warned.add((E, k))
        logging.warning("tutel_amd: routing with E = %d, k = %d is outside the HIP routing kernels' limits (E <= 4096, k <= 16, k * E <= 8192): "
                        "running the reference's ATen op chain on the device instead", E, k)

    # the top-k launch also clears the bucket->token map the location launch fills
    pre = None
    if aten:
        idx2d, gates2d, scores_k = _topk_aten(work.detach(), k, _logits is not None, normalize_gate)
        ws = None
        fused_loss = False
    else:
        if capacity > 0 and not batch_prioritized_routing:
            pre = torch.empty([E * capacity], dtype=torch.int32, device=src.device)
        idx2d, gates2d, ws, scores_k = ops.gate_topk(work.detach(), k, apply_softmax=_logits is not None,
                                                     normalize_gate=normalize_gate,
                                                     want_scores=(_logits is not None and (needs_grad or loss_fn not in (None, losses.gshard_loss))),
                                                     clear=pre)

    if aten:
        if batch_prioritized_routing:
            order = torch.argsort(-scores_k.max(dim=1)[0], stable=True)
            inv = torch.empty_like(order)
            inv[order] = torch.arange(T, device=order.device)
            loc_s, cnt, stats = _locations_aten(idx2d[:, order].contiguous(), E)
            loc2d = loc_s[:, inv].contiguous()
        else:
            loc2d, cnt, stats = _locations_aten(idx2d, E)
        l_aux_k, smap = None, None
    elif batch_prioritized_routing:
        # tokens ranked by -max score get their buckets first (fast_dispatch.py:138-141,155-157):
        # run the same stable rank on the importance-sorted order, then undo the permutation.
        sc = scores_k if _logits is not None and scores_k is not None else (work if _logits is None else torch.softmax(work.float(), 1))
        order = torch.argsort(-sc.max(dim=1)[0], stable=True)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(T, device=order.device)
        loc_s, cnt, stats, l_aux_k, _ = ops.compute_location(idx2d[:, order].contiguous(), E, ws=None, capacity=0)
        loc2d = loc_s[:, inv].contiguous()
        smap = None
        fused_loss = False
    else:
        loc2d, cnt, stats, l_aux_k, smap = ops.compute_location(
            idx2d, E, ws=ws, capacity=capacity, want_l_aux=fused_loss,
            l_aux_dtype=src.dtype if ops.supported_dtype(src.dtype) else torch.float32, cleared_slot_map=pre)

    if capacity_factor <= 0:
        cap = stats[0]
        capacity = int(simple_all_reduce(cap, group=group, op=torch.distributed.ReduceOp.MAX))  # the one host sync the API implies
        if capacity_factor < 0:
            capacity = min(capacity, k * int(-capacity_factor * spe))
        rem = capacity % alignment
        capacity += (alignment - rem) if rem > 0 else 0
        smap = None
    if smap is None and capacity > 0:
        smap = ops.slot_map(idx2d, loc2d, E, capacity)
    elif capacity == 0:
        smap = torch.empty([0], dtype=torch.int32, device=src.device)

    # gates / loss: kernel values on the inference path; differentiable torch forms when training
    gate_list = None
    if needs_grad or not ops.routing_dtype(src.dtype):
        # training (autograd through the gates), or an exotic scores dtype routed on its fp32 image
        # (then the gate VALUES are still computed in the scores' own precision, as the reference
        # does before its fp32 dispatch cast, fast_dispatch.py:151,173-175,105)
        sc = scores if scores is not None else torch.softmax(_logits, dim=1)
        if not needs_grad:
            sc = sc.detach()
        gate_list = [sc.gather(1, idx2d[j].long().unsqueeze(-1)).squeeze(-1) for j in range(k)]
        if k > 1 and normalize_gate:
            denom = torch.clamp(sum(gate_list), min=torch.finfo(gate_list[0].dtype).eps)
            gate_list = [g / denom for g in gate_list]
    elif gates2d.dtype != src.dtype:
        gates2d = gates2d.to(src.dtype)

    if loss_fn is None:
        l_aux = None
    elif fused_loss:
        l_aux = l_aux_k[0] if l_aux_k.dtype == src.dtype else l_aux_k[0].to(src.dtype)
    else:
        sc = scores if scores is not None else (scores_k if scores_k is not None and not needs_grad else torch.softmax(_logits, dim=1))
        l_aux = loss_fn(sc, idx2d.t().long())

    if get_world_rank(group) == 0 and logging.getLogger().isEnabledFor(logging.INFO):
        logging.info("Capacity = %d, real-time capacity-factor for top-%d = %s", capacity, k_req, capacity / max(1, k * spe))

    return RoutingPlan(E, idx2d, loc2d, gates2d if gate_list is None else None, capacity, cnt, smap, gate_list), l_aux


def get_dispatch_count(critial_data):
    return critial_data[-1]


def _dispatcher_for(data, crit, is_postscore):
    d = TutelMoeFastDispatcher(crit[0], 0, data.size(-1), data.dtype)
    d.update(*crit[1:-1], is_postscore=is_postscore, plan=crit if isinstance(crit, RoutingPlan) else None)
    return d


def fast_encode(data, critial_data, is_postscore=True):
    assert data.is_contiguous(), "Input tensor for encode/decode should be in contiguous memory format."
    E = critial_data[0]
    return _dispatcher_for(data, critial_data, is_postscore).encode(data).view(E, -1, data.size(-1))


def fast_decode(data, critial_data, is_postscore=True):
    assert data.is_contiguous(), "Input tensor for encode/decode should be in contiguous memory format."
    return _dispatcher_for(data, critial_data, is_postscore).decode(data).view(-1, data.size(-1))
