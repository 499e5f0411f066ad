"""TEST-ONLY stand-in for tutel_amd.ops on CPU tensors, backed by the oracle.

The product has no CPU path (tutel_amd.ops raises on CPU tensors).  The host-side logic around
the kernels -- capacity / alignment arithmetic, the expert-parallel exchange and its layouts,
overlap chunking, dtype policy, autograd wiring -- still has to be testable without a GPU and
with gloo at world_size 2, so these tests monkeypatch the `ops` functions with the oracle."""
import torch

from oracle import moe_oracle as O


def gate_topk(inp, k, apply_softmax=False, normalize_gate=True, want_scores=False, ws=None, clear=None):
    scores = torch.softmax(inp, dim=1) if apply_softmax else inp
    k = min(k, scores.shape[1])
    idx = O.topk_indices(scores, k)
    gates = [scores.gather(1, i.long().unsqueeze(-1)).squeeze(-1) for i in idx]
    if k > 1 and normalize_gate:
        denom = torch.clamp(sum(gates), min=torch.finfo(gates[0].dtype).eps)
        gates = [g / denom for g in gates]
    return torch.stack(idx), torch.stack(gates), {"scores": scores, "idx0": idx[0]}, scores


def _smap(idx, loc, E, C):
    k, T = idx.shape
    sm = torch.full([E * C], -1, dtype=torch.int32)
    for j in range(k):
        keep = (loc[j] < C) & (idx[j] >= 0) & (idx[j] < E)
        t = torch.arange(T)[keep]
        sm[idx[j][keep].long() * C + loc[j][keep].long()] = (j * T + t).int()
    return sm


def compute_location(idx, E, ws=None, capacity=0, want_l_aux=False, l_aux_dtype=torch.float32, cleared_slot_map=None):
    loc, cnt = O.compute_locations([idx[j] for j in range(idx.shape[0])], E)
    loc = torch.stack(loc)
    stats = cnt.max().reshape(1)
    l_aux = None
    if want_l_aux and ws is not None:
        if idx.shape[1] == 0:   # a rank without tokens: the HIP path writes 0 (the reference's loss divides by the token count)
            l_aux = torch.zeros([1], dtype=l_aux_dtype)
        else:
            l_aux = O.gshard_loss(ws["scores"], ws["idx0"]).to(l_aux_dtype).reshape(1)
    smap = _smap(idx, loc, E, capacity) if capacity > 0 else None
    return loc, cnt, stats, l_aux, smap


def slot_map(idx, loc, E, capacity):
    return _smap(idx, loc, E, capacity)


def cumsum_sub_one(mask):
    return O.cumsum_sub_one(mask)


def _crit(idx, loc, gates, C, E):
    k = idx.shape[0]
    g = [gates[j] for j in range(k)] if gates is not None else [torch.ones(idx.shape[1]) for _ in range(k)]
    return (E, [idx[j] for j in range(k)], [loc[j] for j in range(k)], g, C, None)


def fast_encode(x, smap, gates, n_slots):
    T, M = x.shape
    out = torch.zeros([n_slots, M], dtype=torch.float32)
    used = smap >= 0
    q = smap[used].long()
    rows = x.float()[q % T]
    if gates is not None:
        rows = rows * gates.reshape(-1).float()[q].unsqueeze(1)
    out[used] = rows
    return out.to(x.dtype)


def fast_decode(buf, idx, loc, gates, capacity, num_experts=0, chunk_rows=0):
    E = buf.shape[0] // max(capacity, 1)
    if chunk_rows > 0:  # chunk-major [C/c, E, c, M] -> plain [E, C, M]
        buf = buf.view(capacity // chunk_rows, E, chunk_rows, -1).permute(1, 0, 2, 3).reshape(E * capacity, -1)
    crit = _crit(idx, loc, gates, capacity, E)
    return O.fast_decode(buf.view(E, capacity, -1), crit, is_postscore=True)


def gate_grad(x, buf, idx, loc, capacity):
    return torch.stack([O.gate_grad(x, buf, idx[j], loc[j], capacity) for j in range(idx.shape[0])])


def install(monkeypatch):
    from tutel_amd import ops
    for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode",
                 "fast_decode", "gate_grad"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(ops, "supported_dtype", lambda dt: dt in (torch.float32, torch.float16, torch.bfloat16))
    monkeypatch.setattr(ops, "routing_dtype", lambda dt: dt in (torch.float32, torch.float16, torch.bfloat16, torch.float64))
