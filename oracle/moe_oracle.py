"""
oracle/moe_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's MoE *forward* hot path (microsoft/tutel @ 2025-02-04):
gate -> softmax -> top-k routing (compute_location) -> fast_encode -> all-to-all layout ->
per-expert FFN -> all-to-all layout -> fast_decode.  Every function cites the reference lines
it follows (paths relative to /root/reference).

Who may use this file: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg --
as the checker only.  Nothing under tutel_amd/ imports it; the product fails loudly when its
HIP library is missing instead of falling back to this code.

How it is pinned (so that "parity" means parity with the reference, not with ourselves):
  * tests/test_oracle_vs_reference.py runs the reference itself in this container (python
    package from /root/reference + its C++ CPU kernels compiled into oracle/_ref/ by
    oracle/Makefile) and checks every function below against it on seeded inputs;
  * tests/golden/*.npz are outputs of that reference (generator tests/golden/make_golden.py)
    and are replayed against this oracle on any box (tests/test_oracle_golden.py);
  * the reference's own golden-loss file tests/test_baseline.json (head committed as
    tests/golden/reference_baseline_losses.json): losses[0] of its 9 entries, a forward-only quantity,
    is reproduced from helloworld's seeds by helloworld_problem() + moe_forward() below.

Arithmetic split: integer/index work and the scatter/gather loops are plain C
(oracle/moe_oracle.c via ctypes); softmax / matmul / dtype rounding use torch-CPU ATen ops,
which is precisely what the reference calls on its own CPU path (torch 2.10.0 in this image).
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmoe_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libmoe_oracle.so"])
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------
# routing
# ---------------------------------------------------------------------------------------------
TIE_RULE = "aten"   # "aten": the reference's CPU torch.topk, ties included (aten_topk.c); "lowest": lowest expert index first (the
                    # product's TUTEL_OPT_TIE_RULE = 0; rounds 1-5)


def topk_indices(scores, k, tie_rule=None):
    """k index vectors [T] int32 -- torch.topk(scores, k, dim=1).indices of the reference's CPU path (fast_dispatch.py:146-148),
    INCLUDING its order among exactly equal scores: ATen's CPU kernel (nth_element / partial_sort over (value, index) pairs with a
    value-only comparator) restated in oracle/aten_topk.c and pinned against live torch.topk on tie-heavy rows
    (tests/test_oracle_vs_reference.py).  tie_rule="lowest": descending score, ties -> lowest expert index (tie-free rows: the same)."""
    T, E = scores.shape
    k = min(k, E)
    rule = tie_rule or TIE_RULE
    assert rule in ("aten", "lowest")
    idx = torch.empty([k, T], dtype=torch.int32)
    L = _lib()
    if scores.dtype == torch.float64:
        s = scores.contiguous()
        fn = L.orc_aten_topk_f64 if rule == "aten" else L.orc_topk_f64
    else:
        s = scores.float().contiguous()  # bf16/fp16 -> fp32 is exact, ordering (and NaN-ness) preserved
        fn = L.orc_aten_topk_f32 if rule == "aten" else L.orc_topk_f32
    rc = fn(_p(s), T, E, k, _p(idx))
    assert rule == "lowest" or rc == 0
    return [idx[j].clone() for j in range(k)]


def cumsum_sub_one(mask):
    """Reference: jit_kernels/gating.py:13-15,19-24 (torch.cumsum(mask,0)-1)."""
    m = mask.to(torch.int32).contiguous()
    out = torch.empty_like(m)
    _lib().orc_cumsum_sub_one_i32(_p(m), m.shape[0], m.shape[1], _p(out))
    return out


def compute_locations(idx_list, E):
    """(loc_list, dispatch_count). Reference: fast_dispatch.py:150,159-171,177-178."""
    k, T = len(idx_list), idx_list[0].numel()
    idx = torch.stack([x.to(torch.int32).view(-1) for x in idx_list]).contiguous()
    loc = torch.empty_like(idx)
    cnt = torch.empty([E], dtype=torch.int32)
    _lib().orc_compute_locations(_p(idx), T, E, k, _p(loc), _p(cnt))
    return [loc[j].clone() for j in range(k)], cnt


def gshard_loss(scores, idx0):
    """Reference: losses.py:12-19 (only the FIRST choice enters the loss)."""
    T, E = scores.shape
    mask = torch.zeros([T, E], dtype=scores.dtype)
    mask.scatter_(1, idx0.long().unsqueeze(-1), E / T)
    me = torch.sum(scores, dim=0)
    ce = torch.sum(mask, dim=0)
    return torch.sum(me * ce) / T


def load_importance_loss(scores_wo_noise, topk_logits, E, gate_noise):
    """Reference: losses.py:21-42.  l = (cv2(importance) + cv2(load)) / 2 with cv2(v) = var(v) / (mean(v)^2 + 1e-10)
    (unbiased variance), importance[e] = sum_t scores[t,e], load[e] = sum_t Phi((scores[t,e] - thr[t]) / (gate_noise / E)),
    thr[t] = the k-th (last) of the token's top-k NOISY LOGITS.  (The reference really does subtract a logit from a
    probability -- restated as written.)  All in fp32; Phi through erf, as torch's Normal.cdf evaluates it."""
    assert gate_noise > 0
    s = scores_wo_noise.float()

    def cv2(v):
        return v.var() / (v.mean() ** 2 + 1e-10)
    thr = topk_logits[:, -1].reshape(-1, 1).float()
    sigma = torch.tensor([gate_noise / E], dtype=torch.float32)
    load = (0.5 * (1 + torch.erf((s - thr) * sigma.reciprocal() / math.sqrt(2)))).sum(0)
    return (cv2(s.sum(0)) + cv2(load)) / 2.0


def capacity_of(T, E, k, capacity_factor, dispatch_count, alignment=1):
    """Reference: fast_dispatch.py:188-199 (single rank: the all-reduce MAX is the identity)."""
    spe = (T + E - 1) // E
    if capacity_factor > 0:
        cap = k * int(capacity_factor * spe)
    else:
        cap = int(dispatch_count.max())
        if capacity_factor < 0:
            cap = min(cap, k * int(-capacity_factor * spe))
    rem = cap % alignment
    if rem > 0:
        cap += alignment - rem
    return cap


def extract_critical(scores, top_k, capacity_factor=1.0, normalize_gate=True, alignment=1,
                     with_loss=True, topk_override=None, num_samples=None):
    """Reference: fast_dispatch.py:143-204 (batch_prioritized_routing=False).
    Returns ((E, idx_list, loc_list, gate_list, capacity, dispatch_count), l_aux).
    `topk_override`: inject the reference's own topk indices (used to compare everything
    downstream of a tie independently of the tie rule).  `num_samples`: the all-reduced MAX token count
    of `inequivalent_tokens=True` (fast_dispatch.py:181-186) -- the capacity follows it, not the local T."""
    T, E = scores.shape
    k = min(top_k, E)
    idx_list = topk_override if topk_override is not None else topk_indices(scores, k)
    # gates_s[k] = (scores * one_hot).sum(1) == scores[t, idx_k[t]] exactly (:150-151)
    gates = [scores.gather(1, i.long().unsqueeze(-1)).squeeze(-1) for i in idx_list]
    l_aux = gshard_loss(scores, idx_list[0]) if with_loss else None
    loc_list, cnt = compute_locations(idx_list, E)
    if k > 1 and normalize_gate:  # :173-175 -- python sum(): ((0 + g0) + g1) ... in scores dtype
        denom = torch.clamp(sum(gates), min=torch.finfo(gates[0].dtype).eps)
        gates = [g / denom for g in gates]
    cap = capacity_of(T if num_samples is None else num_samples, E, k, capacity_factor, cnt, alignment)
    return (E, idx_list, loc_list, gates, cap, cnt), l_aux


# ---------------------------------------------------------------------------------------------
# dispatch / combine
# ---------------------------------------------------------------------------------------------
def _dispatch_dtype(dtype):
    # fast_dispatch.py:94-96: fp32 unless (fp16 on a CUDA build); this image's torch is a HIP
    # build, where the reference always dispatches in fp32 -- fp64 inputs included.
    return torch.float32


def fast_encode(x, crit, is_postscore=True):
    """[T,M] -> [E,C,M].  Reference: fast_dispatch.py:209-214,101-128,18-29 and the CPU kernel
    custom_kernel.cpp:293-300."""
    E, idx_list, loc_list, gates, C, _ = crit
    T, M = x.shape
    dt = _dispatch_dtype(x.dtype)
    xin = x.to(dt).contiguous()
    out = torch.zeros([E * C, M], dtype=dt)
    for j in range(len(idx_list)):
        g = torch.ones([T], dtype=dt) if is_postscore else gates[j].to(dt).contiguous()
        i32 = idx_list[j].to(torch.int32).contiguous()
        l32 = loc_list[j].to(torch.int32).contiguous()
        _lib().orc_encode_f32(_p(g), _p(i32), _p(l32), _p(xin), _p(out), T, M, C)
    return out.to(x.dtype).view(E, C, M)


def fast_decode(y, crit, is_postscore=True):
    """[E,C,M] -> [T,M].  Reference: fast_dispatch.py:216-221,130-134,52-66 and the CPU kernel
    custom_kernel.cpp:301-312; the k temps are summed left to right in fp32, one final cast."""
    E, idx_list, loc_list, gates, C, _ = crit
    M = y.shape[-1]
    T = idx_list[0].numel()
    dt = _dispatch_dtype(y.dtype)
    buf = y.reshape(E * C, M).to(dt).contiguous()
    acc = None
    for j in range(len(idx_list)):
        g = gates[j].to(dt).contiguous() if is_postscore else torch.ones([T], dtype=dt)
        i32 = idx_list[j].to(torch.int32).contiguous()
        l32 = loc_list[j].to(torch.int32).contiguous()
        tmp = torch.empty([T, M], dtype=dt)
        _lib().orc_decode_f32(_p(g), _p(i32), _p(l32), _p(tmp), _p(buf), T, M, C)
        acc = tmp if acc is None else acc + tmp
    return acc.to(y.dtype)


def gate_grad(x, buf, idx, loc, C):
    """Reference: custom_kernel.cpp:313-322 (backward only)."""
    T, M = x.shape
    xin, b = x.float().contiguous(), buf.reshape(-1, M).float().contiguous()
    out = torch.empty([T], dtype=torch.float32)
    _lib().orc_gate_grad_f32(_p(out), _p(idx.to(torch.int32).contiguous()),
                             _p(loc.to(torch.int32).contiguous()), _p(xin), _p(b), T, M, C)
    return out


# ---------------------------------------------------------------------------------------------
# experts / gate
# ---------------------------------------------------------------------------------------------
def expert_ffn(x, w1, b1, w2, b2, act=torch.relu, accum_fp32=False):
    """x [E_loc,R,M]; w1 [E_loc,H,M]; w2 [E_loc,H,M_out]; b1 [E_loc,H]; b2 [E_loc,M_out].
    Reference: experts/ffn.py:114-120.  accum_fp32=False reproduces the reference op by op in
    the tensors' own dtype (ATen bmm rounds after each op); accum_fp32=True computes in fp32 on
    the same (already rounded) inputs and rounds the hidden activation and the output once --
    the tolerance anchor for the fused bf16/fp16 MFMA kernel (SURVEY section 7 hard part 7)."""
    if not accum_fp32:
        y = torch.matmul(x, w1.permute(0, 2, 1))
        if b1 is not None:
            y = torch.add(y, b1.unsqueeze(1))
        y = act(y)
        y = torch.matmul(y, w2)
        if b2 is not None:
            y = torch.add(y, b2.unsqueeze(1))
        return y
    dt = x.dtype
    h = torch.matmul(x.float(), w1.float().permute(0, 2, 1))
    if b1 is not None:
        h = h + b1.float().unsqueeze(1)
    h = act(h).to(dt).float()
    y = torch.matmul(h, w2.float())
    if b2 is not None:
        y = y + b2.float().unsqueeze(1)
    return y.to(dt)


def expert_llama_ffn(x, w1, w2, w3, act=torch.nn.functional.silu, accum_fp32=False):
    """SwiGLU expert: x [E_loc,R,M]; w1, w2 [E_loc,M,H]; w3 [E_loc,H,M] -> [E_loc,R,M].
    Reference: experts/llama_ffn.py:33-42.  accum_fp32 as in expert_ffn: fp32 accumulation on the
    same inputs, act(y1) rounded once (it is stored), the gated product rounded once, output once."""
    if not accum_fp32:
        y1 = torch.matmul(x, w1)
        y2 = torch.matmul(x, w2)
        return torch.matmul(act(y1) * y2, w3)
    dt = x.dtype
    g = act(torch.matmul(x.float(), w1.float())).to(dt).float()
    h = (g * torch.matmul(x.float(), w2.float())).to(dt).float()
    return torch.matmul(h, w3.float()).to(dt)


def cosine_gate_logits(x, proj_w, proj_b, sim, temperature, fp32_gate=False):
    """Cosine top-k gate logits.  Reference: gates/cosine_top.py:22-34 (clamp_max = log(1/0.01)
    evaluated in fp32, :15).  x [T,M]; proj_w [P,M]; proj_b [P]; sim [P,E]; temperature [1]."""
    if fp32_gate:
        x, proj_w, proj_b, sim = x.float(), proj_w.float(), proj_b.float(), sim.float()
    else:
        x = x.to(proj_w.dtype)
    F = torch.nn.functional
    logits = torch.matmul(F.normalize(F.linear(x, proj_w, proj_b), dim=1), F.normalize(sim, dim=0))
    clamp_max = torch.log(torch.tensor(1. / 0.01)).item()
    return logits * torch.clamp(temperature, max=clamp_max).exp()


def gate_scores(x, wg, fp32_gate=False):
    """Reference: gates/top.py:20-22 and moe_layer.py:290.  Returns (scores, logits_dtype)."""
    w = wg.float() if fp32_gate else wg
    logits = torch.nn.functional.linear(x.to(w.dtype), w)
    return torch.softmax(logits, dim=1), logits.dtype


# ---------------------------------------------------------------------------------------------
# all-to-all layout (W ranks simulated in one process)
# ---------------------------------------------------------------------------------------------
def a2a_dispatch(per_rank):
    """list of W tensors [E,C,M] (one per source rank) -> list of W tensors [E_loc, W*C, M].
    Reference: communicate.py:181-192,447-503 (all_to_all(y,1,0)) + :606-613."""
    W = len(per_rank)
    E, C, M = per_rank[0].shape
    E_loc = E // W
    send = torch.stack([t.contiguous() for t in per_rank]).contiguous()
    recv = torch.empty_like(send)
    _lib().orc_a2a_dispatch_layout(_p(send), _p(recv), W, E_loc, C, M, send.element_size())
    return [recv[d].view(E_loc, W * C, M) for d in range(W)]


def a2a_combine(per_rank, C):
    """inverse of a2a_dispatch: list of W [E_loc, W*C, M] -> list of W [E, C, M].
    Reference: communicate.py:447-503 (all_to_all(y,0,1)) + :615-622."""
    W = len(per_rank)
    E_loc, R, M = per_rank[0].shape
    out = [torch.empty([W * E_loc, C, M], dtype=per_rank[0].dtype) for _ in range(W)]
    for d in range(W):
        v = per_rank[d].view(E_loc, W, C, M)
        for s in range(W):
            out[s].view(W, E_loc, C, M)[d] = v[:, s]
    return out


# ---------------------------------------------------------------------------------------------
# the layer
# ---------------------------------------------------------------------------------------------
def moe_forward(x, wg, w1, b1, w2, b2, top_k=2, capacity_factor=1.0, fp32_gate=False,
                normalize_gate=True, is_postscore=True, act=torch.relu, alignment=1,
                accum_fp32=False, topk_override=None, logits_fn=None, expert_fn=None, noise=None,
                gate_noise=0.0, is_gshard_loss=True, encode_fn=None, decode_fn=None):
    """Single-rank MOELayer.forward.  Reference: moe_layer.py:255-363 (dtype chain :264-270,
    :327, :359-361).  x [..., M] -> (y [..., M_out], l_aux, crit, stages dict).
    logits_fn(x[T,M]) -> logits replaces the linear gate (custom / cosine gates, moe_layer.py:283);
    expert_fn(enc[E,C,M]) -> [E,C,M_out] replaces the ReLU FFN (custom / llama experts, :251).
    noise [T,E]: the randn_like draw of a TRAINING forward with gate_noise > 0 (:285-288; routed on
    logits + gate_noise * noise / E); is_gshard_loss=False: the load-importance loss (:291-296)."""
    orig_shape, orig_dtype = x.shape, x.dtype
    M = orig_shape[-1]
    xr = x.reshape(-1, M).to(w1.dtype)
    if logits_fn is not None:
        logits = logits_fn(xr)
    else:
        w = wg.float() if fp32_gate else wg
        logits = torch.nn.functional.linear(xr.to(w.dtype), w)   # gates/top.py:20-22
    logits_dtype = logits.dtype
    noisy = logits if noise is None else logits + gate_noise * noise.to(logits.dtype) / logits.shape[1]
    scores = torch.softmax(noisy, dim=1)
    crit, l_aux = extract_critical(scores, top_k, capacity_factor, normalize_gate, alignment,
                                   topk_override=topk_override)
    if not is_gshard_loss:
        ids = torch.stack(crit[1], dim=1).long()
        l_aux = load_importance_loss(torch.softmax(logits, dim=1), noisy.gather(1, ids), logits.shape[1], gate_noise)
    enc = (encode_fn or fast_encode)(xr.to(logits_dtype), crit, is_postscore).to(xr.dtype)
    if expert_fn is not None:
        ffn = expert_fn(enc)
    else:
        ffn = expert_ffn(enc, w1, b1, w2, b2, act, accum_fp32=accum_fp32)
    dec = (decode_fn or fast_decode)(ffn.to(logits_dtype), crit, is_postscore)
    y = dec.view(list(orig_shape[:-1]) + [ffn.shape[-1]]).to(orig_dtype)
    return y, l_aux, crit, {"scores": scores, "encoded": enc, "expert_out": ffn}


def moe_forward_ep(xs, wg, w1s, b1s, w2s, b2s, top_k=2, capacity_factor=1.0, fp32_gate=False,
                   normalize_gate=True, is_postscore=True, act=torch.relu, alignment=1,
                   accum_fp32=False, inequivalent_tokens=False, return_expert_inputs=False):
    """Expert-parallel forward with W ranks simulated in-process: xs[r] is rank r's [T,M]
    tokens, w1s[r] etc. rank r's local expert weights.  Reference: moe_layer.py:344-351 with
    num_local_experts > 0 (E = E_loc*W, moe_layer.py:46-55).  inequivalent_tokens: the ranks hold
    different numbers of tokens and size their buckets from the largest (fast_dispatch.py:181-186)."""
    W = len(xs)
    crits, encs, ldt = [], [], None
    n_max = max(int(x.shape[0]) for x in xs) if inequivalent_tokens else None
    for r in range(W):
        xr = xs[r].to(w1s[r].dtype)
        scores, ldt = gate_scores(xr, wg, fp32_gate)
        crit, _ = extract_critical(scores, top_k, capacity_factor, normalize_gate, alignment, with_loss=xr.shape[0] > 0,
                                   num_samples=n_max)
        crits.append(crit)
    if capacity_factor <= 0:
        # dropless: the capacity is the all-reduce MAX over the ranks of the largest expert load (fast_dispatch.py:191-193),
        # then clamped / aligned like the single-rank value (:194-199)
        E, k = crits[0][0], len(crits[0][1])
        cap = max(int(c[5].max()) if c[5].numel() else 0 for c in crits)
        n = n_max if n_max is not None else int(xs[0].shape[0])
        if capacity_factor < 0:
            cap = min(cap, k * int(-capacity_factor * ((n + E - 1) // E)))
        if cap % alignment:
            cap += alignment - cap % alignment
        crits = [(c[0], c[1], c[2], c[3], cap, c[5]) for c in crits]
    for r in range(W):
        xr = xs[r].to(w1s[r].dtype)
        encs.append(fast_encode(xr.to(ldt), crits[r], is_postscore).to(xr.dtype))
    C = crits[0][4]
    assert all(c[4] == C for c in crits)
    recv = a2a_dispatch(encs)
    outs = [expert_ffn(recv[r], w1s[r], b1s[r], w2s[r], b2s[r], act, accum_fp32=accum_fp32)
            for r in range(W)]
    back = a2a_combine(outs, C)
    ys = [fast_decode(back[r].to(ldt), crits[r], is_postscore).to(xs[r].dtype) for r in range(W)]
    if return_expert_inputs:   # [E_loc, W*C, M] per rank: what the reference's experts see after all_to_all(y, 1, 0)
        return ys, crits, recv
    return ys, crits


# ---------------------------------------------------------------------------------------------
# deterministic synthetic problem (mirrors helloworld.py:76-85,112-113 and ffn.py:39-49)
# ---------------------------------------------------------------------------------------------
def make_problem(T, M, H, E, dtype=torch.float32, seed=0, out_dim=None):
    """nn.Linear-style init like FusedExpertsNetwork.reset_parameters (ffn.py:39-49) but from
    one explicit generator so the GPU box can regenerate it bit-identically from the seed."""
    g = torch.Generator().manual_seed(seed)
    out_dim = out_dim or M

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    x = torch.randn([T, M], generator=g)
    wg = uni([E, M], 1 / math.sqrt(M))
    w1 = uni([E, H, M], 1 / math.sqrt(M))
    b1 = uni([E, H], 1 / math.sqrt(M))
    w2 = uni([E, H, out_dim], 1 / math.sqrt(H))
    b2 = uni([E, out_dim], 1 / math.sqrt(H))
    return [t.to(dtype) for t in (x, wg, w1, b1, w2, b2)]


def helloworld_problem(batch, tokens, M, H, E_loc, dtype=torch.float32, rank=0, world=1):
    """The inputs tutel/examples/helloworld.py builds for rank `rank`: x [batch, tokens, M], wg [E, M], w1 [E_loc, H, M],
    b1 [E_loc, H], w2 [E_loc, H, M], b2 [E_loc, M] -- the seeds the script passes (`seeds=(1, rank + 1, 1)`, helloworld.py:81),
    consumed as MOELayer.__init__ does (expert weights under seeds[1], moe_layer.py:157-158; gate under seeds[0], :211-212),
    the per-expert nn.Linear draws of FusedExpertsNetwork.reset_parameters (ffn.py:39-49), the default dtype the script
    sets (helloworld.py:60-67) and its tokens (manual_seed(0), fp32 randn on the CPU cast to dtype, :112-113).
    This is what makes the losses of the reference's own tests/test_baseline.json (test_tutel.py:94-152) reproducible."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(rank + 1)
        w1, b1, w2, b2 = [], [], [], []
        for _ in range(E_loc):
            fc1, fc2 = torch.nn.Linear(M, H), torch.nn.Linear(H, M)
            w1.append(fc1.weight.detach())
            b1.append(fc1.bias.detach())
            w2.append(fc2.weight.detach().t())
            b2.append(fc2.bias.detach())
        torch.manual_seed(1)
        wg = torch.nn.Linear(M, E_loc * world, bias=False).weight.detach()
        torch.manual_seed(0)
        x = torch.randn([batch, tokens, M], dtype=torch.float32).to(dtype)
    finally:
        torch.set_default_dtype(old)
    return x, wg, torch.stack(w1), torch.stack(b1), torch.stack(w2), torch.stack(b2)


def helloworld_loss(y):
    """helloworld.py:97,114,132: nll_loss(log_softmax(sum over the model dim), target 0)"""
    out = torch.log_softmax(torch.sum(y, dim=2), dim=1)
    return torch.nn.functional.nll_loss(out, torch.zeros(y.shape[0], dtype=torch.long))


def make_problem_ext(T, M, H, E, P=32, dtype=torch.float32, seed=0):
    """Inputs for the cosine gate + SwiGLU expert case (gates/cosine_top.py:12-16 and
    experts/llama_ffn.py:27-31 initialisers, from one explicit generator):
    x [T,M], proj_w [P,M], proj_b [P], sim [P,E], temperature [1], w1/w2 [E,M,H], w3 [E,H,M].
    The weights use a wider spread than the reference's normal(0, 0.01) so that outputs are not
    vanishingly small next to the comparison tolerances."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn([T, M], generator=g)
    proj_w = (torch.rand([P, M], generator=g) * 2 - 1) / math.sqrt(M)
    proj_b = (torch.rand([P], generator=g) * 2 - 1) / math.sqrt(M)
    sim = torch.randn([P, E], generator=g) * 0.01
    temperature = torch.log(torch.full([1], 1.0 / 0.5))
    w1 = torch.randn([E, M, H], generator=g) / math.sqrt(M)
    w2 = torch.randn([E, M, H], generator=g) / math.sqrt(M)
    w3 = torch.randn([E, H, M], generator=g) / math.sqrt(H)
    return [t.to(dtype) for t in (x, proj_w, proj_b, sim, temperature, w1, w2, w3)]
