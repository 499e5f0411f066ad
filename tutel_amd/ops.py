"""Tensor-level wrappers over the C ABI (include/tutel_amd.h).

PyTorch is plumbing here: it owns the HBM allocations and the current HIP stream; every op
below hands raw device pointers + sizes + that stream to libtutel_amd.so.  All ops require HIP
device tensors and raise otherwise -- there is no CPU or eager implementation behind them.
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
ACT_CODES = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "gelu": _lib.ACT_GELU, "silu": _lib.ACT_SILU}


def _code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.TutelAmdError(f"tutel_amd: dtype {t.dtype} is not supported by the HIP kernels") from None


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.TutelAmdError(
                "tutel_amd: the MoE hot path runs on the HIP device only (got a CPU tensor); "
                "there is no CPU fallback.")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _a16(t):
    """an ACTIVATION operand at a 16-byte aligned address: the kernels fetch rows as 16-byte vectors, upstream's take any contiguous
    tensor -- an offset view (a slice of a larger buffer) is copied into a fresh allocation instead of failing the call.  (Weights are
    not copied per call: a misaligned parameter tensor fails loudly in the C ABI.)"""
    if t is None or t.data_ptr() % 16 == 0:
        return t
    return t.clone(memory_format=torch.contiguous_format)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream (raw getter: ~0.3 us instead of ~10 us for the Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def supported_dtype(dtype):
    return dtype in _DT


def routing_dtype(dtype):
    """dtypes the top-k kernel takes natively (fp64 scores are compared and normalised in fp64)."""
    return dtype in _DT or dtype == torch.float64


# ---------------------------------------------------------------------------------------------
# routing
# ---------------------------------------------------------------------------------------------
def routing_workspace(T, E, k, device):
    n = _lib.lib().tutel_amd_routing_workspace_bytes(T, E, k)
    return torch.empty([max(int(n), 4)], dtype=torch.uint8, device=device)


def gate_topk(inp, k, apply_softmax=False, normalize_gate=True, want_scores=False, ws=None, clear=None):
    """inp [T,E] scores (or logits with apply_softmax) -> idx [k,T] i32, gates [k,T], ws, scores|None.
    clear: optional int32 tensor this launch also fills with -1 (the slot_map of the next call)."""
    _dev(inp)
    assert inp.dim() == 2
    inp = inp.contiguous()
    T, E = inp.shape
    k = min(int(k), E)
    idx = torch.empty([k, T], dtype=torch.int32, device=inp.device)
    gates = torch.empty([k, T], dtype=inp.dtype, device=inp.device)
    scores = torch.empty_like(inp) if (want_scores and apply_softmax) else None
    if ws is None:
        ws = routing_workspace(T, E, k, inp.device)
    L = _lib.lib()
    code = _lib.F64 if inp.dtype == torch.float64 else _code(inp)
    _lib.check(L.tutel_amd_gate_topk(_ptr(inp), code, int(bool(apply_softmax)), T, E, k,
                                     int(bool(normalize_gate)), _ptr(scores), _ptr(idx), _ptr(gates),
                                     _ptr(ws), ws.numel(), _ptr(clear), clear.numel() if clear is not None else 0,
                                     _stream()), "tutel_amd_gate_topk")
    return idx, gates, ws, (scores if apply_softmax else inp)


def gate_proj_splits(T, M, E, dtype):
    """split count of the native gate projection for this shape (0: not covered -> F.linear + gate_topk)"""
    if dtype not in (torch.bfloat16, torch.float16):
        return 0
    return int(_lib.lib().tutel_amd_gate_proj_splits(int(T), int(M), int(E), _lib.BF16 if dtype == torch.bfloat16 else _lib.F16))


def gate_proj(x, wg, partials=None):
    """x [T, M], wg [E, M] (16-bit) -> fp32 partial sums [S, T, E] of x @ wg^T (split-K, csrc/gate_proj.hip); None when the shape is
    not covered.  gate_topk_partials() adds them in split order and rounds once to x.dtype."""
    _dev(x)
    assert x.dim() == 2 and wg.dim() == 2 and x.shape[1] == wg.shape[1] and x.dtype == wg.dtype
    T, M = x.shape
    E = wg.shape[0]
    S = gate_proj_splits(T, M, E, x.dtype)
    if S == 0 or T == 0:
        return None
    x, wg = x.contiguous(), wg.contiguous()
    if partials is None:
        partials = torch.empty([S, T, E], dtype=torch.float32, device=x.device)
    assert partials.numel() >= S * T * E and partials.dtype == torch.float32
    _lib.check(_lib.lib().tutel_amd_gate_proj(_ptr(x), _ptr(wg), _code(x), T, M, E, _ptr(partials), partials.numel() * 4, _stream()),
               "tutel_amd_gate_proj")
    return partials


def gate_topk_partials(partials, dtype, k, normalize_gate=True, want_logits=False, want_scores=False, ws=None, clear=None):
    """partials [S, T, E] fp32 (gate_proj) -> idx [k,T] i32, gates [k,T] dtype, ws, logits|None, scores|None: softmax + top-k on
    logits = dtype(sum_s partials[s]), exactly as gate_topk(logits, apply_softmax=True)."""
    _dev(partials)
    assert partials.dim() == 3 and partials.dtype == torch.float32 and partials.is_contiguous()
    S, T, E = partials.shape
    k = min(int(k), E)
    dev = partials.device
    idx = torch.empty([k, T], dtype=torch.int32, device=dev)
    gates = torch.empty([k, T], dtype=dtype, device=dev)
    logits = torch.empty([T, E], dtype=dtype, device=dev) if want_logits else None
    scores = torch.empty([T, E], dtype=dtype, device=dev) if want_scores else None
    if ws is None:
        ws = routing_workspace(T, E, k, dev)
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F16
    _lib.check(_lib.lib().tutel_amd_gate_topk_partials(_ptr(partials), S, code, T, E, k, int(bool(normalize_gate)), _ptr(logits),
                                                       _ptr(scores), _ptr(idx), _ptr(gates), _ptr(ws), ws.numel(), _ptr(clear),
                                                       clear.numel() if clear is not None else 0, _stream()),
               "tutel_amd_gate_topk_partials")
    return idx, gates, ws, logits, scores


def gate_logits(x, wg):
    """x @ wg^T in x.dtype through the split-K kernel: the SAME logits, bit for bit, that the one-call path routes on (its top-k kernel
    adds the same partial sums in the same order) -- so that a layer's routing does not depend on which path its planner picks.
    None when the shape is not covered."""
    part = gate_proj(x, wg)
    if part is None:
        return None
    return gate_topk_partials(part, x.dtype, 1, want_logits=True)[3]


def cache_warm(t, chunk_bytes=None, n_chunks=1, stride_bytes=0, blocks=0, offset=0):
    """plain loads over n_chunks ranges of chunk_bytes (stride_bytes apart) of tensor t, from byte `offset` (memory-side cache
    warm-up on the current stream, csrc/gate_proj.hip)"""
    _dev(t)
    total = t.numel() * t.element_size()
    chunk_bytes = total - offset if chunk_bytes is None else int(chunk_bytes)
    assert offset + (n_chunks - 1) * stride_bytes + chunk_bytes <= total
    _lib.check(_lib.lib().tutel_amd_cache_warm(t.data_ptr() + offset, chunk_bytes, int(n_chunks), int(stride_bytes), int(blocks), None, _stream()),
               "tutel_amd_cache_warm")


def compute_location(idx, E, ws=None, capacity=0, want_l_aux=False, l_aux_dtype=torch.float32, cleared_slot_map=None):
    """idx [k,T] -> loc [k,T], dispatch_count [E], stats [1] (max count), l_aux [1]|None, slot_map|None.
    cleared_slot_map: an [E*capacity] int32 tensor already filled with -1 (see gate_topk(clear=...))."""
    _dev(idx)
    assert idx.dtype == torch.int32 and idx.dim() == 2 and idx.is_contiguous()
    k, T = idx.shape
    dev = idx.device
    hist_ready = ws is not None
    if ws is None:
        ws = routing_workspace(T, E, k, dev)
    loc = torch.empty_like(idx)
    cnt = torch.empty([E], dtype=torch.int32, device=dev)
    stats = torch.empty([1], dtype=torch.int32, device=dev)
    l_aux = torch.empty([1], dtype=l_aux_dtype, device=dev) if (want_l_aux and hist_ready) else None
    smap = None
    if capacity > 0:
        if cleared_slot_map is not None:
            assert cleared_slot_map.numel() == E * capacity and cleared_slot_map.dtype == torch.int32
            smap = cleared_slot_map
        else:
            smap = torch.empty([E * capacity], dtype=torch.int32, device=dev)
    L = _lib.lib()
    _lib.check(L.tutel_amd_compute_location(_ptr(idx), T, E, k, int(hist_ready), _ptr(ws), ws.numel(),
                                            _ptr(loc), _ptr(cnt), _ptr(stats), _ptr(l_aux),
                                            _code(l_aux) if l_aux is not None else 0,
                                            int(capacity), _ptr(smap), int(cleared_slot_map is not None), _stream()),
               "tutel_amd_compute_location")
    return loc, cnt, stats, l_aux, smap


def slot_map(idx, loc, E, capacity):
    _dev(idx, loc)
    assert idx.dtype == torch.int32 and loc.dtype == torch.int32
    assert idx.is_contiguous() and loc.is_contiguous() and idx.shape == loc.shape
    k, T = idx.shape
    smap = torch.empty([E * capacity], dtype=torch.int32, device=idx.device)
    _lib.check(_lib.lib().tutel_amd_slot_map(_ptr(idx), _ptr(loc), T, E, k, int(capacity), _ptr(smap),
                                             _stream()), "tutel_amd_slot_map")
    return smap


def cumsum_sub_one(mask):
    _dev(mask)
    m = mask.to(torch.int32).contiguous()
    out = torch.empty_like(m)
    _lib.check(_lib.lib().tutel_amd_cumsum_sub_one(_ptr(m), _ptr(out), m.shape[0], m.shape[1], _stream()),
               "tutel_amd_cumsum_sub_one")
    return out


# ---------------------------------------------------------------------------------------------
# dispatch / combine
# ---------------------------------------------------------------------------------------------
def fast_encode(x, smap, gates, n_slots, capacity=0, num_experts=0, chunk_rows=0, expert_slice=0, ep_world=1):
    """x [T,M] -> [n_slots, M]; gates [k,T] or None (is_postscore).  smap is always in plain bucket order;
    chunk_rows / expert_slice select the order of the OUTPUT rows (see fast_decode)."""
    _dev(x, smap, gates)
    assert x.dim() == 2 and x.is_contiguous() and smap.dtype == torch.int32 and smap.numel() == n_slots
    x = _a16(x)
    T, M = x.shape
    out = torch.empty([n_slots, M], dtype=x.dtype, device=x.device)
    if gates is not None:
        assert gates.is_contiguous()
    _lib.check(_lib.lib().tutel_amd_fast_encode(_ptr(x), _code(x), _ptr(smap), _ptr(gates),
                                                _code(gates) if gates is not None else 0, T, M,
                                                int(n_slots), int(capacity), int(num_experts), int(chunk_rows),
                                                int(expert_slice), int(ep_world), _ptr(out), _stream()),
               "tutel_amd_fast_encode")
    return out


def fast_decode(buf, idx, loc, gates, capacity, num_experts=0, chunk_rows=0, expert_slice=0, ep_world=1):
    """buf [E*C, M] -> [T, M]; gates [k,T] or None (= ones).
    chunk_rows > 0: buf is chunk-major [C/chunk_rows, E, chunk_rows, M]; expert_slice = s > 0: buf is
    expert-sliced [E_loc/s, W, s, C, M] (the two layouts of the overlapped all-to-all)."""
    _dev(buf, idx, loc, gates)
    assert buf.dim() == 2 and buf.is_contiguous()
    buf = _a16(buf)
    assert idx.dtype == torch.int32 and loc.dtype == torch.int32 and idx.is_contiguous() and loc.is_contiguous()
    k, T = idx.shape
    M = buf.shape[1]
    out = torch.empty([T, M], dtype=buf.dtype, device=buf.device)
    if gates is not None:
        assert gates.is_contiguous() and gates.shape == idx.shape
    _lib.check(_lib.lib().tutel_amd_fast_decode(_ptr(buf), _code(buf), _ptr(idx), _ptr(loc), _ptr(gates),
                                                _code(gates) if gates is not None else 0, T, M, k,
                                                int(capacity), int(num_experts), int(chunk_rows), int(expert_slice), int(ep_world),
                                                _ptr(out), _stream()),
               "tutel_amd_fast_decode")
    return out


def gate_grad(x, buf, idx, loc, capacity):
    """ggate [k,T] fp32 = <buf[slot(j,t)], x[t]>."""
    _dev(x, buf, idx, loc)
    assert x.is_contiguous() and buf.is_contiguous() and x.dtype == buf.dtype
    x, buf = _a16(x), _a16(buf)
    k, T = idx.shape
    M = x.shape[1]
    out = torch.empty([k, T], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tutel_amd_gate_grad(_ptr(x), _ptr(buf), _code(x), _ptr(idx), _ptr(loc), T, M, k,
                                              int(capacity), _ptr(out), _stream()), "tutel_amd_gate_grad")
    return out


# ---------------------------------------------------------------------------------------------
# expert grouped GEMM
# ---------------------------------------------------------------------------------------------
def gemm_supported(dtype, N, K):
    return dtype in (torch.bfloat16, torch.float16) and K % 64 == 0 and N % 8 == 0


def expert_gemm(a, w, bias, w_kmajor, act="none", E_loc=None, R=None, a_layout=None, out=None,
                d_layout=None, row_counts=None, row_align=1, mul=None):
    """D[e,r,:] = act(A[e,r,:] @ op(W[e]) + bias[e]) [* mul[e,r,:]].

    a: [E_loc, R, K] contiguous, or any buffer described by a_layout=(stride_e, stride_w, rows_per_w, lda)
    w: [E_loc, N, K] (w_kmajor) or [E_loc, K, N]; bias [E_loc, N] or None.
    out/d_layout likewise (default: new contiguous [E_loc, R, N]); mul (GLU gating operand) has
    the output's layout and dtype."""
    _dev(a, w, bias, out, row_counts, mul)
    assert w.dim() == 3 and w.is_contiguous()
    if w_kmajor:
        El, N, K = w.shape
    else:
        El, K, N = w.shape
    E_loc = El if E_loc is None else E_loc
    if a_layout is None:
        assert a.dim() == 3 and a.is_contiguous() and a.shape[0] == E_loc and a.shape[2] == K
        a = _a16(a)
        R = a.shape[1]
        a_layout = (R * K, 0, R, K)
    assert R is not None
    if out is None:
        out = torch.empty([E_loc, R, N], dtype=a.dtype, device=a.device)
        d_layout = (R * N, 0, R, N)
    assert d_layout is not None
    if bias is not None:
        assert bias.is_contiguous() and bias.shape[-1] == N and bias.dtype == a.dtype
    assert w.dtype == a.dtype
    if mul is not None:
        assert mul.dtype == a.dtype and mul.is_contiguous()
        _lib.check(_lib.lib().tutel_amd_expert_gemm_glu(
            _ptr(a), a_layout[0], a_layout[1], a_layout[2], a_layout[3],
            _ptr(w), int(bool(w_kmajor)), w.stride(0), w.stride(1),
            _ptr(bias), (bias.stride(0) if bias is not None else 0), _ptr(mul),
            _ptr(out), d_layout[0], d_layout[1], d_layout[2], d_layout[3],
            E_loc, R, N, K, _code(a), ACT_CODES[act],
            _ptr(row_counts), int(row_align), _stream()), "tutel_amd_expert_gemm_glu")
        return out
    _lib.check(_lib.lib().tutel_amd_expert_gemm(
        _ptr(a), a_layout[0], a_layout[1], a_layout[2], a_layout[3],
        _ptr(w), int(bool(w_kmajor)), w.stride(0), w.stride(1),
        _ptr(bias), (bias.stride(0) if bias is not None else 0),
        _ptr(out), d_layout[0], d_layout[1], d_layout[2], d_layout[3],
        E_loc, R, N, K, _code(a), ACT_CODES[act],
        _ptr(row_counts), int(row_align), _stream()), "tutel_amd_expert_gemm")
    return out


_zero_rows = {}


def expert_gemm_gather(x, smap, w, bias, w_kmajor, act, R, row_counts=None, row_align=1):
    """fc1 with fast_encode fused: D[e,r,:] = act(x[smap[e*R+r] % T] @ op(W[e]) + bias[e]), zero row where smap < 0."""
    _dev(x, smap, w, bias, row_counts)
    assert x.dim() == 2 and x.is_contiguous() and w.dim() == 3 and w.is_contiguous() and w.dtype == x.dtype
    E_loc, N, K = (w.shape if w_kmajor else (w.shape[0], w.shape[2], w.shape[1]))
    assert x.shape[1] == K and smap.dtype == torch.int32 and smap.numel() == E_loc * R
    x = _a16(x)
    key = (x.device, x.dtype)
    z = _zero_rows.get(key)
    if z is None or z.numel() < K:
        z = _zero_rows[key] = torch.zeros([max(K, 8192)], dtype=x.dtype, device=x.device)
    out = torch.empty([E_loc, R, N], dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().tutel_amd_expert_gemm_gather(
        _ptr(x), x.stride(0), _ptr(smap), x.shape[0], _ptr(z), _ptr(w), int(bool(w_kmajor)), w.stride(0), w.stride(1),
        _ptr(bias), (bias.stride(0) if bias is not None else 0), _ptr(out), R * N, N, E_loc, R, N, K, _code(x),
        ACT_CODES[act], _ptr(row_counts), int(row_align), _stream()), "tutel_amd_expert_gemm_gather")
    return out


def expert_ffn(x, w1, b1, w2_kmajor, b2, act, R=None, smap=None, hid=None):
    """fc1 -> activation -> fc2 of every local expert in ONE persistent launch (tutel_amd_expert_ffn, csrc/expert_ffn.hip):
    x [E_loc, R, M] -- or the token array [T, M] with smap [E_loc * R] (fast_encode fused) -- w1 [E_loc, H, M], w2_kmajor
    [E_loc, M_out, H] (the k-major copy of batched_fc2_w) -> [E_loc, R, M_out].  Returns None when the library answers ENOTSUP
    (shape / layout not covered, or TUTEL_OPT_FFN_FUSED at its default: the persistent launch is opt-in, the two launches measured
    faster): the caller then runs two expert_gemm launches -- same bits."""
    _dev(x, w1, b1, w2_kmajor, b2, smap)
    assert w1.dim() == 3 and w2_kmajor.dim() == 3 and w1.is_contiguous() and w2_kmajor.is_contiguous() and w1.dtype == x.dtype == w2_kmajor.dtype
    E_loc, H, M = w1.shape
    M_out = w2_kmajor.shape[1]
    assert w2_kmajor.shape == (E_loc, M_out, H) and x.is_contiguous()
    x = _a16(x)
    if smap is None:
        assert x.dim() == 3 and x.shape[0] == E_loc and x.shape[2] == M
        R = x.shape[1]
        xs, ldx, T, z = R * M, M, 0, None
    else:
        assert x.dim() == 2 and x.shape[1] == M and smap.dtype == torch.int32 and smap.numel() == E_loc * R
        key = (x.device, x.dtype)
        z = _zero_rows.get(key)
        if z is None or z.numel() < M:
            z = _zero_rows[key] = torch.zeros([max(M, 8192)], dtype=x.dtype, device=x.device)
        xs, ldx, T = 0, x.stride(0), x.shape[0]
    if hid is None:
        hid = torch.empty([E_loc, R, H], dtype=x.dtype, device=x.device)
    out = torch.empty([E_loc, R, M_out], dtype=x.dtype, device=x.device)
    rc = _lib.lib().tutel_amd_expert_ffn(_ptr(x), xs, ldx, _ptr(smap), T, _ptr(z), _ptr(w1), w1.stride(0), w1.stride(1), _ptr(b1),
                                         (b1.stride(0) if b1 is not None else 0), _ptr(hid), R * H, H, _ptr(w2_kmajor), w2_kmajor.stride(0),
                                         w2_kmajor.stride(1), _ptr(b2), (b2.stride(0) if b2 is not None else 0), _ptr(out), R * M_out, M_out,
                                         E_loc, R, M, H, M_out, _code(x), ACT_CODES[act], _stream())
    if rc == _lib.ENOTSUP:
        return None
    _lib.check(rc, "tutel_amd_expert_ffn")
    return out


_OPT_ENV = {_lib.OPT_GEMM_IMPL: "TUTEL_AMD_GEMM_IMPL", _lib.OPT_GEMM_TILE: "TUTEL_AMD_GEMM_BIG", _lib.OPT_DECODE: "TUTEL_AMD_DECODE",
            _lib.OPT_EP_STAGE_GRID: "TUTEL_AMD_EP_STAGE_GRID", _lib.OPT_GEMM_PERSIST: "TUTEL_AMD_GEMM_PERSIST", _lib.OPT_EP_STREAMS: "TUTEL_AMD_EP_STREAMS",
            _lib.OPT_EP_CANARY: "TUTEL_AMD_EP_CANARY", _lib.OPT_GEMM_SPLITK: "TUTEL_AMD_GEMM_SPLITK", _lib.OPT_GEMM_STORE: "TUTEL_AMD_GEMM_STORE",
            _lib.OPT_GEMM_GATHER: "TUTEL_AMD_GEMM_GATHER", _lib.OPT_FUSED_LOCATION: "TUTEL_AMD_FUSED_LOCATION", _lib.OPT_TIE_RULE: "TUTEL_AMD_TIE_RULE", _lib.OPT_FFN_FUSED: "TUTEL_AMD_FFN_FUSED"}
_opts = {}


def set_option(key, value):
    """Tuning knob (_lib.OPT_*): -1 automatic, 0 / 1 forced.  For A/B runs and tests."""
    _lib.check(_lib.lib().tutel_amd_set_option(int(key), int(value)), "tutel_amd_set_option")
    _opts[int(key)] = int(value)


def get_option(key):
    """the knob's current value as the host code sees it (the library seeds its own copy from the same environment variable)"""
    import os
    if int(key) in _opts:
        return _opts[int(key)]
    try:
        return int(os.environ.get(_OPT_ENV[int(key)], "-1"))
    except ValueError:
        return -1


def probe_tr16():
    out = torch.empty([256], dtype=torch.int16, device="cuda")
    _lib.check(_lib.lib().tutel_amd_probe_tr16(_ptr(out), _stream()), "tutel_amd_probe_tr16")
    return out


def stage_timing(mode):
    """0 off; 1 HIP events around every kernel launch of the C ABI from now on; 2 around the two expert GEMMs only
    (measurement only; see include/tutel_amd.h)."""
    _lib.check(_lib.lib().tutel_amd_stage_timing(int(mode)), "tutel_amd_stage_timing")


def stage_report():
    """{stage: (total_us, launches)} of the launches recorded since the last report (waits for them)."""
    import ctypes
    n = len(_lib.STAGES)
    tot, cnt = (ctypes.c_double * n)(), (ctypes.c_int * n)()
    _lib.check(_lib.lib().tutel_amd_stage_report(tot, cnt, n), "tutel_amd_stage_report")
    return {name: (float(tot[i]), int(cnt[i])) for i, name in enumerate(_lib.STAGES)}



def mark():
    """record a step mark on the current stream (see tutel_amd_mark)"""
    _lib.check(_lib.lib().tutel_amd_mark(_stream()), "tutel_amd_mark")


def marks_reserve(n):
    _lib.check(_lib.lib().tutel_amd_marks_reserve(int(n)), "tutel_amd_marks_reserve")


def marks_report(n):
    """milliseconds between consecutive marks recorded since the last report"""
    import ctypes
    buf = (ctypes.c_double * max(n, 1))()
    m = _lib.lib().tutel_amd_marks_report(buf, int(n))
    return [buf[i] * 1e-3 for i in range(m)]
