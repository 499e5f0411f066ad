"""GPU parity tests, layer level: tutel.moe.moe_layer / the low-level tutel.moe API on the
MI355X against (a) fixtures produced by the reference itself and (b) the CPU oracle.

Tolerances (BASELINE.json north_star: 1e-5 fp32 / 1e-3 fp16, bit-exact index assignment):
  fp32 / fp64 : |y - y_ref| <= 1e-5 * max(1, |y_ref|_inf)   (ATen fp32/fp64 GEMMs on both sides,
                summation order differs between CPU and GPU BLAS)
  fp16        : <= 1e-3 absolute
  bf16        : bf16 has 8 significand bits; the bar is 2 bf16 ulps of the result magnitude
                (rtol 2^-7) + 2e-3 absolute against an fp32-accumulated oracle on the same
                bf16-rounded inputs (SURVEY section 7 hard part 7).  Against the reference's OWN
                bf16 fixture (which rounds to bf16 after each of its 4 ATen ops: bmm, +bias, bmm,
                +bias) the bar is 4 bf16 ulps of the output scale: 2^-6 * |y_ref|_inf.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DT = {"float32": torch.float32, "float64": torch.float64, "bfloat16": torch.bfloat16, "float16": torch.float16}


def _t(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.view(torch.bfloat16) if dtype == torch.bfloat16 else t


def make_layer(M, H, E, k, cf, dtype, weights=None, **kw):
    from tutel import moe
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        gate = {"type": "top", "k": k, "capacity_factor": cf}
        gate.update(kw.pop("gate", {}))
        layer = moe.moe_layer(gate_type=gate,
                              experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)},
                              model_dim=M, **kw)
    finally:
        torch.set_default_dtype(old)
    if weights is not None:
        wg, w1, b1, w2, b2 = weights
        with torch.no_grad():
            layer.gates[0].wg.weight.copy_(wg.to(layer.gates[0].wg.weight.dtype))
            layer.experts.batched_fc1_w.copy_(w1)
            layer.experts.batched_fc1_bias.copy_(b1)
            layer.experts.batched_fc2_w.copy_(w2)
            layer.experts.batched_fc2_bias.copy_(b2)
    return layer.cuda()


def _close(y, ref, dtype, vs_lowprec_reference=False):
    y, ref = y.double().cpu(), ref.double().cpu()
    err = (y - ref).abs()
    if vs_lowprec_reference and dtype == torch.bfloat16:
        assert float(err.max()) <= 2 ** -6 * float(ref.abs().max()), (float(err.max()), float(ref.abs().max()))
        return
    if dtype in (torch.float32, torch.float64):
        assert float(err.max()) <= 1e-5 * max(1.0, float(ref.abs().max())), float(err.max())
    elif dtype == torch.float16:
        assert float(err.max()) <= 1e-3, float(err.max())
    else:
        assert bool((err <= 2 ** -7 * ref.abs() + 2e-3).all()), float(err.max())


def test_hip_library_is_the_loaded_code():
    from tutel_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "tutel_amd/lib/libtutel_amd.so" in maps


CASES = sorted(glob.glob(os.path.join(GOLD, "layer_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_layer_vs_reference_fixture(oracle, path):
    z = np.load(path)
    T, M, H, E, k, fp32_gate, post, norm, seed = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    layer = make_layer(M, H, E, k, cf, dtype, weights, is_postscore=bool(post), normalize_gate=bool(norm),
                       gate={"fp32_gate": bool(fp32_gate)}).eval()
    layer._keep_routing = True
    with torch.no_grad():
        y = layer(x.cuda())
    stride = int(z["y_row_stride"][0])
    # token -> expert/slot assignment of THIS forward (whatever path it took), element by element against the reference's
    # idx / loc (only the order-free digest dispatch_count used to be compared on the layer path)
    idx, loc = layer.last_routing
    assert torch.equal(idx.cpu(), torch.from_numpy(z["idx"]).to(torch.int32)), "expert ids"
    assert torch.equal(loc.cpu(), torch.from_numpy(z["loc"]).to(torch.int32)), "slots"
    assert torch.equal(layer.dispatch_count.cpu(), torch.from_numpy(z["dispatch_count"]))
    assert y.dtype == dtype and y.shape == (T, M)
    _close(y[::stride], _t(z["y"], dtype), dtype, vs_lowprec_reference=True)
    assert abs(float(y.l_aux) - float(z["l_aux"][0])) <= (1e-5 if dtype == torch.float32 or fp32_gate else 1e-2)


@pytest.mark.parametrize("dts", ["bfloat16", "float16"])
def test_headline_low_precision_gate_assignment_vs_reference(oracle, dts):
    """north_star: "bit-exact for token-to-expert index assignment" -- checked for the configuration bench.py TIMES (16-bit gate,
    `fp32_gate=False`, the gate projection of csrc/gate_proj.hip inside the one native call) against the REFERENCE's own CPU result at
    the headline shape (tests/golden/headline_gate_*.npz: its logits, scores, idx, loc, counts, written by the live reference).
    The layer's forward is run as the bench runs it and its own logits / routing are read back (`_keep_routing`).  Equality, not a bound:
      * on every token whose 64 logits AND 64 scores carry the reference's bits, idx must be the reference's -- the 89 (bf16) / 15 (fp16)
        tokens whose scores tie exactly at the k / k+1 boundary included (TUTEL_OPT_TIE_RULE: torch.topk's CPU order, csrc/topk_ties.h);
      * a token may differ only if one of its logits rounds the other way in its last bit (the split-K MFMA sum and the host's 16-bit GEMM
        add in different orders -- the reference's own value moves with the host's instruction set) or one of its scores
        does (exp is not bit-specified across implementations); those tokens are counted and written to gpurun_out/;
      * if NO token differs, loc and dispatch_count are the fixture's, element for element; else they are on the prefix before the first."""
    import json
    from tutel_amd import ops
    dtype = DT[dts]
    z = np.load(os.path.join(GOLD, f"headline_gate_{dts}.npz"))
    T, M, _, E, k, seed = [int(v) for v in z["meta"]]
    H = 2048   # the benched layer (the fixture built only the gate; make_problem draws x and wg first, whatever H is)
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    assert abs(float(z["in_checksum"][0]) - float(x.double().abs().sum() + weights[0].double().abs().sum())) < 1e-6 * float(z["in_checksum"][0])
    layer = make_layer(M, H, E, k, 1.0, dtype, weights).eval()
    layer._keep_routing, layer.last_logits = True, None
    with torch.no_grad():
        layer(x.cuda())
    assert layer.last_logits is not None, "the gate projection ran inside the native call (the path bench.py times)"
    L_p = layer.last_logits.clone()
    idx_p, loc_p = [t.cpu() for t in layer.last_routing]
    S_p = ops.gate_topk(L_p, k, apply_softmax=True, want_scores=True)[3].cpu()   # the kernel's scores on those logits (same kernel, same bits)
    L_p = L_p.cpu()
    L_r, S_r = torch.from_numpy(z["logits"]).view(dtype), torch.from_numpy(z["scores"]).view(dtype)
    idx_r, loc_r = torch.from_numpy(z["idx"]), torch.from_numpy(z["loc"])
    logit_rows = (L_p.view(torch.int16) != L_r.view(torch.int16)).any(1)
    score_rows = (S_p.view(torch.int16) != S_r.view(torch.int16)).any(1) & ~logit_rows
    same_bits = ~(logit_rows | score_rows)
    differs = (idx_p != idx_r).any(0)
    # (1) same input bits -> the reference's expert ids, ties included
    assert not bool((differs & same_bits).any()), f"{int((differs & same_bits).sum())} tokens with the reference's logits and scores route differently"
    s32 = S_r.float()
    top = torch.topk(s32, k + 1, dim=1).values
    tie_rows = (top[:, 1:] == top[:, :-1]).any(1)
    assert int((tie_rows & same_bits).sum()) >= (60 if dts == "bfloat16" else 8), "the fixture's tied rows are what this test is about"
    # (2) the oracle, routed on the product's own scores, agrees everywhere (tie rule included)
    crit, _ = oracle.extract_critical(S_p, k, 1.0)
    assert torch.equal(torch.stack(crit[1]).to(torch.int32), idx_p) and torch.equal(torch.stack(crit[2]), loc_p)
    assert torch.equal(layer.dispatch_count.cpu(), crit[5])
    diff = torch.nonzero(differs).flatten().tolist()
    rec = dict(dtype=dts, tokens=T, assignments=k * T, path="MOELayer.forward -> tutel_amd_moe_forward (gate_proj.hip + routing.hip, TUTEL_OPT_TIE_RULE automatic)",
               rows_with_exact_ties_among_top_k_plus_1=int(tie_rows.sum()), tied_rows_with_the_references_bits=int((tie_rows & same_bits).sum()),
               differing_tokens=len(diff), differing_assignments=int((idx_p != idx_r).sum()),
               differing_tokens_with_the_references_logits_and_scores=int((differs & same_bits).sum()),
               tokens_with_a_logit_that_rounds_differently=int(logit_rows.sum()), logits_that_differ=int((L_p.view(torch.int16) != L_r.view(torch.int16)).sum()),
               tokens_with_a_score_that_rounds_differently=int(score_rows.sum()), scores_that_differ=int((S_p.view(torch.int16) != S_r.view(torch.int16)).sum()),
               dispatch_count_equal=bool(torch.equal(layer.dispatch_count.cpu(), torch.from_numpy(z["dispatch_count"]))),
               loc_equal=bool(torch.equal(loc_p, loc_r)))
    out = os.path.join(os.path.dirname(GOLD.rstrip("/")), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"headline_gate_assignment_{dts}.json"), "w") as f:
            json.dump(rec, f)
    print(rec)
    # (3) rows that may differ at all are few: a last-bit rounding has to land on the k / k+1 boundary to move an assignment
    assert len(diff) <= 8, rec
    if not diff:
        assert torch.equal(loc_p, loc_r) and torch.equal(layer.dispatch_count.cpu(), torch.from_numpy(z["dispatch_count"]))
    first = diff[0] if diff else T
    assert torch.equal(loc_p[0, :first], loc_r[0, :first])


NOISY = sorted(glob.glob(os.path.join(GOLD, "noisy_*.npz")))


@pytest.mark.parametrize("path", NOISY, ids=lambda p: os.path.basename(p)[6:-4])
def test_noisy_gate_load_importance_layer_vs_reference_fixture(oracle, monkeypatch, path):
    """The layer path of moe_layer.py:285-296 -- is_gshard_loss=False (load-importance loss, losses.py:21-42) and, in
    training, gate_noise > 0 -- against the REFERENCE's own output: the fixture stores the reference's randn_like draw,
    which is fed to this layer in place of its own (grad enabled, as in training).  Routing bit-exact; y at the dtype's bar."""
    z = np.load(path)
    assert len(NOISY) >= 3
    T, M, H, E, k, fp32_gate, training, seed = [int(v) for v in z["meta"]]
    dtype, gate_noise = DT[str(z["dtype"][0])], float(z["gate_noise"][0])
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    layer = make_layer(M, H, E, k, 1.0, dtype, weights, is_gshard_loss=False, gate={"fp32_gate": bool(fp32_gate), "gate_noise": gate_noise})
    layer.train(bool(training))
    assert layer.gates[0].gate_noise == gate_noise
    noise = torch.from_numpy(z["noise"])
    draws = []
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: draws.append(1) or noise.to(device=t.device, dtype=t.dtype))
    if training:
        y = layer(x.cuda())          # grad enabled: the configuration this branch exists for
        assert y.requires_grad and y.l_aux.requires_grad
    else:
        with torch.no_grad():
            y = layer(x.cuda())
    assert len(draws) == (1 if training else 0), "the noise is drawn once per training forward, never in eval"
    assert torch.equal(layer.dispatch_count.cpu(), torch.from_numpy(z["dispatch_count"]))
    _close(y.detach(), _t(z["y"], dtype), dtype, vs_lowprec_reference=True)
    want = float(z["l_aux"][0])
    assert abs(float(y.l_aux) - want) <= 2e-4 * max(1.0, abs(want)), (float(y.l_aux), want)
    if training:   # and the loss reaches the router: d l_aux / d wg is non-zero and finite
        y.l_aux.backward()
        g = layer.gates[0].wg.weight.grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


@pytest.mark.parametrize("path", [p for p in CASES if "c0_" not in p], ids=lambda p: os.path.basename(p)[6:-4])
def test_low_level_api_vs_reference_fixture(oracle, path):
    """tutel.moe.top_k_routing / fast_encode / fast_decode on the fixture's scores."""
    from tutel import moe
    z = np.load(path)
    T, M, H, E, k, fp32_gate, post, norm, seed = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    gdt = torch.float32 if fp32_gate else dtype
    scores = _t(z["scores"], gdt).cuda()
    crit, l_aux = moe.top_k_routing(scores, k, capacity_factor=cf, normalize_gate=bool(norm))
    assert torch.equal(torch.stack(crit[1]).cpu(), torch.from_numpy(z["idx"]))
    assert torch.equal(torch.stack(crit[2]).cpu(), torch.from_numpy(z["loc"]))
    assert crit[4] == int(z["capacity"][0]) and crit[0] == E
    assert torch.equal(crit[5].cpu(), torch.from_numpy(z["dispatch_count"]))
    assert torch.equal(torch.stack(crit[3]).cpu(), _t(z["gates"], gdt)), "gates bit-exact"
    assert abs(float(l_aux) - float(z["l_aux"][0])) <= (1e-6 if gdt in (torch.float32, torch.float64) else 1e-2)
    if "encoded" in z.files:
        x = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)[0]
        enc = moe.fast_encode(x.cuda(), crit, bool(post))
        assert torch.equal(enc.cpu(), _t(z["encoded"], dtype)), "fast_encode bit-exact"
        # decode of the REFERENCE's expert output must reproduce the reference's layer output
        ffn = _t(z["expert_out"], dtype).cuda()
        dec = moe.fast_decode(ffn.to(gdt), crit, bool(post)).to(dtype)
        assert torch.equal(dec.cpu().float(), _t(z["y"], dtype).float()), "fast_decode reproduces the reference exactly"


def test_headline_integer_fixture_on_gpu():
    """BASELINE configs[1] routing (T=4096, E=64, k=2): reference's assignment, bit-exact."""
    from tutel import moe
    z = np.load(os.path.join(GOLD, "headline_integers.npz"))
    g = torch.Generator().manual_seed(int(z["seed"][0]))
    scores = torch.softmax(torch.randn([4096, 64], generator=g), dim=1).cuda()
    for tag, cf in (("cf1", 1.0), ("dropless", 0.0)):
        crit, l_aux = moe.top_k_routing(scores, 2, capacity_factor=cf)
        assert torch.equal(torch.stack(crit[1]).cpu(), torch.from_numpy(z[f"idx_{tag}"]))
        assert torch.equal(torch.stack(crit[2]).cpu(), torch.from_numpy(z[f"loc_{tag}"]))
        assert crit[4] == int(z[f"capacity_{tag}"][0])
        assert torch.equal(crit[5].cpu(), torch.from_numpy(z[f"count_{tag}"]))
        assert abs(float(l_aux) - float(z[f"l_aux_{tag}"][0])) < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_headline_shape_layer_vs_oracle(oracle, dtype):
    """BASELINE configs[1]: T=4096, M=H=2048, E=64, top-2, cf=1 -- the HIP path end to end (gate, fused routing, encode inside fc1, 2 MFMA
    grouped GEMMs, decode) vs the fp32-accumulating oracle, and its routing ELEMENT BY ELEMENT -- every (token, choice) expert id and bucket
    slot the forward really used (`_keep_routing`) -- against the REFERENCE's own routing of the same tokens through the same gate
    (tests/golden/headline_fp32gate_*.npz, written by the live reference from the layer's real logits; smallest relative gap among a
    row's three largest scores 4e-5, far above what another fp32 summation order of the gate GEMM moves)."""
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=0)
    z = np.load(os.path.join(GOLD, f"headline_fp32gate_{str(dtype).split('.')[-1]}.npz"))
    assert [int(v) for v in z["meta"]] == [T, M, 8, E, k, 0]   # (the fixture built only the gate: make_problem draws x and wg first)
    assert abs(float(z["in_checksum"][0]) - float(x.double().abs().sum() + weights[0].double().abs().sum())) < 1e-6 * float(z["in_checksum"][0])
    layer = make_layer(M, H, E, k, 1.0, dtype, weights, gate={"fp32_gate": True}).eval()
    layer._keep_routing = True
    with torch.no_grad():
        y = layer(x.cuda().view(16, 256, M))
    idx, loc = layer.last_routing
    assert torch.equal(idx.cpu(), torch.from_numpy(z["idx"])), "expert ids vs the reference, element-wise"
    assert torch.equal(loc.cpu(), torch.from_numpy(z["loc"])), "bucket slots vs the reference, element-wise"
    assert torch.equal(layer.dispatch_count.cpu(), torch.from_numpy(z["dispatch_count"])) and int(z["capacity"][0]) == 128
    assert abs(float(y.l_aux) - float(z["l_aux"][0])) < 1e-5
    yo, lo, crit, _ = oracle.moe_forward(x, *weights, top_k=k, capacity_factor=1.0, fp32_gate=True, accum_fp32=True)
    assert crit[4] == 128 and y.shape == (16, 256, M)
    assert torch.equal(idx.cpu(), torch.stack(crit[1]).to(torch.int32)) and torch.equal(loc.cpu(), torch.stack(crit[2]))
    assert torch.equal(layer.dispatch_count.cpu(), crit[5])
    _close(y.view(T, M), yo, dtype)
    assert abs(float(y.l_aux) - float(lo)) < 1e-5


def test_dropless_megablocks_equals_dense(oracle):
    """BASELINE configs[2]: capacity_factor=0 + megablocks_size>0 == the same problem with the
    dense expert GEMM on every row decode reads."""
    T, M, H, E, k = 2048, 256, 256, 16, 2
    dtype = torch.bfloat16
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=2)
    layer = make_layer(M, H, E, k, 0.0, dtype, weights, gate={"fp32_gate": True}).eval()
    with torch.no_grad():
        dense = layer(x.cuda())
        cap_dense = layer.protected_shape[1]
        mega = layer(x.cuda(), megablocks_size=4)
        assert layer.megablocks_size == 4 and layer.protected_shape[1] % 4 == 0 and layer.protected_shape[1] >= cap_dense
    assert torch.equal(dense, mega)
    yo, _, crit, _ = oracle.moe_forward(x, *weights, top_k=k, capacity_factor=0.0, fp32_gate=True, accum_fp32=True)
    assert cap_dense == crit[4] == int(crit[5].max())
    _close(dense, yo, dtype)


@pytest.mark.parametrize("fp32_gate", [True, False])
def test_dropless_headline_shape_vs_oracle(oracle, fp32_gate):
    """BASELINE configs[2] at the headline shape: T=4096, M=H=2048, E=64, top-2, capacity_factor=0
    (capacity = max expert load, 150-160 here: more than one 128-row tile per expert), megablocks_size
    0 and 4: dense == megablocks bitwise, both vs the fp32-accumulating oracle at the dropless capacity."""
    from tutel_amd import ops
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    dtype = torch.bfloat16
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=3)
    layer = make_layer(M, H, E, k, 0.0, dtype, weights, gate={"fp32_gate": fp32_gate}).eval()
    xd = x.cuda()
    layer._keep_routing, layer.last_logits = True, None
    with torch.no_grad():
        dense = layer(xd)
        cap_dense = int(layer.protected_shape[1])
        counts = layer.dispatch_count.cpu()
        # the logits the forward routed on: the in-call projection's (csrc/gate_proj.hip) when it ran, the library GEMM's otherwise
        logits = layer.last_logits if layer.last_logits is not None else layer.gates[0](xd)
        mega = layer(xd, megablocks_size=4)
        cap_mega = int(layer.protected_shape[1])
    assert cap_dense == int(counts.max()) and cap_dense > 128
    assert layer.megablocks_size == 4 and cap_mega == (cap_dense + 3) // 4 * 4
    assert torch.equal(dense, mega)
    if fp32_gate:
        yo, lo, crit, _ = oracle.moe_forward(x, *weights, top_k=k, capacity_factor=0.0, fp32_gate=True, accum_fp32=True)
    else:  # low-precision gate: the oracle routes on the scores the kernel derived from the library GEMM's logits
        scores = ops.gate_topk(logits, k, apply_softmax=True, want_scores=True)[3].cpu()
        wg, w1, b1, w2, b2 = weights
        crit, lo = oracle.extract_critical(scores, k, 0.0)
        yo = oracle.fast_decode(oracle.expert_ffn(oracle.fast_encode(x, crit), w1, b1, w2, b2, accum_fp32=True), crit)
    assert crit[4] == cap_dense and torch.equal(counts, crit[5])
    _close(dense, yo, dtype)


@pytest.mark.parametrize("name,E_loc,k", [("train_losses_top2_e2", 2, 2), ("train_losses_top1_e4", 4, 1)])
def test_training_replay_matches_reference_losses(oracle, name, E_loc, k):
    """fwd + bwd + SGD for a few steps (the style of the reference's golden-loss tests,
    tests/test_tutel.py:94-148): dispatch/combine backward on the HIP kernels."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    T, M, H, E, k_, steps, seed = [int(v) for v in z["meta"]]
    x, *weights = oracle.make_problem(T, M, H, E, seed=seed)
    layer = make_layer(M, H, E, k, 1.0, torch.float32, weights).train()
    opt = torch.optim.SGD(layer.parameters(), lr=1e-2)
    xb = x.cuda().view(4, T // 4, M)
    target = torch.zeros(4, dtype=torch.long, device="cuda")
    got = []
    for _ in range(steps):
        opt.zero_grad()
        out = layer(xb)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(out.sum(dim=2), dim=1), target) + 0.01 * out.l_aux
        loss.backward()
        opt.step()
        got.append(float(loss))
    ref = z["losses"]
    assert np.allclose(np.array(got), ref, rtol=2e-4, atol=2e-4), (got, ref.tolist())


def test_autocast_and_misc_paths(oracle):
    T, M, H, E, k = 512, 128, 128, 8, 2
    x, *weights = oracle.make_problem(T, M, H, E, seed=4)
    layer = make_layer(M, H, E, k, 1.0, torch.float32, weights).eval()
    with torch.no_grad():
        ref = layer(x.cuda())
        # autocast casts the tokens to bf16 before routing (moe_layer.py:26-39,265-266): compare
        # with the un-autocast layer on the same bf16-rounded tokens so the routing is identical
        ref_rounded = layer(x.bfloat16().float().cuda())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = layer(x.cuda())
        assert y.dtype == torch.float32
        assert float((y - ref_rounded).abs().max()) < 0.05 * float(ref_rounded.abs().max())
        y3 = layer(x.cuda().view(2, 4, T // 8, M), top_k=1, capacity_factor=2.0)  # per-call overrides, N-d input
        assert y3.shape == (2, 4, T // 8, M)
    res = make_layer(M, H, E, k, 1.0, torch.float32, weights, result_func=lambda t: t * 2).eval()
    with torch.no_grad():
        assert torch.equal(res(x.cuda()), ref * 2)


def test_decode_chunk_major_layout(oracle):
    """fast_decode on the chunk-major bucket layout [C/c, E, c, M] of the overlapped all-to-all."""
    from tutel_amd import ops
    g = torch.Generator().manual_seed(8)
    T, E, k, M = 700, 12, 2, 96
    scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
    crit, _ = oracle.extract_critical(scores, k, 1.0, alignment=4)
    _, idx_o, loc_o, gates_o, C, _ = crit
    for dtype in (torch.float32, torch.bfloat16):
        y = torch.randn([E, C, M], generator=g).to(dtype)
        want = oracle.fast_decode(y, crit)
        for c in (C // 4, C // 2, C):
            ycm = y.view(E, C // c, c, M).permute(1, 0, 2, 3).contiguous()
            got = ops.fast_decode(ycm.view(E * C, M).cuda(), torch.stack(idx_o).cuda(), torch.stack(loc_o).cuda(),
                                  torch.stack(gates_o).cuda(), C, num_experts=E, chunk_rows=c)
            assert torch.equal(got.cpu().float(), want.float()), (dtype, c)


@pytest.mark.parametrize("degree,E", [(2, 8), (4, 8), (4, 6), (2, 1)])   # E % degree == 0: expert-sliced stages, else capacity chunks
@pytest.mark.parametrize("post", [True, False])
def test_overlapped_path_equals_plain_path(oracle, monkeypatch, degree, E, post):
    """a2a_ffn_overlap_degree > 1 must not change the result (what the reference asserts in
    tests/test_tutel.py:161-176).  The copy-free overlapped routine (chunk-major buckets, comm
    stream + events) is forced to run on one rank so its stream discipline is exercised here."""
    from tutel_amd.impls import moe_layer as ml
    T, M, H, k = (1032 if E == 6 else 1024), 256, 256, min(2, E)   # capacity divisible by every degree tried, so the
    x, *weights = oracle.make_problem(T, M, H, E, dtype=torch.bfloat16, seed=6)   # alignment rule (moe_layer.py:298-301) does not change it
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights, is_postscore=post, gate={"fp32_gate": True}).eval()
    with torch.no_grad():
        plain = layer(x.cuda(), a2a_ffn_overlap_degree=1)
        monkeypatch.setattr(ml, "_FORCE_OVERLAP", True)
        for _ in range(3):  # repeated: allocator reuse across streams must stay safe
            over = layer(x.cuda(), a2a_ffn_overlap_degree=degree)
            torch.cuda.synchronize()
            assert torch.equal(plain, over)
    assert layer.protected_shape[1] % degree == 0


@pytest.mark.parametrize("amp_dtype", [torch.bfloat16, torch.float16])
def test_autocast_runs_the_mfma_gemm_vs_oracle(oracle, monkeypatch, amp_dtype):
    """torch.autocast over an fp32 layer (the reference's AMP recipe, examples/helloworld_amp.py:76-79): tokens travel in
    the autocast dtype (moe_layer.py:26-39,265-266) and the expert GEMMs run on the MFMA kernel with cached low-precision
    casts of the fp32 master weights (ATen autocasts matmul the same way) -- checked against the oracle on weights
    rounded to that dtype, and the casts follow weight updates."""
    from tutel_amd import ops
    T, M, H, E, k = 1024, 256, 512, 8, 2
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, dtype=torch.float32, seed=31)
    layer = make_layer(M, H, E, k, 1.0, torch.float32, (wg, w1, b1, w2, b2), gate={"fp32_gate": True}).eval()
    calls = []
    real = ops.expert_gemm
    monkeypatch.setattr(ops, "expert_gemm", lambda *a, **kw: calls.append(a[0].dtype) or real(*a, **kw))
    xd = x.cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=amp_dtype):
        y = layer(xd)
    assert y.dtype == torch.float32 and calls == [amp_dtype, amp_dtype], calls
    lp = lambda t: t.to(amp_dtype)
    yo, lo, crit, _ = oracle.moe_forward(lp(x), wg, lp(w1), lp(b1), lp(w2), lp(b2), top_k=k, fp32_gate=True, accum_fp32=True)
    assert torch.equal(layer.dispatch_count.cpu(), crit[5])
    _close(y, yo.float(), amp_dtype)
    with torch.no_grad():
        layer.experts.batched_fc2_w.mul_(2.0)       # in-place update bumps the version counter: the cached cast must follow
        layer.experts.batched_fc2_bias.mul_(2.0)
        with torch.autocast("cuda", dtype=amp_dtype):
            y2 = layer(xd)
    _close(y2, 2 * yo.float(), amp_dtype)
    # grad enabled + trainable experts (round 3): forward on the MFMA kernels WITH autograd (experts/ffn.py::_FFNTrain), the
    # data gradient on them too, and the gradients land on the fp32 master weights through the cast
    del calls[:]
    layer.train()
    with torch.autocast("cuda", dtype=amp_dtype):
        y3 = layer(xd)
    assert y3.requires_grad and calls == [amp_dtype, amp_dtype]
    _close(y3.detach(), 2 * yo.float(), amp_dtype)
    y3.float().square().mean().backward()
    gw = layer.experts.batched_fc1_w.grad
    assert gw is not None and gw.dtype == torch.float32 and bool(torch.isfinite(gw).all()) and float(gw.abs().max()) > 0
    assert calls == [amp_dtype] * 3   # + d hid (x does not require grad: no d x launch)


def test_native_pipeline_equals_python_orchestration(oracle, monkeypatch):
    """tutel_amd_ep_forward (one native call for encode .. decode) returns the bits of the Python-orchestrated
    paths it replaces: single rank (fused-encode route), with pre-score gates (encode route), fp16."""
    from tutel_amd.impls import ep_native
    calls = []
    real = ep_native.forward
    monkeypatch.setattr(ep_native, "forward", lambda *a, **kw: calls.append(1) or real(*a, **kw))
    for dtype, post, (T, M, H, E, k) in ((torch.bfloat16, True, (4096, 2048, 2048, 64, 2)), (torch.bfloat16, False, (1024, 256, 512, 8, 2)),
                                         (torch.float16, True, (1000, 320, 256, 5, 3))):
        x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=21)
        layer = make_layer(M, H, E, k, 1.0, dtype, weights, is_postscore=post).eval()
        xd = x.cuda()
        with torch.no_grad():
            monkeypatch.setattr(ep_native, "ENABLED", False)
            want = layer(xd).clone()
            want_aux, want_cnt = float(layer.l_aux), layer.dispatch_count.clone()
            monkeypatch.setattr(ep_native, "ENABLED", True)
            monkeypatch.setattr(ep_native, "FAST_PATH", False)
            del calls[:]
            for _ in range(3):   # cached workspace: repeated calls must stay correct
                got = layer(xd)
                assert torch.equal(got, want)
            assert len(calls) == 3, "the native pipeline must be the path taken"
            assert layer.protected_shape == torch.Size([E, layer.protected_shape[1], M])
            # routing + pipeline in one call (tutel_amd_moe_forward): same bits, same loss, same counts
            monkeypatch.setattr(ep_native, "FAST_PATH", True)
            fcalls = []
            real_fast = ep_native.forward_from_logits
            monkeypatch.setattr(ep_native, "forward_from_logits", lambda *a, **kw: fcalls.append(1) or real_fast(*a, **kw))
            for _ in range(3):
                got = layer(xd)
                assert torch.equal(got, want) and float(got.l_aux) == want_aux and torch.equal(layer.dispatch_count, want_cnt)
            assert len(fcalls) == 3 and len(calls) == 3, "the one-call path must be the path taken"
            monkeypatch.setattr(ep_native, "forward_from_logits", real_fast)


def test_native_pipeline_through_rccl_single_rank():
    """The staged native pipeline -- the library's own RCCL communicator (ncclCommInitRank from a broadcast id),
    its communication stream and event table, ncclAllToAll per stage -- forced onto one rank: degree 2 and 4
    (expert-sliced) and degree 3 (capacity-chunked, 8 % 3 != 0) return the bits of degree 1 and of the
    Python-orchestrated path."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tutel import moe
from tutel_amd.impls import ep_native
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
layer = moe.moe_layer(gate_type={"type": "top", "k": 2, "fp32_gate": True},
                      experts={"type": "ffn", "num_experts_per_device": 8, "hidden_size_per_expert": 512,
                               "activation_fn": lambda t: torch.nn.functional.relu(t)}, model_dim=256).cuda().eval()
torch.set_default_dtype(torch.float32)
x = torch.randn(1536, 256).bfloat16().cuda()      # capacity 2*ceil(1536/8) = 384: divisible by 2, 3, 4
with torch.no_grad():
    ep_native.ENABLED = False
    python_path = layer(x, a2a_ffn_overlap_degree=1).clone()
    ep_native.ENABLED = True
    plain = layer(x, a2a_ffn_overlap_degree=1).clone()
    assert torch.equal(plain, python_path)
    # degree 3 does not divide the 8 local experts: capacity chunks of 128 rows per launch instead of 384 -- another
    # K-tile order inside the GEMM (rotation is on below 256 rows per expert), so its reference is the SAME chunking
    # driven from Python (impls/overlap.py), bit for bit, and the plain result within one bf16 ulp of the output scale
    from tutel_amd.impls import moe_layer as ml, overlap as ov
    ep_native.ENABLED = False
    ml._FORCE_OVERLAP = ov._FORCE_RCCL = True
    chunked = layer(x, a2a_ffn_overlap_degree=3).clone()
    ml._FORCE_OVERLAP = ov._FORCE_RCCL = False
    ep_native.ENABLED = True
    assert float((chunked.float() - plain.float()).abs().max()) <= 2 ** -7 * float(plain.float().abs().max())
    ep_native._FORCE_COMM = True
    for degree in (1, 2, 4, 3, 2):
        for _ in range(3):
            over = layer(x, a2a_ffn_overlap_degree=degree)
            torch.cuda.synchronize()
            assert torch.equal(chunked if degree == 3 else plain, over), degree
    assert ep_native._comms and all(ep_native._comms.values()), "the native communicator must have been created"
# the variable-size exchanges on the same REAL communicator: one grouped ncclSend / ncclRecv loop (to self at world size 1)
comm = ep_native.communicator(None, x.device)
t = torch.arange(100003, device="cuda", dtype=torch.int32)
assert torch.equal(comm.all_to_all_v(t, [t.numel()], [t.numel()]), t)
assert torch.equal(comm.all_gather_v(t.to(torch.bfloat16), [t.numel()]), t.to(torch.bfloat16))
assert comm.all_to_all_v(t[:0], [0], [0]).numel() == 0
torch.cuda.synchronize()
ep_native.destroy_all()
dist.destroy_process_group()
print("NATIVE_RCCL_OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert "NATIVE_RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_overlapped_path_through_rccl_single_rank():
    """Same, with the exchange issued as a real RCCL all_to_all_single in a 1-rank process group
    (fresh process: the group must not leak into other tests)."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tutel import moe
from tutel_amd.impls import moe_layer as ml, overlap
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
layer = moe.moe_layer(gate_type={"type": "top", "k": 2, "fp32_gate": True},
                      experts={"type": "ffn", "num_experts_per_device": 8, "hidden_size_per_expert": 256,
                               "activation_fn": lambda t: torch.nn.functional.relu(t)}, model_dim=256).cuda().eval()
torch.set_default_dtype(torch.float32)
x = torch.randn(1024, 256).bfloat16().cuda()
with torch.no_grad():
    plain = layer(x, a2a_ffn_overlap_degree=1)
    ml._FORCE_OVERLAP = True
    overlap._FORCE_RCCL = True
    for _ in range(3):
        over = layer(x, a2a_ffn_overlap_degree=2)
        torch.cuda.synchronize()
        assert torch.equal(plain, over)
dist.destroy_process_group()
print("RCCL_OVERLAP_OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert "RCCL_OVERLAP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("flags", [
    "--eval --dtype=bfloat16 --batch_size=16 --num_tokens=256 --num_local_experts=64 --top=2 --num_steps=12",                       # BASELINE configs[1]
    "--eval --dtype=bfloat16 --batch_size=16 --num_tokens=256 --num_local_experts=64 --top=2 --capacity_factor=0 --megablocks_size=4 --num_steps=12",  # configs[2]
    "--dtype=float32 --batch_size=4 --num_tokens=512 --hidden_size=128 --num_local_experts=2 --top=1 --num_steps=5",                # configs[0] flags, on the GPU (training)
    "--eval --dtype=float16 --batch_size=8 --num_tokens=512 --model_dim=1024 --hidden_size=1024 --num_local_experts=16 --fp32_gate --a2a_ffn_overlap_degree=2 --num_steps=12",
    "--eval --dtype=bfloat16 --batch_size=8 --num_tokens=256 --model_dim=512 --hidden_size=1024 --num_local_experts=8 --expert_type=llama_ffn --use_tensorcore --num_steps=12",
    "--switch --eval --dtype=float16 --batch_size=8 --num_tokens=512 --model_dim=1024 --hidden_size=1024 --num_local_experts=16 --use_2dh --cap_factor=1.0 --num_steps=16",   # configs[4]'s sweep driver (helloworld_switch.py)
    "--amp --eval --dtype=float32 --batch_size=8 --num_tokens=256 --model_dim=512 --hidden_size=512 --num_local_experts=8 --num_steps=12",                                    # helloworld_amp.py
])
def test_helloworld_driver(flags):
    """The reference's benchmark/driver script surface (examples/helloworld.py flags)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "tutel_amd.examples.helloworld"] + flags.split(), cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "[Summary] Average synchronized step_time" in out, out[-3000:]
    losses = [float(l.split("loss = ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("STEP-")]
    assert len(losses) >= 5 and all(l == l for l in losses)
    if "--switch" in flags:  # the overlap degree enters the capacity alignment (moe_layer.py:298-301): losses move in the last digits
        assert max(losses) - min(losses) <= 0.02 * abs(losses[0]) and out.count("(f = 1.0, r = ") == len(losses)
    elif "--eval" in flags:
        assert len(set(losses)) == 1, "eval steps are deterministic"
    else:
        assert losses[-1] < losses[0], "SGD on the MoE layer must reduce the loss"


def test_helloworld_checkpoint_flag(tmp_path):
    """--checkpoint_path (helloworld.py:103-108,159-160): the first run trains from scratch and saves, the second
    loads the saved state and therefore starts where the first one ended."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    ck = str(tmp_path / "ckpt-{rank}-of-{size}.pt")
    flags = f"--dtype=float32 --batch_size=4 --num_tokens=128 --model_dim=128 --hidden_size=64 --num_local_experts=2 --top=1 --num_steps=10 --checkpoint_path={ck}"
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-m", "tutel_amd.examples.helloworld"] + flags.split(), cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        runs.append([float(l.split("loss = ")[1].split(",")[0]) for l in r.stdout.splitlines() if l.startswith("STEP-")])
    assert os.path.exists(ck.format(rank=0, size=1))
    assert runs[1][0] < runs[0][0] and abs(runs[1][0] - runs[0][-1]) < abs(runs[0][0] - runs[0][-1])


def test_helloworld_training_losses_match_reference():
    """Golden-loss replay in the reference's own test style (tests/test_tutel.py:94-152 over
    helloworld.py): same flags, same seeds -> the reference's printed losses (its CPU path, incl.
    BASELINE configs[0]) must come out of the HIP path: forward, backward through the dispatch
    kernels, SGD.  fp32 losses compared at the reference's own rounding (3 decimals)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(GOLD, "helloworld_losses.json")))["cases"]
    for case in cases:
        r = subprocess.run([sys.executable, "-m", "tutel_amd.examples.helloworld"] + case["flags"].split(), cwd=root,
                           env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=600)
        out = r.stdout + r.stderr
        assert r.returncode == 0, out[-3000:]
        got = [float(l.split("loss = ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("STEP-")]
        want = case["losses"]
        assert len(got) == len(want)
        assert [round(g, 3) for g in got] == [round(w, 3) for w in want] or \
            all(abs(g - w) <= 2e-3 for g, w in zip(got, want)), (got, want)


def _baseline_cases():
    import json
    return json.load(open(os.path.join(GOLD, "reference_baseline_losses.json")))["cases"]


@pytest.mark.parametrize("case", _baseline_cases(), ids=lambda c: "top%d_%s_e%d" % (c["top"], c["dtype"], c["num_local_experts"]))
def test_helloworld_replays_the_reference_test_baseline(case):
    """The reference's own golden-loss regression (tests/test_tutel.py:94-152 against tests/test_baseline.json; head of the
    file committed as tests/golden/reference_baseline_losses.json): helloworld with the flags test_tutel.py:42 builds, TRAINING
    (forward, backward through the HIP dispatch kernels, SGD) on the MI355X.  Compared at the reference test's rounding --
    3 decimals for fp32, 1 decimal otherwise; fp16: the first 2 losses, as the reference test does (:100-104) -- allowing one
    unit of that rounding (the file was written by other GPUs' GEMMs), over the first 16 steps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps = 16
    r = subprocess.run([sys.executable, "-m", "tutel_amd.examples.helloworld", "--num_steps", str(steps)] + case["flags"].split(), cwd=root,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    got = [float(l.split("loss = ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("STEP-")]
    want = [float(v) for v in case["losses"][:steps]]
    assert len(got) == steps
    if case["dtype"] == "float32":
        assert all(abs(g - w) <= 1.5e-3 for g, w in zip(got, want)), (got, want)
    elif case["dtype"] == "float16":
        assert [round(g, 1) for g in got[:2]] == [round(w, 1) for w in want[:2]] or all(abs(g - w) <= 0.06 for g, w in zip(got[:2], want[:2])), (got, want)
        assert all(abs(g - w) <= 0.25 for g, w in zip(got, want)), (got, want)   # 16 fp16 training steps on a different GPU's GEMMs
    else:   # float64 (batch_size 1: the loss collapses to ~0 within a few steps)
        assert all(abs(g - w) <= 1e-4 * max(1.0, abs(w)) for g, w in zip(got, want)), (got, want)


def test_forward_is_hip_graph_capturable_raw(oracle):
    """With capacity_factor > 0 the forward has no host synchronisation, so the whole layer (HIP
    kernels launched through the C ABI on the capturing stream + the hipBLASLt gate GEMM) can be
    captured into a HIP graph and replayed: zero host cost per step."""
    T, M, H, E, k = 2048, 512, 512, 16, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=torch.bfloat16, seed=9)
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
    xs = x.cuda()
    with torch.no_grad():
        want = layer(xs).clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                layer(xs)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = layer(xs)
            xs.copy_(x.cuda())  # same input buffer, refreshed contents
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_batch_prioritized_routing_vs_reference_fixture():
    """batch_prioritized_routing=True: locations are ranks in the order of
    descending max-score; idx / loc / gates / capacity equal the reference's.  (The reference's
    dispatch_count under BPR is the last token's row, not per-expert totals -- SURVEY section 7
    quirk 8; here it stays the true totals.)"""
    from tutel import moe
    z = np.load(os.path.join(GOLD, "bpr_routing.npz"))
    for tag in ("a", "b", "c"):
        T, E, k, cap = [int(v) for v in z[f"meta_{tag}"]]
        scores = torch.from_numpy(z[f"scores_{tag}"]).cuda()
        crit, _ = moe.top_k_routing(scores, k, capacity_factor=float(z[f"cf_{tag}"][0]), batch_prioritized_routing=True)
        assert crit[4] == cap
        assert torch.equal(torch.stack(crit[1]).cpu(), torch.from_numpy(z[f"idx_{tag}"]))
        assert torch.equal(torch.stack(crit[2]).cpu(), torch.from_numpy(z[f"loc_{tag}"])), tag
        assert torch.equal(torch.stack(crit[3]).cpu(), torch.from_numpy(z[f"gates_{tag}"]))
        assert int(crit[5].sum()) == k * T
        x = torch.randn(T, 32, device="cuda")
        y = moe.fast_decode(moe.fast_encode(x, crit), crit)
        kept = (torch.stack(crit[2]) < cap)
        w = (torch.stack(crit[3]) * kept).sum(0)
        torch.testing.assert_close(y, x * w.unsqueeze(1), rtol=1e-5, atol=1e-6)


def test_graphed_forward_wrapper(oracle):
    from tutel_amd.impls.graph import GraphedForward
    T, M, H, E, k = 1024, 256, 256, 8, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=torch.bfloat16, seed=10)
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
    xs = x.cuda()
    with torch.no_grad():
        want = layer(xs).clone()
        want2 = layer(xs * 2).clone()
    g = GraphedForward(layer, xs)
    assert torch.equal(g(xs), want)
    assert torch.equal(g(xs * 2), want2) and torch.equal(g(xs), want)
    with pytest.raises(ValueError):
        GraphedForward(layer, xs, capacity_factor=0.0)


def test_graph_replay_equals_eager_at_bench_config(oracle):
    """`bench.py --graph`: the captured forward at BASELINE configs[1] (bf16 gate, as benched) returns the
    eager forward's bits, replay after replay."""
    from tutel_amd.impls.graph import GraphedForward
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=torch.bfloat16, seed=12)
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
    xs = x.cuda().view(16, 256, M)
    with torch.no_grad():
        want = layer(xs).clone()
        want_neg = layer(-xs).clone()
    g = GraphedForward(layer, xs)
    for _ in range(3):
        assert torch.equal(g(xs), want)
        assert torch.equal(g(-xs), want_neg)


# ---- cosine gate + SwiGLU (llama_ffn) expert ---------------------------------
def make_ext_layer(M, H, E, P, k, cf, dtype, fp32_gate, tensors):
    from tutel import moe
    _, pw, pb, sim, temp, w1, w2, w3 = tensors
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = moe.moe_layer(
            gate_type={"type": "cosine_top", "k": k, "fp32_gate": fp32_gate, "capacity_factor": cf, "proj_dim": P},
            experts={"type": "llama_ffn", "num_experts_per_device": E, "hidden_size_per_expert": H}, model_dim=M)
    finally:
        torch.set_default_dtype(old)
    g = layer.gates[0]
    with torch.no_grad():
        g.cosine_projector.weight.copy_(pw); g.cosine_projector.bias.copy_(pb)
        g.sim_matrix.copy_(sim); g.temperature.copy_(temp)
        layer.experts.W_fc1.copy_(w1.reshape(-1)); layer.experts.W_fc2.copy_(w2.reshape(-1))
        layer.experts.W_fc3.copy_(w3.reshape(-1))
    return layer.cuda().eval()


EXT = sorted(glob.glob(os.path.join(GOLD, "ext_*.npz")))


@pytest.mark.parametrize("path", EXT, ids=lambda p: os.path.basename(p)[4:-4])
def test_cosine_gate_llama_expert_vs_reference_fixture(oracle, path):
    z = np.load(path)
    T, M, H, E, P, k, fp32_gate, seed = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    tensors = oracle.make_problem_ext(T, M, H, E, P, dtype=dtype, seed=seed)
    layer = make_ext_layer(M, H, E, P, k, cf, dtype, bool(fp32_gate), tensors)
    with torch.no_grad():
        x = tensors[0].cuda()
        logits = layer.gates[0](x)
        y = layer(x)
    gdt = torch.float32 if fp32_gate else dtype
    torch.testing.assert_close(logits.cpu(), _t(z["logits"], gdt), rtol=1e-5, atol=1e-6)
    # routing kernels on the reference's own scores: assignment bit-exact
    from tutel import moe
    crit, _ = moe.top_k_routing(torch.softmax(_t(z["logits"], gdt), dim=1).cuda(), k, capacity_factor=cf)
    assert torch.equal(torch.stack(crit[1]).cpu(), torch.from_numpy(z["idx"]))
    assert torch.equal(torch.stack(crit[2]).cpu(), torch.from_numpy(z["loc"]))
    assert crit[4] == int(z["capacity"][0])
    _close(y, _t(z["y"], dtype), dtype, vs_lowprec_reference=True)
    assert abs(float(y.l_aux) - float(z["l_aux"][0])) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_llama_expert_fused_glu_gemm_vs_oracle(oracle, dtype):
    """The three-launch SwiGLU path (silu fused into W_fc1's GEMM, the gating product into
    W_fc2's) against the fp32-accumulating oracle, at a shape with partial tiles."""
    from tutel_amd.experts.llama_ffn import LlamaFFNNetwork
    E, R, M, H = 6, 200, 256, 320
    g = torch.Generator().manual_seed(3)
    x = (torch.randn([E, R, M], generator=g) * 0.5).to(dtype)  # keeps |y| < 1: the fp16 bar is absolute
    net = LlamaFFNNetwork(M, H, E, 1).to(dtype)
    w1 = (torch.randn([E, M, H], generator=g) / M ** 0.5).to(dtype)
    w2 = (torch.randn([E, M, H], generator=g) / M ** 0.5).to(dtype)
    w3 = (torch.randn([E, H, M], generator=g) / H ** 0.5).to(dtype)
    with torch.no_grad():
        net.W_fc1.copy_(w1.reshape(-1)); net.W_fc2.copy_(w2.reshape(-1)); net.W_fc3.copy_(w3.reshape(-1))
    net = net.cuda().eval()

    class Ctx:
        group = None
    with torch.no_grad():
        assert net.can_fuse(x.cuda(), Ctx)
        y = net(x.cuda(), Ctx)
    ref = oracle.expert_llama_ffn(x, w1, w2, w3, accum_fp32=True)
    _close(y, ref, dtype)
    # and the ATen path (taken under autograd) agrees with the reference formula op for op
    y2 = net(x.cuda(), Ctx)
    assert y2.requires_grad
    _close(y2.detach(), oracle.expert_llama_ffn(x, w1, w2, w3), dtype, vs_lowprec_reference=True)


@pytest.mark.parametrize("shape", [(1024, 512, 256, 32, 2), (4096, 2048, 2048, 64, 2)], ids=["small", "headline"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_low_precision_gate_layer_vs_oracle_on_its_own_scores(oracle, dtype, shape):
    """bf16 / fp16 gate (`fp32_gate=False`, the reference default): logits come from the library GEMM
    in the gate dtype; the oracle is given the scores the routing kernel derived from them (softmax is
    not bit-specified across exp implementations), so token -> expert assignment must agree exactly
    and the output within the dtype's bar."""
    from tutel_amd import ops
    T, M, H, E, k = shape   # "headline" = BASELINE configs[1] exactly as bench.py runs it (fp32_gate=False)
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=11)
    layer = make_layer(M, H, E, k, 1.0, dtype, weights).eval()
    layer._keep_routing, layer.last_logits = True, None
    with torch.no_grad():
        xd = x.cuda()
        y = layer(xd)
        native = layer.last_logits is not None   # the projection ran inside the native call (csrc/gate_proj.hip): route the oracle on ITS logits
        logits = layer.last_logits if native else layer.gates[0](xd)
        lib_logits = layer.gates[0](xd)
    assert logits.dtype == dtype and native
    # the in-call projection against the library GEMM: both accumulate in fp32 and round once, in different orders
    assert float((logits.float() - lib_logits.float()).abs().max()) <= (2 ** -6 if dtype == torch.bfloat16 else 2 ** -9) * max(1.0, float(lib_logits.float().abs().max()))
    scores = ops.gate_topk(logits, k, apply_softmax=True, want_scores=True)[3].cpu()
    sref = torch.softmax(logits.float(), dim=1).cpu()
    assert float((scores.float() - sref).abs().max()) <= (2 ** -8 if dtype == torch.bfloat16 else 2 ** -11)
    wg, w1, b1, w2, b2 = weights
    crit, lo = oracle.extract_critical(scores, k, 1.0)
    enc = oracle.fast_encode(x, crit)
    yo = oracle.fast_decode(oracle.expert_ffn(enc, w1, b1, w2, b2, accum_fp32=True), crit)
    assert torch.equal(layer.dispatch_count.cpu(), crit[5])
    _close(y, yo, dtype)
    assert abs(float(y.l_aux) - float(lo)) <= 1e-2


def test_eval_weight_prelayout_tracks_weight_updates(oracle):
    """The eval path keeps a k-major copy of fc2's weights (experts/ffn.py KMajorCache).  It must follow
    every in-place update of the parameter, equal the stored-layout path bit for bit, and be dropped by
    train() / invalidate_prepacked()."""
    T, M, H, E, k = 512, 128, 192, 4, 2
    dtype = torch.bfloat16
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=2)
    layer = make_layer(M, H, E, k, 1.0, dtype, weights, gate={"fp32_gate": True}).eval()
    xd = x.cuda()
    with torch.no_grad():
        y1 = layer(xd)
        assert "fc2" in layer.experts._kmajor._store
        layer.train()          # training-mode module: stored layout, no copy
        assert not layer.experts._kmajor._store
        y1t = layer(xd)
        assert torch.equal(y1, y1t), "k-major copy and stored layout give identical results"
        layer.eval()
        layer.experts.batched_fc2_w.mul_(2)      # in-place update bumps the version counter
        y2 = layer(xd)
        b2 = layer.experts.batched_fc2_bias.float().mean()
        assert not torch.equal(y1, y2)
        wg, w1, b1, w2, b2_ = weights
        yo, *_ = oracle.moe_forward(x, wg, w1, b1, (w2.float() * 2).to(dtype), b2_, top_k=k, fp32_gate=True, accum_fp32=True)
        _close(y2, yo, dtype)
        layer.experts.batched_fc2_w.data.mul_(0.5)   # bypasses the version counter ...
        layer.experts.invalidate_prepacked()         # ... so the documented call is needed
        assert torch.equal(layer(xd), y1)


def test_inference_mode_equals_no_grad(oracle):
    T, M, H, E, k = 512, 128, 128, 8, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=torch.bfloat16, seed=4)
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
    with torch.no_grad():
        y0 = layer(x.cuda())
    with torch.inference_mode():
        y1 = layer(x.cuda())
        crit, _ = __import__("tutel").moe.top_k_routing(torch.softmax(torch.randn(64, 8, device="cuda"), 1), 2)
    assert torch.equal(y0, y1) and float(y0.l_aux) == float(y1.l_aux)


def test_variable_token_counts_reuse_workspaces(oracle):
    """20 forwards with 20 different token counts (serving): at most 3 workspace allocations (LRU over size buckets), and
    every output equals the one a fresh layer computes for that batch alone."""
    import random
    M, H, E, k = 256, 256, 16, 2
    x, *weights = oracle.make_problem(4096, M, H, E, dtype=torch.bfloat16, seed=31)
    layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
    counts = random.Random(2).sample(range(1000, 4097), 20)
    xd = x.cuda()
    outs = {}
    with torch.no_grad():
        for T in counts:
            outs[T] = layer(xd[:T]).clone()
    assert layer._ep_workspace_allocations <= 3, layer._ep_workspace_allocations
    for T in counts[:4] + [max(counts), min(counts)]:
        fresh = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights).eval()
        with torch.no_grad():
            assert torch.equal(fresh(xd[:T]), outs[T]), T
    yo, *_ = oracle.moe_forward(x[:counts[0]], *weights, top_k=k, accum_fp32=True)
    _close(outs[counts[0]], yo, torch.bfloat16)


def test_gate_is_projected_once_per_forward(oracle):
    """the one-call path used to run gate(x) before its last eligibility checks, and the general path ran it
    again on a bail-out (hooks fired twice, the projection ran twice).  One projection per forward on every path."""
    M, H, E, k = 256, 256, 8, 2
    x, *weights = oracle.make_problem(1024, M, H, E, dtype=torch.bfloat16, seed=32)
    xd = x.cuda()
    for cfg in ("inference", "trainable_router_frozen_experts", "bpr", "load_importance", "training"):
        kw = {}
        if cfg == "bpr":
            kw["batch_prioritized_routing"] = True
        if cfg == "load_importance":
            kw.update(is_gshard_loss=False, gate={"gate_noise": 1.0})
        layer = make_layer(M, H, E, k, 1.0, torch.bfloat16, weights, **kw)
        layer.train(cfg == "training")
        calls = []
        layer.gates[0].register_forward_hook(lambda *a: calls.append(1))
        if cfg == "trainable_router_frozen_experts":
            for p in layer.experts.parameters():
                p.requires_grad_(False)
            y = layer(xd)               # grad enabled: logits require grad -> the one-call path does not apply
            assert y.requires_grad
        elif cfg == "training":
            y = layer(xd)
        else:
            with torch.no_grad():
                y = layer(xd)
        assert calls == [1], (cfg, calls)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_training_forward_and_data_gradients_on_the_mfma_gemm(oracle, monkeypatch, dtype):
    """Round 3: with autograd the bf16 / fp16 ReLU experts run forward and data-gradient GEMMs on the grouped MFMA kernels
    (experts/ffn.py::_FFNTrain; weight gradients on ATen).  Against (a) the same layer on the ATen path (the reference's op
    sequence) and (b) an fp32 autograd reference of the same layer: outputs and ALL gradients, at the dtype's bar."""
    from tutel_amd import ops
    from tutel_amd.experts import ffn
    T, M, H, E, k = 1024, 256, 384, 8, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=41)

    def run(fused, dt):
        monkeypatch.setattr(ffn, "_TRAIN_FUSED", fused)
        w = [t.to(dt) for t in weights]
        layer = make_layer(M, H, E, k, 1.0, dt, w, gate={"fp32_gate": True}).train()
        xin = x.to(dt).cuda().requires_grad_(True)
        calls = []
        real = ops.expert_gemm
        monkeypatch.setattr(ops, "expert_gemm", lambda *a, **kw: calls.append(1) or real(*a, **kw))
        y = layer(xin)
        loss = (y.float() * torch.linspace(-1, 1, M, device="cuda")).sum() + y.l_aux.float()
        loss.backward()
        monkeypatch.setattr(ops, "expert_gemm", real)
        ex = layer.experts
        return (y.detach().float().cpu(), [g.float().cpu() for g in (xin.grad, ex.batched_fc1_w.grad, ex.batched_fc1_bias.grad,
                                                                     ex.batched_fc2_w.grad, ex.batched_fc2_bias.grad,
                                                                     layer.gates[0].wg.weight.grad)], len(calls))
    y_f, g_f, n_f = run(True, dtype)
    y_a, g_a, n_a = run(False, dtype)
    y_r, g_r, _ = run(False, torch.float32)
    assert n_f == 4 and n_a == 0, (n_f, n_a)     # 2 forward + 2 data-gradient launches; none on the ATen path
    eps = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    names = ("x", "fc1_w", "fc1_bias", "fc2_w", "fc2_bias", "gate_w")
    for tag, a, b, mult in [("y vs aten", y_f, y_a, 4), ("y vs fp32", y_f, y_r, 4)] + \
            [(f"d{n} vs aten", a, b, 8) for n, a, b in zip(names, g_f, g_a)] + [(f"d{n} vs fp32", a, b, 8) for n, a, b in zip(names, g_f, g_r)]:
        # two bars: the whole tensor in the Frobenius norm at a few ulps, and every element at a looser one -- ReLU's derivative
        # is discontinuous, so a pre-activation within rounding of zero flips its mask between ANY two low-precision
        # implementations (and against fp32), which moves single elements of the gradients by a whole term
        scale = float(b.abs().max())
        fro = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert fro <= mult * eps, (tag, "relative Frobenius error", fro)
        assert float((a - b).abs().max()) <= 12 * mult * eps * scale + 1e-6, (tag, float((a - b).abs().max()), scale)


def test_gate_projection_inside_the_native_call(oracle, monkeypatch):
    """Round 5: for a plain 16-bit linear gate the one-call path projects the logits itself (csrc/gate_proj.hip + the top-k kernel
    adding the split-K partial sums) instead of F.linear.  The oracle, routed on the logits the call returned, must reproduce the
    assignment exactly and the output within the dtype's bar; switching the feature off (or hooking the gate) takes the library
    projection and lands within the same bar; the logits of the two projections agree to an ulp of the dtype."""
    from tutel_amd import ops
    from tutel_amd.impls import moe_layer as ML
    T, M, H, E, k = 2048, 512, 256, 16, 2
    for dtype in (torch.bfloat16, torch.float16):
        x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=41)
        wg, w1, b1, w2, b2 = weights
        layer = make_layer(M, H, E, k, 1.0, dtype, weights).eval()
        layer._keep_routing, layer.last_logits = True, None
        xd = x.cuda()
        with torch.no_grad():
            y = layer(xd)
        assert layer.last_logits is not None, "the projection ran inside the native call"
        logits = layer.last_logits.clone()
        idx, loc = [t.cpu() for t in layer.last_routing]
        scores = ops.gate_topk(logits, k, apply_softmax=True, want_scores=True)[3].cpu()
        crit, lo = oracle.extract_critical(scores, k, 1.0)
        assert torch.equal(idx, torch.stack([t.to(torch.int32) for t in crit[1]])) and torch.equal(loc, torch.stack([t.to(torch.int32) for t in crit[2]]))
        assert torch.equal(layer.dispatch_count.cpu(), crit[5])
        yo = oracle.fast_decode(oracle.expert_ffn(oracle.fast_encode(x, crit), w1, b1, w2, b2, accum_fp32=True), crit)
        _close(y, yo, dtype)
        assert abs(float(y.l_aux) - float(lo)) <= 1e-2
        with torch.no_grad():
            lib_logits = layer.gates[0](xd)
        ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
        assert float((logits.float() - lib_logits.float()).abs().max()) <= ulp * max(1.0, float(lib_logits.float().abs().max()))
        # off: the library projection, same bar against the oracle on ITS logits
        monkeypatch.setattr(ML, "_NATIVE_GATE", False)
        layer.last_logits = None
        with torch.no_grad():
            y2 = layer(xd)
        assert layer.last_logits is None
        crit2, _ = oracle.extract_critical(ops.gate_topk(lib_logits, k, apply_softmax=True, want_scores=True)[3].cpu(), k, 1.0)
        assert torch.equal(layer.dispatch_count.cpu(), crit2[5])
        monkeypatch.setattr(ML, "_NATIVE_GATE", True)
        # a hook on the gate must keep firing: the projection stays outside
        calls = []
        h = layer.gates[0].register_forward_hook(lambda *a: calls.append(1))
        with torch.no_grad():
            y3 = layer(xd)
        h.remove()
        assert calls == [1] and torch.equal(y3, y2)
        # HIP-graph replay of the one-call path with the projection inside: equal to eager, replay after replay
        from tutel_amd.impls.graph import GraphedForward
        with torch.no_grad():
            gf = GraphedForward(layer, xd)
            for _ in range(3):
                assert torch.equal(gf(xd), y)
            x2 = torch.roll(xd, 1, 0)
            assert torch.equal(gf(x2), layer(x2))


FL_SHAPES = [  # (T, M, H, E, k, capacity_factor): capacity <= 128 rows per expert and k * T <= 15360 entries -- where the fused-location kernel applies
    (4096, 2048, 2048, 64, 2, 1.0),   # the headline: the ring kernel is the automatic choice
    (1000, 256, 512, 16, 2, 1.0), (330, 128, 256, 8, 3, 1.0), (777, 64, 256, 128, 1, 1.0), (2000, 192, 320, 16, 2, 0.5), (64, 64, 256, 4, 2, 1.0),
    (7680, 128, 256, 128, 2, 1.0),
    # NOT eligible (capacity 512 rows per expert; k * T = 16384 entries): the location kernel must run, whatever the option says --
    # an eligibility query that fell through into a launch once made the one-call path skip the location kernel here
    (4096, 256, 512, 16, 2, 1.0), (8192, 128, 256, 128, 2, 1.0),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", FL_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fused_location_keeps_every_bit(oracle, shape, dtype):
    """Round 5: on the single-rank one-call path the locations are computed INSIDE the first expert GEMM (every block ranks its own
    expert's (choice, token) entries; no tutel_amd_compute_location launch; dispatch_count / loss in an extra block of the decode
    launch).  Against the unfused path (TUTEL_OPT_FUSED_LOCATION = 0) every output must keep its bits -- y, l_aux, dispatch_count, idx,
    loc, the slot map -- and idx / loc / counts must equal the oracle's on the scores the kernel derived.  The small shapes force the
    128 x 256 ring kernel (TUTEL_OPT_GEMM_IMPL = 4), which is the only one that has the fused form."""
    from tutel_amd import _lib, ops
    T, M, H, E, k, cf = shape
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=51)
    layer = make_layer(M, H, E, k, cf, dtype, weights).eval()
    layer._keep_routing, layer.last_logits = True, None
    xd = x.cuda()
    got = {}
    try:
        ops.set_option(_lib.OPT_GEMM_IMPL, 4)
        for mode in (0, -1):
            ops.set_option(_lib.OPT_FUSED_LOCATION, mode)
            layer.__dict__.pop("_ep_workspaces", None)
            ops.stage_timing(1)
            with torch.no_grad():
                y = layer(xd)
            torch.cuda.synchronize()
            rep = ops.stage_report()
            ops.stage_timing(0)
            ws = list(layer._ep_workspaces.values())[0]
            C = int(layer.protected_shape[1])
            got[mode] = dict(y=y.clone(), l_aux=y.l_aux.clone(), cnt=layer.dispatch_count.clone(), idx=layer.last_routing[0].clone(),
                             loc=layer.last_routing[1].clone(), smap=ws.slot_map[:E * C].clone(),
                             logits=(layer.last_logits if layer.last_logits is not None else layer.gates[0](xd)).clone(),
                             location_launches=rep["location"][1])
    finally:
        ops.stage_timing(0)
        ops.set_option(_lib.OPT_FUSED_LOCATION, -1)
        ops.set_option(_lib.OPT_GEMM_IMPL, -1)
    # (M >= 128: the fused form's prologue counts the weight DMA of TWO K-tiles behind the idx bytes -- ADVICE r5: with M = 64 there is
    # one, the wait would not cover the scan's input; such shapes take the location kernel, as the (777, 64, ...) case checks)
    eligible = k * ((T + E - 1) // E) * cf <= 128 and k * T <= 15360 and M >= 128
    assert got[0]["location_launches"] == 1 and got[-1]["location_launches"] == (0 if eligible else 1), "the fused path runs exactly where it applies"
    for name in ("y", "l_aux", "cnt", "idx", "loc", "smap", "logits"):
        assert torch.equal(got[0][name], got[-1][name]), name
    scores = ops.gate_topk(got[-1]["logits"], k, apply_softmax=True, want_scores=True)[3].cpu()
    crit, _ = oracle.extract_critical(scores, k, cf)
    assert torch.equal(got[-1]["idx"].cpu(), torch.stack([t.to(torch.int32) for t in crit[1]]))
    assert torch.equal(got[-1]["loc"].cpu(), torch.stack([t.to(torch.int32) for t in crit[2]]))
    assert torch.equal(got[-1]["cnt"].cpu(), crit[5])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_layer_input_that_starts_at_an_odd_element(oracle, dtype):
    """a layer input that is a contiguous slice of a larger buffer starting at an odd element (data_ptr % 16 != 0): the forward copies it
    to an aligned allocation (moe_layer.py) instead of failing in the C ABI -- the same bits as from an aligned input, on the one-call
    path (bf16) and on the ATen-expert path (fp32)."""
    T, M, H, E, k = 1000, 128, 256, 16, 2
    x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=77)
    layer = make_layer(M, H, E, k, 1.0, dtype, weights).eval()
    big = torch.empty(T * M + 8, dtype=dtype, device="cuda")
    with torch.no_grad():
        want = layer(x.cuda())
        for off in (1, 3, 5):
            xv = big[off:off + T * M].view(T, M)
            xv.copy_(x)
            assert xv.data_ptr() % 16 != 0
            y = layer(xv)
            assert torch.equal(y, want) and float(y.l_aux) == float(want.l_aux), off
