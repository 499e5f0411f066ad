"""Generates tests/golden/*.npz by running the REFERENCE itself (microsoft/tutel, python package
from /root/reference + its own C++ CPU kernels compiled into oracle/_ref/ by oracle/Makefile).
Only runs in the build container (the GPU box has no /root/reference); the fixtures are committed.

    make -C oracle && python tests/golden/make_golden.py

Inputs are regenerated from seeds by oracle.moe_oracle.make_problem on any box (torch CPU
generator, deterministic), so the fixtures hold the reference's OUTPUTS plus an input checksum.
With --check the script instead compares the oracle against the live reference, function by
function (used by tests/test_oracle_vs_reference.py)."""
import argparse
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TUTEL_REFERENCE", "/root/reference")
# the reference package is also called `tutel`: it must win over the repo's alias package
sys.path = [REF, os.path.join(ROOT, "oracle", "_ref")] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT] + [ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

logging.disable(logging.CRITICAL)
import tutel  # noqa: E402
assert os.path.abspath(tutel.__file__).startswith(REF), tutel.__file__
from tutel import moe as ref_moe  # noqa: E402
from tutel.impls import losses as ref_losses  # noqa: E402
from oracle import moe_oracle as O  # noqa: E402

DT = {"float32": torch.float32, "float64": torch.float64, "bfloat16": torch.bfloat16, "float16": torch.float16}

# (name, T, M, H, E, k, capacity_factor, dtype, fp32_gate, is_postscore, normalize_gate)
LAYER_CASES = [
    ("f32_k2_cf1", 512, 64, 32, 16, 2, 1.0, "float32", False, True, True),
    ("f32_k1_cf1", 512, 64, 32, 16, 1, 1.0, "float32", False, True, True),
    ("f32_k2_drop", 512, 64, 32, 16, 2, 0.5, "float32", False, True, True),
    ("f32_k2_dropless", 512, 64, 32, 16, 2, 0.0, "float32", False, True, True),
    ("f32_k2_cf2_prescore", 512, 64, 32, 16, 2, 2.0, "float32", False, False, True),
    ("f32_k4_nonorm", 300, 64, 32, 12, 4, 1.0, "float32", False, True, False),
    ("f64_k2_cf1", 256, 64, 32, 8, 2, 1.0, "float64", False, True, True),
    ("bf16_k2_fp32gate", 512, 64, 64, 16, 2, 1.0, "bfloat16", True, True, True),
    ("f16_k2_fp32gate", 512, 64, 64, 16, 2, 1.0, "float16", True, True, True),
    ("c0_plumbing", 2048, 2048, 128, 2, 1, 1.0, "float32", False, True, True),  # BASELINE configs[0] shape
]


# cosine gate + SwiGLU expert: (name, T, M, H, E, P, k, cf, dtype, fp32_gate)
EXT_CASES = [
    ("f32_cosine_llama", 512, 64, 64, 16, 32, 2, 1.0, "float32", False),
    ("bf16_cosine_llama_fp32gate", 512, 64, 64, 16, 32, 2, 1.0, "bfloat16", True),
    ("f32_cosine_llama_k1_drop", 384, 64, 64, 8, 16, 1, 0.5, "float32", False),
]


def build_reference_ext_layer(T, M, H, E, P, k, cf, dtype, fp32_gate, seed):
    x, pw, pb, sim, temp, w1, w2, w3 = O.make_problem_ext(T, M, H, E, P, dtype=dtype, seed=seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ref_moe.moe_layer(
            gate_type={"type": "cosine_top", "k": k, "fp32_gate": fp32_gate, "capacity_factor": cf, "proj_dim": P},
            experts={"type": "llama_ffn", "num_experts_per_device": E, "hidden_size_per_expert": H},
            model_dim=M)
    finally:
        torch.set_default_dtype(old)
    g = layer.gates[0]
    with torch.no_grad():
        g.cosine_projector.weight.copy_(pw); g.cosine_projector.bias.copy_(pb)
        g.sim_matrix.copy_(sim); g.temperature.copy_(temp)
        layer.experts.W_fc1.copy_(w1.reshape(-1)); layer.experts.W_fc2.copy_(w2.reshape(-1))
        layer.experts.W_fc3.copy_(w3.reshape(-1))
    layer.eval()
    return layer, (x, pw, pb, sim, temp, w1, w2, w3)


def run_ext_case(case, seed=4321):
    name, T, M, H, E, P, k, cf, dts, fp32_gate = case
    layer, tensors = build_reference_ext_layer(T, M, H, E, P, k, cf, DT[dts], fp32_gate, seed)
    x = tensors[0]
    with torch.no_grad():
        y = layer(x)
        logits = layer.gates[0](x)
        crit, _ = ref_moe.top_k_routing(torch.softmax(logits, dim=1), k, capacity_factor=cf)
    out = dict(meta=np.array([T, M, H, E, P, k, int(fp32_gate), seed], dtype=np.int64), cf=np.array([cf]),
               dtype=np.array([dts]), in_checksum=np.array([checksum(tensors)]), logits=np_(logits),
               idx=np.stack([np_(i) for i in crit[1]]), loc=np.stack([np_(i) for i in crit[2]]),
               gates=np.stack([np_(g) for g in crit[3]]), capacity=np.array([crit[4]]),
               l_aux=np.array([float(y.l_aux)]), y=np_(y))
    return name, out, (layer, x)


def oracle_ext_forward(case, seed=4321, accum_fp32=False):
    name, T, M, H, E, P, k, cf, dts, fp32_gate = case
    x, pw, pb, sim, temp, w1, w2, w3 = O.make_problem_ext(T, M, H, E, P, dtype=DT[dts], seed=seed)
    return O.moe_forward(x, None, w1, None, None, None, top_k=k, capacity_factor=cf,
                         logits_fn=lambda t: O.cosine_gate_logits(t, pw, pb, sim, temp, fp32_gate),
                         expert_fn=lambda e: O.expert_llama_ffn(e, w1, w2, w3, accum_fp32=accum_fp32))


def build_reference_layer(T, M, H, E, k, cf, dtype, fp32_gate, is_postscore, normalize_gate, seed):
    x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ref_moe.moe_layer(
            gate_type={"type": "top", "k": k, "fp32_gate": fp32_gate, "capacity_factor": cf},
            experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                     "activation_fn": lambda t: torch.nn.functional.relu(t)},
            model_dim=M, is_postscore=is_postscore, normalize_gate=normalize_gate)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        layer.gates[0].wg.weight.copy_(wg.to(layer.gates[0].wg.weight.dtype))
        layer.experts.batched_fc1_w.copy_(w1)
        layer.experts.batched_fc1_bias.copy_(b1)
        layer.experts.batched_fc2_w.copy_(w2)
        layer.experts.batched_fc2_bias.copy_(b2)
    layer.eval()
    return layer, (x, wg, w1, b1, w2, b2)


def np_(t):
    t = t.detach()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()  # raw bits
    return t.numpy()


def checksum(tensors):
    return float(sum(t.double().abs().sum() for t in tensors))


def run_layer_case(case, seed=1234):
    name, T, M, H, E, k, cf, dts, fp32_gate, post, norm = case
    dtype = DT[dts]
    layer, (x, wg, w1, b1, w2, b2) = build_reference_layer(T, M, H, E, k, cf, dtype, fp32_gate, post, norm, seed)
    with torch.no_grad():
        y = layer(x)
        l_aux = y.l_aux
        # intermediates through the reference's own low-level API
        logits = layer.gates[0](x)
        scores = torch.softmax(logits, dim=1)
        crit, l2 = ref_moe.top_k_routing(scores, k, capacity_factor=cf, normalize_gate=norm)
        enc = ref_moe.fast_encode(x.to(logits.dtype), crit, post).to(x.dtype)
        ffn = layer.experts(enc, layer)
    out = dict(
        meta=np.array([T, M, H, E, k, int(fp32_gate), int(post), int(norm), seed], dtype=np.int64),
        cf=np.array([cf]), dtype=np.array([dts]), in_checksum=np.array([checksum([x, wg, w1, b1, w2, b2])]),
        scores=np_(scores), idx=np.stack([np_(i) for i in crit[1]]), loc=np.stack([np_(i) for i in crit[2]]),
        gates=np.stack([np_(g) for g in crit[3]]), capacity=np.array([crit[4]]),
        dispatch_count=np_(crit[5].to(torch.int32)), l_aux=np.array([float(l_aux)]))
    # big outputs are stored as a row subsample + a checksum to keep the fixtures small
    stride = 1 if y.numel() <= (1 << 16) else 32
    out.update(y=np_(y[::stride].contiguous()), y_row_stride=np.array([stride]),
               y_abs_sum=np.array([float(y.double().abs().sum())]))
    if stride > 1:
        out["scores"] = np_(scores[::stride].contiguous())
    if T * M <= 512 * 64:
        out.update(encoded=np_(enc), expert_out=np_(ffn))
    return name, out, (layer, x, crit)


def headline_integer_case(seed=0):
    """BASELINE configs[1] shape, integer tensors only (T=4096, E=64, k=2, cf=1): tiny."""
    g = torch.Generator().manual_seed(seed)
    scores = torch.softmax(torch.randn([4096, 64], generator=g), dim=1)
    out = {}
    for cf in (1.0, 0.0):
        crit, l_aux = ref_moe.top_k_routing(scores, 2, capacity_factor=cf)
        tag = "cf1" if cf > 0 else "dropless"
        out[f"idx_{tag}"] = np.stack([np_(i) for i in crit[1]])
        out[f"loc_{tag}"] = np.stack([np_(i) for i in crit[2]])
        out[f"capacity_{tag}"] = np.array([crit[4]])
        out[f"count_{tag}"] = np_(crit[5].to(torch.int32))
        out[f"l_aux_{tag}"] = np.array([float(l_aux)])
    out["seed"] = np.array([seed])
    return out


def headline_low_precision_gate_case(dts, seed=11):
    """BASELINE configs[1] with the gate exactly as bench.py runs it (`fp32_gate=False`: a bf16 / fp16 nn.Linear): the reference's
    own logits, scores and routing at T = 4096, M = 2048, E = 64, k = 2.  Only the gate is built (H = 8 keeps make_problem cheap;
    the test regenerates x / wg with the same arguments)."""
    dtype = DT[dts]
    T, M, H, E, k = 4096, 2048, 8, 64, 2
    layer, (x, wg, *_rest) = build_reference_layer(T, M, H, E, k, 1.0, dtype, False, True, True, seed)
    with torch.no_grad():
        logits = layer.gates[0](x)
        assert logits.dtype == dtype
        scores = torch.softmax(logits, dim=1)
        crit, l_aux = ref_moe.top_k_routing(scores, k, capacity_factor=1.0)
    return dict(meta=np.array([T, M, H, E, k, seed], dtype=np.int64), dtype=np.array([dts]),
                in_checksum=np.array([checksum([x, wg])]), logits=np_(logits) if dtype == torch.bfloat16 else logits.view(torch.int16).numpy(),
                scores=np_(scores) if dtype == torch.bfloat16 else scores.view(torch.int16).numpy(),
                idx=np.stack([np_(i) for i in crit[1]]).astype(np.int32), loc=np.stack([np_(i) for i in crit[2]]).astype(np.int32),
                gates=np.stack([(g.view(torch.int16)).numpy() for g in crit[3]]), capacity=np.array([crit[4]]),
                dispatch_count=np_(crit[5].to(torch.int32)), l_aux=np.array([float(l_aux)]))


def headline_fp32_gate_case(dts, seed=0):
    """BASELINE configs[1] with `fp32_gate=True` (gates/top.py:7-22: the projection in fp32 whatever the experts' dtype): the
    reference's routing at T = 4096, M = 2048, E = 64, k = 2 from the LAYER's own logits (its nn.Linear gate on make_problem's tokens,
    not softmax(randn) like headline_integers.npz) -- idx / loc / counts for the element-wise test of the HIP layer at the headline
    shape.  Only the gate is built (H = 8); the test regenerates x / wg with the same arguments (make_problem draws them first)."""
    dtype = DT[dts]
    T, M, H, E, k = 4096, 2048, 8, 64, 2
    layer, (x, wg, *_rest) = build_reference_layer(T, M, H, E, k, 1.0, dtype, True, True, True, seed)
    with torch.no_grad():
        logits = layer.gates[0](x)
        assert logits.dtype == torch.float32
        scores = torch.softmax(logits, dim=1)
        crit, l_aux = ref_moe.top_k_routing(scores, k, capacity_factor=1.0)
        top3 = torch.topk(scores, k + 1, dim=1).values
    return dict(meta=np.array([T, M, H, E, k, seed], dtype=np.int64), dtype=np.array([dts]), in_checksum=np.array([checksum([x, wg])]),
                idx=np.stack([np_(i) for i in crit[1]]).astype(np.int32), loc=np.stack([np_(i) for i in crit[2]]).astype(np.int32),
                capacity=np.array([crit[4]]), dispatch_count=np_(crit[5].to(torch.int32)), l_aux=np.array([float(l_aux)]),
                # smallest relative gap between neighbouring scores among each row's three largest: rows below ~1e-6 could legitimately
                # order differently under another fp32 summation order of the gate GEMM (there are none in these fixtures)
                min_rel_gap=np.array([float(((top3[:, :-1] - top3[:, 1:]) / top3[:, :-1]).min())]))


def train_losses_case(E_loc=2, k=2, steps=4, T=1024, M=256, H=256, seed=5):
    """A short training replay in the style of the reference's golden-loss tests
    (tests/test_tutel.py:94-148 over examples/helloworld.py:126-146): fwd + bwd + SGD, fp32."""
    layer, (x, *_rest) = build_reference_layer(T, M, H, E_loc, k, 1.0, torch.float32, False, True, True, seed)
    layer.train()
    opt = torch.optim.SGD(layer.parameters(), lr=1e-2)
    xb = x.view(4, T // 4, M)
    target = torch.zeros(4, dtype=torch.long)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        out = layer(xb)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(out.sum(dim=2), dim=1), target) + 0.01 * out.l_aux
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return dict(losses=np.array(losses), meta=np.array([T, M, H, E_loc, k, steps, seed], dtype=np.int64))


# gate noise + load-importance loss (moe_layer.py:285-296, losses.py:21-42):
# (name, T, M, H, E, k, dtype, fp32_gate, gate_noise, training)
NOISY_CASES = [
    ("f32_noise1_loadimp_train", 512, 64, 32, 16, 2, "float32", False, 1.0, True),
    ("f32_noise1_loadimp_eval", 512, 64, 32, 16, 2, "float32", False, 1.0, False),   # eval: no draw, the loss still uses gate_noise
    ("bf16_noise2_loadimp_train_fp32gate", 384, 64, 64, 8, 2, "bfloat16", True, 2.0, True),
]
NOISE_SEED = 99


def run_noisy_case(case, seed=2468):
    """the reference layer with is_gshard_loss=False and gate_noise > 0; the draw of torch.randn_like(logits) is reproduced
    from the seed set right before the forward and stored, so that any implementation can be fed the same noise"""
    name, T, M, H, E, k, dts, fp32_gate, gate_noise, training = case
    x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=DT[dts], seed=seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(DT[dts])
    try:
        layer = ref_moe.moe_layer(
            gate_type={"type": "top", "k": k, "fp32_gate": fp32_gate, "gate_noise": gate_noise},
            experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                     "activation_fn": lambda t: torch.nn.functional.relu(t)},
            model_dim=M, is_gshard_loss=False)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        layer.gates[0].wg.weight.copy_(wg.to(layer.gates[0].wg.weight.dtype))
        layer.experts.batched_fc1_w.copy_(w1); layer.experts.batched_fc1_bias.copy_(b1)
        layer.experts.batched_fc2_w.copy_(w2); layer.experts.batched_fc2_bias.copy_(b2)
    layer.train(training)
    with torch.no_grad():
        logits = layer.gates[0](x)
        torch.manual_seed(NOISE_SEED)
        noise = torch.randn_like(logits)
        torch.manual_seed(NOISE_SEED)
        y = layer(x)
    out = dict(meta=np.array([T, M, H, E, k, int(fp32_gate), int(training), seed], dtype=np.int64), dtype=np.array([dts]),
               gate_noise=np.array([gate_noise]), in_checksum=np.array([checksum([x, wg, w1, b1, w2, b2])]),
               noise=np_(noise), y=np_(y), l_aux=np.array([float(y.l_aux)]), dispatch_count=np_(layer.dispatch_count.to(torch.int32)))
    return name, out, (layer, x, noise)


def oracle_noisy_forward(case, noise, seed=2468):
    name, T, M, H, E, k, dts, fp32_gate, gate_noise, training = case
    x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=DT[dts], seed=seed)
    return O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, fp32_gate=fp32_gate, noise=noise if training else None,
                         gate_noise=gate_noise, is_gshard_loss=False)


def check_oracle_against_reference():
    """Function-by-function comparison of oracle/ against the live reference."""
    bad = []

    def expect(cond, what):
        if not cond:
            bad.append(what)

    for dts in ("float32", "float64", "bfloat16", "float16"):
        dtype = DT[dts]
        for (T, E, k, cf) in [(512, 16, 2, 1.0), (512, 16, 1, 1.0), (4096, 64, 2, 1.0), (512, 16, 2, 0.0),
                              (300, 7, 3, 1.5), (512, 16, 2, -0.5), (64, 2, 2, 1.0), (1000, 130, 4, 1.0)]:
            g = torch.Generator().manual_seed(T + E + k)
            scores = torch.softmax(torch.randn([T, E], generator=g), dim=1).to(dtype)
            for norm in (True, False):
                cr, lr = ref_moe.top_k_routing(scores, k, capacity_factor=cf, normalize_gate=norm)
                co, lo = O.extract_critical(scores, k, cf, normalize_gate=norm)
                tag = f"{dts} T={T} E={E} k={k} cf={cf} norm={norm}"
                same_idx = all(torch.equal(a, b) for a, b in zip(cr[1], co[1]))
                # every dtype, exact ties of the 16-bit scores included: the oracle's top-k IS the reference's CPU torch.topk
                # (oracle/aten_topk.c restates ATen's kernel, tie order and all) -- no "downstream of the tie" escape any more
                expect(same_idx, "topk indices " + tag)
                if dtype in (torch.bfloat16, torch.float16) and T >= 512:
                    top = torch.topk(scores.float(), min(k + 1, E), dim=1).values
                    expect(bool((top[:, 1:] == top[:, :-1]).any()), "the 16-bit cases are meant to contain exact ties " + tag)
                    li = torch.stack(O.topk_indices(scores, k, tie_rule="lowest")); ri = torch.stack(cr[1])
                    rows = (li != ri).any(0)
                    s_o = scores.gather(1, li.t().long())[rows]; s_r = scores.gather(1, ri.t().long())[rows]
                    expect(torch.equal(s_o.sort(1)[0], s_r.sort(1)[0]), "the lowest-index rule differs on tied rows only " + tag)
                expect(all(torch.equal(a, b) for a, b in zip(cr[2], co[2])), "locations " + tag)
                expect(all(torch.equal(a, b) for a, b in zip(cr[3], co[3])), "gates " + tag)
                expect(cr[4] == co[4], "capacity " + tag)
                expect(torch.equal(cr[5].to(torch.int32), co[5]), "dispatch_count " + tag)
                expect(torch.equal(lr, lo), "l_aux " + tag)
            M = 48
            x = torch.randn([T, M], generator=g).to(dtype)
            y = torch.randn([E * cr[4], M], generator=g).to(dtype).view(E, -1, M)
            for post in (True, False):
                expect(torch.equal(ref_moe.fast_encode(x, cr, post), O.fast_encode(x, co, post)), f"encode post={post} " + tag)
                expect(torch.equal(ref_moe.fast_decode(y, cr, post), O.fast_decode(y, co, post)), f"decode post={post} " + tag)
        m = (torch.rand([777, 33]) < 0.2).to(torch.int64)
        from tutel.jit_kernels.gating import fast_cumsum_sub_one
        expect(torch.equal(fast_cumsum_sub_one(m).to(torch.int32), O.cumsum_sub_one(m)), "cumsum_sub_one")
    # whole layer, including the dtype chain, against the reference layer
    for case in LAYER_CASES[:9]:
        name, out, (layer, x, crit) = run_layer_case(case)
        _, T, M, H, E, k, cf, dts, fp32_gate, post, norm = case
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=DT[dts], seed=1234)
        yo, lo, co, _ = O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, capacity_factor=cf, fp32_gate=fp32_gate,
                                      normalize_gate=norm, is_postscore=post)
        with torch.no_grad():
            yr = layer(x)
        expect(torch.equal(yr, yo), f"layer output {name} maxdiff={(yr.double() - yo.double()).abs().max():.3e}")
        expect(abs(float(yr.l_aux) - float(lo)) == 0, f"layer l_aux {name}")
    for case in EXT_CASES:
        name, out, (layer, x) = run_ext_case(case)
        yo, lo, co, st = oracle_ext_forward(case)
        with torch.no_grad():
            yr = layer(x)
        expect(torch.equal(yr, yo), f"cosine+llama layer output {name} maxdiff={(yr.double() - yo.double()).abs().max():.3e}")
        expect(float(yr.l_aux) == float(lo), f"cosine+llama l_aux {name}")
    for case in NOISY_CASES:
        name, out, (layer, x, noise) = run_noisy_case(case)
        yo, lo, co, _ = oracle_noisy_forward(case, noise)
        expect(torch.equal(torch.from_numpy(out["dispatch_count"]), co[5]), f"noisy-gate routing {name}")
        yr = torch.from_numpy(out["y"]).view(torch.bfloat16) if case[6] == "bfloat16" else torch.from_numpy(out["y"])
        expect(torch.equal(yr, yo), f"noisy-gate layer output {name} maxdiff={(yr.double() - yo.double()).abs().max():.3e}")
        expect(abs(float(out["l_aux"][0]) - float(lo)) <= 1e-6 * max(1.0, abs(float(lo))), f"load_importance l_aux {name}: {float(out['l_aux'][0])} vs {float(lo)}")
    # gate gradient kernel (backward-only row)
    from tutel.impls.jit_compiler import tutel_custom_kernel as ck  # noqa
    g = torch.Generator().manual_seed(9)
    T, E, k, M = 200, 6, 2, 40
    scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
    cr, _ = ref_moe.top_k_routing(scores, k, capacity_factor=0.75)
    x, buf = torch.randn([T, M], generator=g), torch.randn([E * cr[4], M], generator=g)
    for j in range(k):
        gg = torch.empty([T])
        ck.invoke_cpu_fp32([gg, cr[1][j], cr[2][j], x, buf], [T, M, cr[4]], 2)
        expect(torch.equal(gg, O.gate_grad(x, buf, cr[1][j], cr[2][j], cr[4])), f"gate_grad j={j}")
    # the dispatcher class with MASKED tokens (idx < 0: the documented way to drop tokens through fast_dispatcher.update,
    # fast_dispatch.py:101-134; CPU kernel custom_kernel.cpp:293-312 skips them)
    g = torch.Generator().manual_seed(13)
    T, E, k, M = 300, 8, 2, 24
    scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
    cr, _ = ref_moe.top_k_routing(scores, k, capacity_factor=1.0)
    idx = [i.clone() for i in cr[1]]
    idx[0][::7] = -1
    idx[1][::5] = -1
    x = torch.randn([T, M], generator=g)
    for post in (True, False):
        d = ref_moe.fast_dispatcher(E, cr[4], M, torch.float32)
        d.update(idx, cr[2], cr[3], capacity=cr[4], is_postscore=post)
        enc_r = d.encode(x)
        dec_r = d.decode(enc_r.view(E, -1, M))
        crit_o = (E, idx, cr[2], cr[3], cr[4], cr[5])
        enc_o = O.fast_encode(x, crit_o, post)
        expect(torch.equal(enc_r.view(-1, M), enc_o.view(-1, M)), f"dispatcher encode with masked tokens post={post}")
        expect(torch.equal(dec_r.view(-1, M), O.fast_decode(enc_o, crit_o, post).view(-1, M)), f"dispatcher decode with masked tokens post={post}")
    # host-side loss functions of the product (pure torch) against the reference's
    from tutel_amd.impls import losses as my_losses
    g = torch.Generator().manual_seed(77)
    sc = torch.softmax(torch.randn([640, 24], generator=g), dim=1)
    ids = torch.topk(sc, 2, dim=1).indices
    expect(torch.equal(my_losses.gshard_loss(sc, ids), ref_losses.gshard_loss(sc, ids)), "gshard_loss (torch form)")
    logits = torch.randn([640, 24], generator=g)
    tl = logits.gather(1, ids)
    a = my_losses.load_importance_loss(torch.softmax(logits, 1), tl, 24, 0.7)
    b_ = ref_losses.load_importance_loss(torch.softmax(logits, 1), tl, 24, 0.7)
    expect(torch.allclose(a, b_, rtol=1e-6, atol=1e-7), f"load_importance_loss {float(a)} vs {float(b_)}")
    for b in bad:
        print("MISMATCH:", b)
    print("oracle-vs-reference: %d mismatches" % len(bad))
    return len(bad)


def check_product_host_logic_against_reference():
    """The drop-in surface itself, product vs live reference in ONE process: the product's layer (tutel_amd, kernels
    replaced by the oracle through tests/_cpu_ops.py -- the oracle equals the reference's CPU kernels bit for bit, see above) and the
    reference's layer, built from the same arguments: identical initial weights from the seeds, multiple gates + gate_index,
    forward-time top_k / capacity_factor overrides, reserve_dims with a custom expert module, result_func, the deprecated
    spellings -- outputs, l_aux and attributes must be EQUAL."""
    bad = []

    def expect(cond, what):
        if not cond:
            bad.append(what)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _cpu_ops
    from tutel_amd import ops as my_ops
    from tutel_amd.impls import moe_layer as my_ml
    for name in ("gate_topk", "compute_location", "slot_map", "cumsum_sub_one", "fast_encode", "fast_decode", "gate_grad"):
        setattr(my_ops, name, getattr(_cpu_ops, name))
    relu = lambda t: torch.nn.functional.relu(t)

    # (d) seeds -> identical parameters, no copying (ffn.py:39-49 RNG order, moe_layer.py:112-233 seeding)
    kw = dict(gate_type=[{"type": "top", "k": 1}, {"type": "top", "k": 2, "capacity_factor": 1.5}],
              experts={"type": "ffn", "num_experts_per_device": 4, "hidden_size_per_expert": 24, "activation_fn": relu},
              model_dim=32, seeds=(5, 6, 7), is_postscore=False)
    fresh = lambda d: {**d, "experts": dict(d["experts"]), "gate_type": [dict(g_) for g_ in d["gate_type"]] if isinstance(d["gate_type"], list) else (dict(d["gate_type"]) if isinstance(d["gate_type"], dict) else d["gate_type"])}   # (the reference pops keys out of the dicts it is given)
    ref, mine = ref_moe.moe_layer(**fresh(kw)), my_ml.MOELayer(**fresh(kw))
    rs, ms = ref.state_dict(), mine.state_dict()
    expect(sorted(rs) == sorted(ms), f"state_dict keys {sorted(set(rs) ^ set(ms))}")
    expect(all(torch.equal(rs[n], ms[n]) for n in rs if n in ms), "seeded initial parameters")
    # (a) multiple gates, gate_index, forward-time overrides
    g = torch.Generator().manual_seed(2)
    x = torch.randn([3, 50, 32], generator=g)
    ref.eval(); mine.eval()
    for kwargs in ({}, {"gate_index": 1}, {"gate_index": 1, "top_k": 1}, {"capacity_factor": 2.0}, {"gate_index": 1, "capacity_factor": 0.0},
                   {"gate_index": 0, "capacity_factor": -0.5, "top_k": 3}):
        with torch.no_grad():
            yr, ym = ref(x, **kwargs), mine(x, **kwargs)
        expect(torch.equal(yr, ym), f"forward{kwargs}: max diff {(yr - ym).abs().max():.3e}")
        expect(float(yr.l_aux) == float(ym.l_aux), f"l_aux forward{kwargs}")
        expect(torch.equal(ref.dispatch_count.to(torch.int32), mine.dispatch_count.to(torch.int32)), f"dispatch_count forward{kwargs}")
    for attr in ("num_global_experts", "num_local_experts", "sharded_count", "world_size", "valid_rs", "adaptive_degree", "model_dim"):
        expect(getattr(ref, attr) == getattr(mine, attr), f"attribute {attr}: {getattr(ref, attr)} vs {getattr(mine, attr)}")
    expect([(g_.top_k, g_.gate_noise, g_.capacity_factor) for g_ in ref.gates] == [(g_.top_k, g_.gate_noise, g_.capacity_factor) for g_ in mine.gates], "gate attributes")

    # (a') training: forward + backward through dispatch / combine (data and gate gradients) and the experts, both layers
    for post in (True, False):
        kw_t = dict(gate_type={"type": "top", "k": 2}, experts={"type": "ffn", "num_experts_per_device": 4, "hidden_size_per_expert": 24, "activation_fn": relu},
                    model_dim=32, seeds=(5, 6, 7), is_postscore=post)
        ref_t, mine_t = ref_moe.moe_layer(**fresh(kw_t)).train(), my_ml.MOELayer(**fresh(kw_t)).train()
        xt = torch.randn([100, 32], generator=g)
        grads = []
        for lay in (ref_t, mine_t):
            xi = xt.clone().requires_grad_(True)
            y = lay(xi)
            (y.square().sum() + 3.0 * y.l_aux).backward()
            grads.append([xi.grad] + [p.grad for _, p in sorted(lay.named_parameters()) if p.grad is not None])
        # (the router's gradient comes out of differentiable torch forms of the same formulas on both sides, in another op order: fp32 rounding)
        expect(len(grads[0]) == len(grads[1]) and all(torch.allclose(a, b, rtol=1e-5, atol=1e-5 * max(1.0, float(b.abs().max()))) for a, b in zip(*grads)),
               f"training gradients is_postscore={post}: " + str([float((a - b).abs().max()) for a, b in zip(*grads)]))
    # (b) reserve_dims = 2 with a custom expert module, result_func, scan_expert_func
    class Scale(torch.nn.Module):   # called as experts(x[E_loc, R, a, b], ctx) (moe_layer.py:250-253)
        def __init__(self, model_dim, num_experts_per_device, sharded_count, **kw_):
            super().__init__()
            torch.manual_seed(3)
            self.s = torch.nn.Parameter(torch.rand(num_experts_per_device, 1, 1, 1))

        def forward(self, x, ctx):
            assert x.dim() == 4
            return x * self.s + 1.0
    seen_r, seen_m = [], []
    kw = dict(gate_type={"type": "top", "k": 2}, experts={"type": "custom", "module": Scale, "num_experts_per_device": 3}, model_dim=24,
              seeds=(1, 2, 3), result_func=lambda t: t * 2)
    ref = ref_moe.moe_layer(scan_expert_func=lambda n, p: seen_r.append(n), **fresh(kw)).eval()
    mine = my_ml.MOELayer(scan_expert_func=lambda n, p: seen_m.append(n), **fresh(kw)).eval()
    x = torch.randn([2, 30, 4, 6], generator=g)
    with torch.no_grad():
        yr, ym = ref(x, reserve_dims=2), mine(x, reserve_dims=2)
    expect(yr.shape == ym.shape == x.shape and torch.equal(yr, ym), "reserve_dims = 2 + custom expert + result_func")
    expect(seen_r == seen_m and all(hasattr(p, "_tutel_expert") for p in mine.experts.parameters()), "scan_expert_func / _tutel_expert tag")
    expect(ref.protected_shape == mine.protected_shape, f"protected_shape {ref.protected_shape} vs {mine.protected_shape}")
    # (c) deprecated spellings and error behaviour
    for build in (ref_moe.moe_layer, my_ml.MOELayer):
        lay = build(gate_type="Top2Gate", experts={"type": "ffn", "count_per_node": 2, "hidden_size_per_expert": 8, "activation_fn": relu},
                    model_dim=8, pad_samples=True)
        expect(lay.gates[0].top_k == 2 and lay.num_local_experts == 2, f"{build.__module__}: deprecated spellings")
        for badkw, exc in ((dict(no_such_option=1), Exception), (dict(parallel_type="nonsense"), Exception)):
            try:
                build(gate_type={"type": "top", "k": 1}, experts={"type": "ffn", "num_experts_per_device": -1 if "parallel_type" in badkw else 1,
                                                                   "hidden_size_per_expert": 8, "activation_fn": relu}, model_dim=8, **badkw)
                expect("parallel_type" in badkw, f"{build.__module__}: {badkw} must raise")   # (sharded_count == 1: any parallel_type passes)
            except exc:
                pass
    for b in bad:
        print("MISMATCH:", b)
    print("product-vs-reference host logic: %d mismatches" % len(bad))
    return len(bad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only-headline-gate", action="store_true", help="(re)write only headline_gate_*.npz")
    ap.add_argument("--only-headline-fp32-gate", action="store_true", help="(re)write only headline_fp32gate_*.npz")
    args = ap.parse_args()
    if args.only_headline_gate:
        for dts in ("bfloat16", "float16"):
            out = headline_low_precision_gate_case(dts)
            np.savez_compressed(os.path.join(HERE, f"headline_gate_{dts}.npz"), **out)
            print("wrote headline_gate", dts, int(out["capacity"][0]), float(out["l_aux"][0]))
        return
    if args.only_headline_fp32_gate:
        for dts in ("bfloat16", "float16"):
            out = headline_fp32_gate_case(dts)
            np.savez_compressed(os.path.join(HERE, f"headline_fp32gate_{dts}.npz"), **out)
            print("wrote headline_fp32gate", dts, int(out["capacity"][0]), float(out["l_aux"][0]), float(out["min_rel_gap"][0]))
        return
    if args.check:
        n = check_oracle_against_reference()
        n += check_product_host_logic_against_reference()
        sys.exit(1 if n else 0)
    for case in LAYER_CASES:
        name, out, _ = run_layer_case(case)
        np.savez_compressed(os.path.join(HERE, f"layer_{name}.npz"), **out)
        print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items() if k in ("y", "idx")})
    for case in EXT_CASES:
        name, out, _ = run_ext_case(case)
        np.savez_compressed(os.path.join(HERE, f"ext_{name}.npz"), **out)
        print("wrote ext", name)
    for case in NOISY_CASES:
        name, out, _ = run_noisy_case(case)
        np.savez_compressed(os.path.join(HERE, f"noisy_{name}.npz"), **out)
        print("wrote noisy", name, float(out["l_aux"][0]))
    np.savez_compressed(os.path.join(HERE, "headline_integers.npz"), **headline_integer_case())
    for dts in ("bfloat16", "float16"):
        np.savez_compressed(os.path.join(HERE, f"headline_gate_{dts}.npz"), **headline_low_precision_gate_case(dts))
        np.savez_compressed(os.path.join(HERE, f"headline_fp32gate_{dts}.npz"), **headline_fp32_gate_case(dts))
    # batch-prioritised routing (fast_dispatch.py:138-141,155-157): tokens ranked by -max score get buckets first
    g = torch.Generator().manual_seed(31)
    bpr = {}
    for tag, (T, E, k, cf) in {"a": (1000, 16, 2, 0.5), "b": (4096, 64, 2, 1.0), "c": (300, 8, 1, 0.25)}.items():
        while True:  # the reference orders equal importance scores by an unstable argsort: keep the fixture tie-free
            scores = torch.softmax(torch.randn([T, E], generator=g), dim=1)
            if scores.max(dim=1)[0].unique().numel() == T:
                break
        crit, _ = ref_moe.top_k_routing(scores, k, capacity_factor=cf, batch_prioritized_routing=True)
        bpr[f"scores_{tag}"] = np_(scores)
        bpr[f"idx_{tag}"] = np.stack([np_(i) for i in crit[1]])
        bpr[f"loc_{tag}"] = np.stack([np_(i) for i in crit[2]])
        bpr[f"gates_{tag}"] = np.stack([np_(i) for i in crit[3]])
        bpr[f"meta_{tag}"] = np.array([T, E, k, crit[4]], dtype=np.int64)
        bpr[f"cf_{tag}"] = np.array([cf])
    np.savez_compressed(os.path.join(HERE, "bpr_routing.npz"), **bpr)
    np.savez_compressed(os.path.join(HERE, "train_losses_top2_e2.npz"), **train_losses_case(2, 2))
    np.savez_compressed(os.path.join(HERE, "train_losses_top1_e4.npz"), **train_losses_case(4, 1))
    print("done")


if __name__ == "__main__":
    main()
