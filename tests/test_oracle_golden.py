"""The oracle replayed against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs on any box: this is what pins the oracle on the GPU box,
where /root/reference does not exist."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DT = {"float32": torch.float32, "float64": torch.float64, "bfloat16": torch.bfloat16, "float16": torch.float16}


def _t(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.view(torch.bfloat16) if dtype == torch.bfloat16 else t


def _same_as_reference(got, want, dtype, what):
    """bit for bit -- with ONE host-dependent exception.  Quantities downstream of the experts' GEMM in a 16-bit dtype come from
    ATen's CPU matmul, whose accumulation order depends on the instruction set of the host (AMX / avx512_fp16 / avx512_bf16 kernels of
    oneDNN vs the AVX-512 conversion fallback): the same seeded inputs give 59 of 32 768 fp16 outputs one place apart between the
    host the fixtures were written on and a Xeon without those extensions (round 5; the reference itself, run there, moves with the
    oracle: tests/test_oracle_vs_reference.py compares the two live on whatever host this is).  Such a host gets last-place
    slack on < 2 % of the elements and says so; fp32 / fp64 and everything integer (indices, locations, counts, encoded rows) stay
    exact on every host.  (The slack is one place of the LARGEST element, not of each element: an output is a sum of k rounded
    products, and a last-place change of one summand is many places of a sum that cancelled.)"""
    if torch.equal(got, want):
        return
    assert dtype in (torch.float16, torch.bfloat16), f"{what}: must equal the reference bit for bit"
    # one place of the largest element: 1e-3 for fp16 outputs of size 1, north_star's own bar
    place = 2.0 ** (-10 if dtype == torch.float16 else -7) * float(want.double().abs().max())
    g, w = got.double(), want.double()
    differ, worst = int((g != w).sum()), float((g - w).abs().max())
    assert worst <= place and differ <= 0.02 * w.numel(), \
        f"{what}: {differ} of {w.numel()} elements differ from the reference, the worst by {worst:.3e} (one place of {dtype} here: {place:.3e})"
    import warnings
    warnings.warn(f"{what}: {differ} of {w.numel()} {dtype} elements one place off the fixture -- this host's ATen CPU GEMM accumulates "
                  f"in another order than the host the fixture was written on; the exact comparison could not be made here")


CASES = sorted(glob.glob(os.path.join(GOLD, "layer_*.npz")))


def test_fixtures_present():
    assert len(CASES) >= 10 and len(glob.glob(os.path.join(GOLD, "ext_*.npz"))) >= 3 and os.path.exists(os.path.join(GOLD, "headline_integers.npz"))
    assert len(NOISY) >= 3


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_oracle_layer_matches_reference_fixture(oracle, path):
    z = np.load(path)
    T, M, H, E, k, fp32_gate, post, norm, seed = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    chk = float(sum(t.double().abs().sum() for t in (x, wg, w1, b1, w2, b2)))
    assert chk == float(z["in_checksum"][0]), "seeded inputs must regenerate bit-identically"
    y, l_aux, crit, st = oracle.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, capacity_factor=cf,
                                            fp32_gate=bool(fp32_gate), normalize_gate=bool(norm), is_postscore=bool(post))
    stride = int(z["y_row_stride"][0])
    assert torch.equal(torch.stack(crit[1]), torch.from_numpy(z["idx"])), "expert indices"
    assert torch.equal(torch.stack(crit[2]), torch.from_numpy(z["loc"])), "locations"
    assert crit[4] == int(z["capacity"][0])
    assert torch.equal(crit[5], torch.from_numpy(z["dispatch_count"]))
    gdt = torch.float32 if fp32_gate else dtype
    assert torch.equal(torch.stack(crit[3]), _t(z["gates"], gdt)), "gates"
    assert torch.equal(st["scores"][::stride], _t(z["scores"], gdt))
    assert float(l_aux) == float(z["l_aux"][0])
    _same_as_reference(y[::stride], _t(z["y"], dtype), dtype, "layer output")
    if torch.equal(y[::stride], _t(z["y"], dtype)):
        assert float(y.double().abs().sum()) == float(z["y_abs_sum"][0])
    else:
        assert abs(float(y.double().abs().sum()) - float(z["y_abs_sum"][0])) <= 1e-4 * float(z["y_abs_sum"][0])
    if "encoded" in z.files:
        assert torch.equal(st["encoded"], _t(z["encoded"], dtype))
        _same_as_reference(st["expert_out"], _t(z["expert_out"], dtype), dtype, "expert output")


NOISY = sorted(glob.glob(os.path.join(GOLD, "noisy_*.npz")))


@pytest.mark.parametrize("path", NOISY, ids=lambda p: os.path.basename(p)[6:-4])
def test_oracle_noisy_gate_load_importance_matches_reference_fixture(oracle, path):
    """moe_layer.py:285-296 (gate noise in training, load-importance loss): the reference's output with a stored noise draw."""
    z = np.load(path)
    T, M, H, E, k, fp32_gate, training, seed = [int(v) for v in z["meta"]]
    dtype, gate_noise = DT[str(z["dtype"][0])], float(z["gate_noise"][0])
    x, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    assert float(sum(t.double().abs().sum() for t in (x, wg, w1, b1, w2, b2))) == float(z["in_checksum"][0])
    noise = torch.from_numpy(z["noise"]) if training else None
    y, l_aux, crit, _ = oracle.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, fp32_gate=bool(fp32_gate), noise=noise,
                                           gate_noise=gate_noise, is_gshard_loss=False)
    assert torch.equal(crit[5], torch.from_numpy(z["dispatch_count"]))
    _same_as_reference(y, _t(z["y"], dtype), dtype, "layer output")
    assert abs(float(l_aux) - float(z["l_aux"][0])) <= 1e-6 * max(1.0, abs(float(l_aux)))


EP = sorted(glob.glob(os.path.join(GOLD, "ep_*.npz")))


@pytest.mark.parametrize("path", EP, ids=lambda p: os.path.basename(p)[3:-4])
def test_oracle_expert_parallel_matches_reference_fixture(oracle, path):
    """moe_forward_ep (the W-rank simulation every multi-rank test is checked against) replayed against outputs of the reference
    itself running with W ranks over gloo (tests/golden/make_golden_ep.py): per rank y, the [E_loc, W*C, M] rows its experts
    received after the all-to-all, dispatch counts, l_aux -- bit for bit; incl. unequal token counts and dropless capacity."""
    z = np.load(path)
    assert len(EP) >= 6
    W, T, M, H, E_loc, k, fp32_gate = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    tokens, uneq = [int(v) for v in z["tokens"]], bool(int(z["inequivalent"][0]))
    E = E_loc * W
    xs = [oracle.make_problem(T, M, H, E, dtype=dtype, seed=100 + r)[0][:tokens[r]] for r in range(W)]
    _, wg, w1, b1, w2, b2 = oracle.make_problem(T, M, H, E, dtype=dtype, seed=7)
    parts = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(W)]
    ys, crits, recvs = oracle.moe_forward_ep(xs, wg, parts(w1), parts(b1), parts(w2), parts(b2), top_k=k, capacity_factor=cf,
                                             fp32_gate=bool(fp32_gate), inequivalent_tokens=uneq, return_expert_inputs=True)
    for r in range(W):
        _same_as_reference(ys[r], _t(z[f"y_{r}"], dtype), dtype, f"rank {r}: y")
        assert torch.equal(recvs[r], _t(z[f"recv_{r}"], dtype).reshape(recvs[r].shape)), f"rank {r}: rows after the all-to-all"
        assert torch.equal(crits[r][5], torch.from_numpy(z[f"count_{r}"])), f"rank {r}: dispatch counts"


def test_headline_integer_fixture(oracle):
    """BASELINE configs[1] shape: token->expert/slot assignment of the reference, bit-exact."""
    z = np.load(os.path.join(GOLD, "headline_integers.npz"))
    g = torch.Generator().manual_seed(int(z["seed"][0]))
    scores = torch.softmax(torch.randn([4096, 64], generator=g), dim=1)
    for tag, cf in (("cf1", 1.0), ("dropless", 0.0)):
        crit, l_aux = oracle.extract_critical(scores, 2, cf)
        assert torch.equal(torch.stack(crit[1]), torch.from_numpy(z[f"idx_{tag}"]))
        assert torch.equal(torch.stack(crit[2]), torch.from_numpy(z[f"loc_{tag}"]))
        assert crit[4] == int(z[f"capacity_{tag}"][0])
        assert torch.equal(crit[5], torch.from_numpy(z[f"count_{tag}"]))
        assert float(l_aux) == float(z[f"l_aux_{tag}"][0])


@pytest.mark.parametrize("dts", ["bfloat16", "float16"])
def test_headline_low_precision_gate_fixture(oracle, dts):
    """BASELINE configs[1] with the gate as bench.py runs it (`fp32_gate=False`): the reference's own logits / scores / routing.
    On the reference's scores the oracle's expert ids, slots and counts EQUAL the reference's on every row -- including the 89 (bf16) /
    15 (fp16) tokens whose 16-bit scores tie exactly at the k / k+1 boundary, where the answer is whatever ATen's CPU top-k leaves
    (oracle/aten_topk.c).  The lowest-index rule of rounds 1-5 (tie_rule="lowest") differs on exactly those rows."""
    dtype = getattr(torch, dts)
    z = np.load(os.path.join(GOLD, f"headline_gate_{dts}.npz"))
    T, M, H, E, k, seed = [int(v) for v in z["meta"]]
    x, wg, *_ = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    assert abs(float(z["in_checksum"][0]) - float(x.double().abs().sum() + wg.double().abs().sum())) < 1e-6 * float(z["in_checksum"][0])
    scores, idx_r = torch.from_numpy(z["scores"]).view(dtype), torch.from_numpy(z["idx"])
    crit, _ = oracle.extract_critical(scores, k, 1.0)
    assert torch.equal(torch.stack(crit[1]).to(torch.int32), idx_r)
    assert torch.equal(torch.stack(crit[2]), torch.from_numpy(z["loc"])) and torch.equal(crit[5], torch.from_numpy(z["dispatch_count"]))
    assert torch.equal(torch.stack(crit[3]).view(torch.int16), torch.from_numpy(z["gates"]))
    idx_l = torch.stack(oracle.topk_indices(scores, k, tie_rule="lowest"))
    diff = torch.nonzero((idx_l != idx_r).any(0)).flatten().tolist()
    for t in diff:
        assert all(scores[t, idx_l[j, t]] == scores[t, idx_r[j, t]] for j in range(k)), f"token {t}: not a tie"
    assert (len(diff), int((idx_l != idx_r).sum())) == ((89, 120) if dts == "bfloat16" else (15, 20))


@pytest.mark.parametrize("dts", ["bfloat16", "float16"])
def test_headline_fp32_gate_fixture(oracle, dts):
    """BASELINE configs[1] with fp32_gate=True: the reference's routing of make_problem's tokens through its own nn.Linear gate at
    T = 4096, E = 64 (the fixture the HIP layer is compared with element-wise, tests/test_layer_gpu.py) equals the oracle's."""
    dtype = getattr(torch, dts)
    z = np.load(os.path.join(GOLD, f"headline_fp32gate_{dts}.npz"))
    T, M, H, E, k, seed = [int(v) for v in z["meta"]]
    x, wg, *_ = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed)
    scores, _ = oracle.gate_scores(x, wg, True)
    crit, l_aux = oracle.extract_critical(scores, k, 1.0)
    assert torch.equal(torch.stack(crit[1]).to(torch.int32), torch.from_numpy(z["idx"])) and torch.equal(torch.stack(crit[2]), torch.from_numpy(z["loc"]))
    assert torch.equal(crit[5], torch.from_numpy(z["dispatch_count"])) and crit[4] == int(z["capacity"][0])
    assert abs(float(l_aux) - float(z["l_aux"][0])) < 1e-6 and float(z["min_rel_gap"][0]) > 1e-5


def test_aten_topk_restatement_equals_torch_topk(oracle):
    """oracle/aten_topk.c (ATen's CPU top-k + the libstdc++ routines under it, restated) against LIVE torch.topk of this box's torch --
    the op the reference calls (fast_dispatch.py:146) -- on rows made of a few distinct values, so that nearly every row ties: both
    branches (partial_sort for k * 64 <= E, nth_element + sort otherwise), every dtype the gate can have, NaNs (sorted first)."""
    g = torch.Generator().manual_seed(0)
    n = 0
    for E in list(range(1, 40)) + [63, 64, 65, 96, 127, 128, 129, 192, 256, 300, 1024]:
        for k in [1, 2, 3, 4, 8, 16]:
            if k > E:
                continue
            for levels in [1, 2, 3, 17]:
                for dt in [torch.float32, torch.bfloat16, torch.float16, torch.float64]:
                    s = (torch.randint(0, levels, (48, E), generator=g).to(torch.float32) / 8).to(dt)
                    if levels == 3:
                        s[::5, E // 2] = float("nan")
                    want = torch.topk(s, k, dim=1).indices.to(torch.int32).t()
                    assert torch.equal(torch.stack(oracle.topk_indices(s, k)), want), (E, k, levels, dt)
                    n += 1
    s = torch.softmax(torch.randn(4096, 64, generator=g), 1).bfloat16()          # the headline's kind of row
    assert torch.equal(torch.stack(oracle.topk_indices(s, 2)), torch.topk(s, 2, dim=1).indices.to(torch.int32).t())
    assert not torch.equal(torch.stack(oracle.topk_indices(s, 2, tie_rule="lowest")), torch.topk(s, 2, dim=1).indices.to(torch.int32).t())
    assert n > 3000


def test_oracle_edge_cases(oracle):
    # every token dropped for one choice (capacity 1), empty experts, k > E clamps, T < E
    scores = torch.softmax(torch.randn(5, 9), dim=1)
    crit, _ = oracle.extract_critical(scores, 20, 1.0)
    assert len(crit[1]) == 9 and crit[4] == 9 * int(1.0 * 1)
    x = torch.randn(5, 6)
    enc = oracle.fast_encode(x, crit)
    dec = oracle.fast_decode(enc, crit, is_postscore=False)
    kept = torch.stack([l < crit[4] for l in crit[2]]).sum(0).float()
    torch.testing.assert_close(dec, x * kept.unsqueeze(1), rtol=1e-6, atol=1e-6)  # 9 sequential fp32 adds
    # a2a layout round trip
    per_rank = [torch.randn(4, 3, 5) for _ in range(2)]
    back = oracle.a2a_combine(oracle.a2a_dispatch(per_rank), 3)
    assert all(torch.equal(a, b) for a, b in zip(per_rank, back))


EXT = sorted(glob.glob(os.path.join(GOLD, "ext_*.npz")))


@pytest.mark.parametrize("path", EXT, ids=lambda p: os.path.basename(p)[4:-4])
def test_oracle_cosine_gate_llama_expert_matches_reference_fixture(oracle, path):
    """cosine_top gate + llama_ffn (SwiGLU) expert, reference layer output bit for bit."""
    z = np.load(path)
    T, M, H, E, P, k, fp32_gate, seed = [int(v) for v in z["meta"]]
    dtype, cf = DT[str(z["dtype"][0])], float(z["cf"][0])
    x, pw, pb, sim, temp, w1, w2, w3 = oracle.make_problem_ext(T, M, H, E, P, dtype=dtype, seed=seed)
    chk = float(sum(t.double().abs().sum() for t in (x, pw, pb, sim, temp, w1, w2, w3)))
    assert chk == float(z["in_checksum"][0])
    gdt = torch.float32 if fp32_gate else dtype
    logits = oracle.cosine_gate_logits(x, pw, pb, sim, temp, bool(fp32_gate))
    assert torch.equal(logits, _t(z["logits"], gdt))
    y, l_aux, crit, _ = oracle.moe_forward(
        x, None, w1, None, None, None, top_k=k, capacity_factor=cf,
        logits_fn=lambda t: oracle.cosine_gate_logits(t, pw, pb, sim, temp, bool(fp32_gate)),
        expert_fn=lambda e: oracle.expert_llama_ffn(e, w1, w2, w3))
    assert torch.equal(torch.stack(crit[1]), torch.from_numpy(z["idx"]))
    assert torch.equal(torch.stack(crit[2]), torch.from_numpy(z["loc"]))
    assert torch.equal(torch.stack(crit[3]), _t(z["gates"], gdt)) and crit[4] == int(z["capacity"][0])
    assert float(l_aux) == float(z["l_aux"][0])
    _same_as_reference(y, _t(z["y"], dtype), dtype, "layer output")


# ---- the reference's OWN golden file: tests/test_baseline.json (head committed as reference_baseline_losses.json) ----
def _baseline_cases():
    import json
    return json.load(open(os.path.join(GOLD, "reference_baseline_losses.json")))["cases"]


def _ref_round(v, dtype):
    return round(float(v), 3 if "32" in dtype else 1)   # test_tutel.py:52-63,79-83


@pytest.mark.parametrize("case", _baseline_cases(), ids=lambda c: "top%d_%s_e%d" % (c["top"], c["dtype"], c["num_local_experts"]))
def test_oracle_reproduces_first_loss_of_reference_baseline(oracle, case):
    """losses[0] of each of the 9 entries is a pure FORWARD quantity of the untrained layer: helloworld's seeded weights and
    tokens through the oracle's forward, helloworld's loss, compared at the reference test's own rounding (and to 2e-5
    relative for fp32 / 1e-12 for fp64 -- the file was written by GPUs, this runs on CPU BLAS)."""
    dtype = DT[case["dtype"]]
    x, wg, w1, b1, w2, b2 = oracle.helloworld_problem(case["batch_size"], case["num_tokens"], case["model_dim"], case["hidden_size"],
                                                      case["num_local_experts"], dtype)
    with torch.no_grad():
        # 16-bit entries: fp32 accumulation, one rounding per GEMM -- the arithmetic of the GPUs that wrote the file (and minutes
        # faster on a host whose ATen has no fp16 GEMM kernel: 0.6 GFLOP/s on a Xeon without AVX512-FP16, 275 GFLOP per entry)
        y, _, crit, _ = oracle.moe_forward(x, wg, w1, b1, w2, b2, top_k=case["top"], capacity_factor=1.0,
                                           accum_fp32=dtype in (torch.float16, torch.bfloat16))
        loss = float(oracle.helloworld_loss(y))
    want = float(case["losses"][0])
    assert _ref_round(loss, case["dtype"]) == _ref_round(want, case["dtype"]), (loss, want)
    if case["dtype"] == "float32":
        assert abs(loss - want) <= 2e-5 * abs(want), (loss, want)
    elif case["dtype"] == "float64":
        assert abs(loss - want) <= 1e-12 * abs(want), (loss, want)
