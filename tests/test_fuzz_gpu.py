"""A wide seeded fuzz of the routing -> dispatch -> combine chain against the oracle: the form of
test_ops_gpu.py::test_routing_randomized_shapes_vs_oracle with expert counts up to the kernels' 4096, k up to 16, capacity alignment, fp64
scores, tie-heavy rows in every dtype -- every integer, every gate, every encoded / decoded element bit for bit.  150 cases in the default run,
1500 with --runslow; `python tests/test_fuzz_gpu.py [cases] [seed] [routing|gemm]` runs any length and writes gpurun_out/r6_<what>_fuzz_<seed>.json
(round 6: two seeds x 1500 cases are on record in profiles/)."""
import json
import os
import random
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

E_CHOICES = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 130, 192, 255, 256, 257, 300, 500, 512, 513, 1000,
             1024, 1025, 1500, 2048, 3000, 4095, 4096]
T_CHOICES = [1, 2, 5, 31, 63, 64, 65, 127, 128, 129, 300, 777, 1000, 2047, 2048, 4096, 4097, 8191, 8192, 8193, 10000, 16384, 20000]


def run_routing_fuzz(oracle, n_cases, seed, verbose=False):
    """-> list of failure descriptions (empty = every case equal)"""
    from tutel import moe
    rnd = random.Random(seed)
    bad, t0 = [], time.time()
    for case in range(n_cases):
        E = rnd.choice(E_CHOICES)
        T = rnd.choice(T_CHOICES)
        if T * E > (1 << 25):
            T = max(1, (1 << 25) // E)
        k = min(E, rnd.choice([1, 2, 2, 2, 3, 4, 5, 8, 11, 16]))
        while k * E > 8192:
            k -= 1
        cf = rnd.choice([1.0, 1.0, 0.5, 2.0, 1.25, 0.0, -0.5, 4.0, 0.1, -2.0])
        dtype = rnd.choice([torch.float32, torch.float32, torch.bfloat16, torch.bfloat16, torch.float16, torch.float64])
        norm = rnd.random() < 0.7
        align = rnd.choice([1, 1, 1, 2, 4, 32])
        g = torch.Generator().manual_seed(seed * 100003 + case)
        scale = rnd.choice([0.2, 0.5, 1.0, 3.0, 8.0])
        logits = torch.randn([T, E], generator=g, dtype=torch.float64) * scale
        if rnd.random() < 0.25:
            logits = (logits * 2).round() / 2           # tie-heavy in every dtype
        scores = torch.softmax(logits, dim=1).to(dtype)
        M = rnd.choice([8, 40, 64, 100, 256])
        post = rnd.random() < 0.7
        tag = f"case {case}: T={T} E={E} k={k} cf={cf} {dtype} norm={norm} align={align} scale={scale} M={M} post={post}"
        try:
            crit_o, l_o = oracle.extract_critical(scores, k, cf, normalize_gate=norm, alignment=align)
            crit, l_aux = moe.top_k_routing(scores.cuda(), k, capacity_factor=cf, normalize_gate=norm, alignment=align)
            assert torch.equal(torch.stack(crit[1]).cpu(), torch.stack(crit_o[1])), "idx"
            assert torch.equal(torch.stack(crit[2]).cpu(), torch.stack(crit_o[2])), "loc"
            assert crit[4] == crit_o[4], f"capacity {crit[4]} vs {crit_o[4]}"
            assert torch.equal(crit[5].cpu(), crit_o[5]), "dispatch_count"
            assert torch.equal(torch.stack(crit[3]).cpu().double(), torch.stack(crit_o[3]).double()), "gates"
            tol = 1e-5 if dtype in (torch.float32, torch.float64) else 2e-2
            assert abs(float(l_aux) - float(l_o)) <= tol * max(1.0, abs(float(l_o))), f"l_aux {float(l_aux)} vs {float(l_o)}"
            if crit[4] > 0 and dtype != torch.float64 and M * E * crit[4] < (1 << 26):
                x = torch.randn([T, M], generator=g).to(dtype)
                enc = moe.fast_encode(x.cuda(), crit, is_postscore=post)
                enc_o = oracle.fast_encode(x, crit_o, is_postscore=post)
                assert torch.equal(enc.cpu(), enc_o), "encode"
                dec = moe.fast_decode(enc, crit, is_postscore=post)
                assert torch.equal(dec.cpu(), oracle.fast_decode(enc_o, crit_o, is_postscore=post)), "decode"
        except Exception as ex:  # noqa: BLE001 -- the sweep reports every failing case
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300])
            if verbose:
                print("FAIL", bad[-1], flush=True)
        if verbose and (case + 1) % 100 == 0:
            print(f"{case + 1} cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


def run_gemm_fuzz(n_cases, seed, verbose=False):
    """random grouped-GEMM problems (rows per expert around every tile edge, ragged N, 1 .. 16 K-tiles, both weight layouts, every
    activation, with / without bias, with / without the fused gather) through a randomly FORCED kernel choice: the result must be within
    the rounding bound of an fp32 reference AND equal, bit for bit, to the automatic choice's (include/tutel_amd.h: every kernel walks K
    in the same order for a given problem).  -> list of failure descriptions"""
    import math
    from tutel_amd import ops, _lib
    rnd = random.Random(seed)
    bad, t0 = [], time.time()
    acts = {"none": lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu}
    for case in range(n_cases):
        E = rnd.choice([1, 1, 2, 3, 4, 6, 9])
        R = rnd.choice([1, 2, 7, 31, 63, 64, 65, 100, 127, 128, 129, 160, 255, 256, 257, 300, 511, 512, 513, 700])
        N = rnd.choice([8, 16, 24, 64, 72, 120, 128, 136, 192, 248, 256, 264, 328, 512, 520, 1024, 2048])
        K = 64 * rnd.choice([1, 1, 2, 2, 3, 4, 5, 8, 12, 16, 32])
        if E * R * N * K > (1 << 31):
            E = 1
        kmajor = rnd.random() < 0.7
        act = rnd.choice(["none", "relu", "relu", "gelu", "silu"])
        dtype = rnd.choice([torch.bfloat16, torch.float16])
        with_bias = rnd.random() < 0.7
        gather = kmajor and rnd.random() < 0.3
        impl, tile = rnd.choice([-1, 0, 1, 4]), rnd.choice([-1, 0, 1, 2, 3, 4])
        tag = f"gemm case {case}: E={E} R={R} N={N} K={K} kmajor={kmajor} act={act} {dtype} bias={with_bias} gather={gather} impl={impl} tile={tile}"
        g = torch.Generator().manual_seed(seed * 7919 + case)
        w = ((torch.rand([E, N, K] if kmajor else [E, K, N], generator=g) * 2 - 1) / math.sqrt(K)).to(dtype)
        bias = torch.randn([E, N], generator=g).to(dtype) if with_bias else None
        try:
            if gather:
                T = rnd.choice([1, 50, 1000])
                x = torch.randn([T, K], generator=g).to(dtype)
                smap = torch.randint(-1, 2 * T, (E * R,), generator=g, dtype=torch.int32)   # -1: empty slot; q >= T: choice q // T of token q % T
                a = torch.where((smap >= 0).unsqueeze(-1), x[(smap.clamp(min=0) % T).long()], torch.zeros([], dtype=dtype)).view(E, R, K)
                run = lambda: ops.expert_gemm_gather(x.cuda(), smap.cuda(), w.cuda(), bias.cuda() if with_bias else None, kmajor, act, R).cpu()
            else:
                a = torch.randn([E, R, K], generator=g).to(dtype)
                run = lambda: ops.expert_gemm(a.cuda(), w.cuda(), bias.cuda() if with_bias else None, kmajor, act=act).cpu()
            auto = run()
            ops.set_option(_lib.OPT_GEMM_IMPL, impl)
            ops.set_option(_lib.OPT_GEMM_TILE, tile)
            try:
                forced = run()
            finally:
                ops.set_option(_lib.OPT_GEMM_IMPL, -1)
                ops.set_option(_lib.OPT_GEMM_TILE, -1)
            ref = torch.matmul(a.float(), w.float().permute(0, 2, 1) if kmajor else w.float())
            if with_bias:
                ref = ref + bias.float().unsqueeze(1)
            ref = acts[act](ref).to(dtype).float()
            tol = dict(rtol=2 ** -7, atol=2e-3) if dtype == torch.bfloat16 else dict(rtol=2 ** -10, atol=3e-4)
            torch.testing.assert_close(auto.float(), ref, **tol)
            assert torch.equal(auto.view(torch.int16), forced.view(torch.int16)), f"forced kernel differs from the automatic one in {int((auto.view(torch.int16) != forced.view(torch.int16)).sum())} elements"
        except Exception as ex:  # noqa: BLE001
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300].replace("\n", " "))
            if verbose:
                print("FAIL", bad[-1], flush=True)
        if verbose and (case + 1) % 100 == 0:
            print(f"{case + 1} gemm cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [150, pytest.param(1500, marks=pytest.mark.slow)])
def test_routing_dispatch_combine_fuzz_vs_oracle(oracle, n_cases):
    bad = run_routing_fuzz(oracle, n_cases, seed=6060)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [120, pytest.param(1200, marks=pytest.mark.slow)])
def test_grouped_gemm_fuzz_vs_fp32_reference_and_across_kernels(n_cases):
    bad = run_gemm_fuzz(n_cases, seed=6061)
    assert not bad, "\n".join(bad[:20])


if __name__ == "__main__":
    from oracle import moe_oracle
    moe_oracle._lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 6060
    what = sys.argv[3] if len(sys.argv) > 3 else "routing"
    failed = run_gemm_fuzz(n, sd, verbose=True) if what == "gemm" else run_routing_fuzz(moe_oracle, n, sd, verbose=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"r6_{what}_fuzz_{sd}.json"), "w") as f:
        json.dump(dict(source="tests/test_fuzz_gpu.py", cases=n, seed=sd, failed=failed), f, indent=1)
    print("cases", n, "failed", len(failed))
    sys.exit(1 if failed else 0)
