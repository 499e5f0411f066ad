"""Seeded fuzzers of the single-GPU path, each a `run_*` function returning its failing cases (tests/test_ep_fuzz_one_gpu.py holds the
expert-parallel one):
  routing   routing -> dispatch -> combine against the oracle: the form of test_ops_gpu.py::test_routing_randomized_shapes_vs_oracle with
            expert counts up to the kernels' 4096, k up to 16, capacity alignment, fp64 scores, tie-heavy rows in every dtype -- every
            integer, every gate, every encoded / decoded element bit for bit; every third case also through the fused softmax + top-k
  gemm      grouped GEMM vs an fp32 reference and, bit for bit, across a randomly forced kernel choice
  layer     MOELayer forwards across the eligibility edges of the one-call path (dropless / megablocks, HIP-graph replays)
  train     training steps: output and all gradients vs the fp32 layer, bar relative to upstream's ATen op sequence
  ext       cosine top-k gate over SwiGLU experts
The default run takes 120 / 120 / 30 / 30 / 40 cases, --runslow the full-length forms; `python tests/test_fuzz_gpu.py [cases] [seed] [what]`
runs any length and writes gpurun_out/r6_<what>_fuzz_<seed>.json (round 6's records, incl. a soak with other seeds: profiles/README.md)."""
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
            if case % 3 == 0 and dtype != torch.float64:
                # the fused softmax + top-k form (what the layer launches): scores within a few ulps of softmax, routing exact on the
                # scores the kernel itself produced
                from tutel_amd import ops
                lg = logits.to(dtype)
                idx_k, gates_k, _, sc_k = ops.gate_topk(lg.cuda(), k, apply_softmax=True, normalize_gate=norm, want_scores=True)
                ref = torch.softmax(lg.float(), dim=1)
                stol = {torch.float32: 4e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dtype]   # (fp32: the order of a 4096-term sum)
                assert float((sc_k.cpu().float() - ref).abs().max()) <= stol * max(1.0, float(ref.max())) + 1e-7, "fused softmax scores"
                crit_s, _ = oracle.extract_critical(sc_k.cpu(), k, cf, normalize_gate=norm, alignment=align)
                assert torch.equal(idx_k.cpu(), torch.stack(crit_s[1]).to(torch.int32)), "fused softmax + top-k: idx"
                assert torch.equal(gates_k.cpu().double(), torch.stack(crit_s[3]).double()), "fused softmax + top-k: gates"
            # the data's dtype: the scores' own in every other case, else one of the three the dispatch kernels take (upstream casts the
            # gates to its fp32 dispatch dtype, fast_dispatch.py:94-128 -- any pairing is legal there)
            xdt = dtype if (case % 2 == 0 and dtype != torch.float64) else (torch.float32, torch.bfloat16, torch.float16)[(case // 2) % 3]
            if crit[4] > 0 and M * E * crit[4] < (1 << 26):
                x = torch.randn([T, M], generator=g).to(xdt)
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


def _close_at_scale(y, ref, dtype):
    """test_layer_gpu._close for outputs whose magnitude is not pinned below 1: every element within 2 ulps of its own magnitude plus ONE
    ulp at the tensor's scale (what a 1-ulp flip of an expert output contributes where a token's k choices cancel -- the bar of
    test_ep_ranks_one_gpu.py), and all but 0.2 % of the elements within the 2 ulps + the small absolute floor alone."""
    y, ref = y.double().cpu(), ref.double().cpu()
    eps = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}.get(dtype, 1e-5)
    floor = {torch.bfloat16: 2e-3, torch.float16: 3e-4}.get(dtype, 1e-6)
    err, scale = (y - ref).abs(), float(ref.abs().max())
    assert bool((err <= eps * ref.abs() + max(floor, eps * scale)).all()), f"max error {float(err.max()):.3e} at scale {scale:.3f}"
    over = float((err > eps * ref.abs() + floor).double().mean())
    assert over <= 2e-3, f"{over:.2%} of the elements beyond 2 ulps + {floor}"


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
        counts, row_align = None, 1
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
                if case % 4 == 1:
                    # dropless row counts (sparse_bmm_infer, custom_kernel.cpp:874-889): rows past an expert's count, rounded up to
                    # row_align, are neither computed nor written -- a sentinel must survive there, the rows before it must be right
                    row_align = (1, 4, 32)[(case // 4) % 3]
                    counts = torch.randint(0, R + 1, (E,), generator=g, dtype=torch.int32)
                    counts[case % E] = (0, R)[(case // 8) % 2]

                    def run():
                        o = torch.full([E, R, N], 3.0, dtype=dtype, device="cuda")
                        ops.expert_gemm(a.cuda(), w.cuda(), bias.cuda() if with_bias else None, kmajor, act=act, out=o, d_layout=(R * N, 0, R, N),
                                        row_counts=counts.cuda(), row_align=row_align)
                        return o.cpu()
                else:
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
            if counts is not None:
                for e in range(E):
                    n = min(R, (int(counts[e]) + row_align - 1) // row_align * row_align)
                    assert bool((auto[e, n:] == 3.0).all()), f"expert {e}: rows past the aligned count {n} were written"
                    ref[e, n:] = 3.0
            torch.testing.assert_close(auto.float(), ref, **tol)
            assert torch.equal(auto.view(torch.int16), forced.view(torch.int16)), f"forced kernel differs from the automatic one in {int((auto.view(torch.int16) != forced.view(torch.int16)).sum())} elements"
        except Exception as ex:  # noqa: BLE001
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300].replace("\n", " "))
            if verbose:
                print("FAIL", bad[-1], flush=True)
        if verbose and (case + 1) % 100 == 0:
            print(f"{case + 1} gemm cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


def run_layer_fuzz(oracle, n_cases, seed, verbose=False):
    """random single-rank MOELayer forwards (eval): the routing the forward really used (`_keep_routing`) must equal the oracle's on the
    scores the kernels derived -- ids, slots, counts, capacity, element for element -- and y the oracle's encode -> fp32-accumulating FFN
    -> decode on that routing, within the rounding bound of test_layer_gpu._close.  The shapes cross every eligibility edge of the
    one-call path (capacity 128, k * T = 15 360, model_dim 128, E = 128, dims that are no multiple of 64 -> the ATen experts)."""
    from test_layer_gpu import make_layer, _close
    from tutel_amd import ops
    rnd = random.Random(seed)
    bad, t0 = [], time.time()
    for case in range(n_cases):
        E = rnd.choice([1, 2, 3, 4, 6, 8, 16, 16, 32, 64, 64, 128, 130, 256])
        T = rnd.choice([1, 3, 64, 100, 127, 128, 129, 500, 1000, 1024, 2000, 4096, 5000, 7680, 7681, 8192])
        k = min(E, rnd.choice([1, 2, 2, 2, 3, 4]))
        M = rnd.choice([64, 128, 128, 192, 256, 256, 512, 1024, 40, 100])
        H = rnd.choice([64, 128, 256, 256, 320, 512, 1024, 72, 200])
        cf = rnd.choice([1.0, 1.0, 1.0, 0.5, 2.0, 1.25, 0.0, 0.0, -1.5])
        dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
        fp32_gate = rnd.random() < 0.5
        norm, post = rnd.random() < 0.7, rnd.random() < 0.7
        mega = rnd.choice([0, 0, 1, 2, 4]) if cf == 0.0 else 0   # dropless: the experts skip the rows past their own count (megablocks_size)
        while k * max(1, (T + E - 1) // E) * max(abs(cf), 1.0) * E * M * H > (1 << 32):   # keeps the CPU side of a case near a second
            T = max(1, T // 2)
        tag = f"layer case {case}: T={T} M={M} H={H} E={E} k={k} cf={cf} {dtype} fp32_gate={fp32_gate} norm={norm} post={post} megablocks={mega}"
        try:
            x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed * 31 + case)
            layer = make_layer(M, H, E, k, cf, dtype, weights, gate={"fp32_gate": fp32_gate}, normalize_gate=norm, is_postscore=post).eval()
            layer._keep_routing, layer.last_logits = True, None
            if case % 3 == 1:
                # an input of another dtype than the experts': cast to the parameters' dtype on the way in, the result back on the way out
                # (moe_layer.py:264-270, 359-361) -- widened for 16-bit experts, narrowed to bf16 for fp32 experts
                in_dtype = torch.float32 if dtype != torch.float32 else torch.bfloat16
                x_in = x.to(in_dtype)
                x = x_in.to(dtype)
            else:
                in_dtype, x_in = dtype, x
            xd = x_in.cuda()
            if (k * int(cf * ((T + E - 1) // E)) if cf > 0 else 1) == 0:
                # capacity 0: upstream's forward fails on an ambiguous reshape of the empty buckets (moe_layer.py:218 / fast_dispatch.py:214),
                # and so does this one -- the same RuntimeError, not a silent result
                with pytest.raises(RuntimeError, match="cannot reshape tensor of 0 elements"), torch.no_grad():
                    layer(xd)
                continue
            with torch.no_grad():
                y = layer(xd, megablocks_size=mega)
                logits = layer.last_logits if layer.last_logits is not None else layer.gates[0](xd.to(dtype))
                scores = ops.gate_topk(logits.contiguous(), k, apply_softmax=True, want_scores=True)[3].cpu()
            # (upstream rounds the capacity up to megablocks_size where that mode is live: more than one local expert, moe_layer.py:278-300)
            crit, l_o = oracle.extract_critical(scores, k, cf, normalize_gate=norm, alignment=mega if (mega > 0 and E > 1) else 1)
            idx, loc = layer.last_routing
            assert torch.equal(idx.cpu(), torch.stack([t.to(torch.int32) for t in crit[1]])), "idx"
            assert torch.equal(loc.cpu(), torch.stack([t.to(torch.int32) for t in crit[2]])), "loc"
            assert torch.equal(layer.dispatch_count.cpu(), crit[5]), "dispatch_count"
            assert int(layer.protected_shape[1]) == crit[4] or crit[4] == 0, f"capacity {int(layer.protected_shape[1])} vs {crit[4]}"
            assert abs(float(y.l_aux) - float(l_o)) <= (1e-5 if scores.dtype == torch.float32 else 2e-2) * max(1.0, abs(float(l_o))), "l_aux"
            w1, b1, w2, b2 = weights[1:]
            enc = oracle.fast_encode(x.to(scores.dtype), crit, post).to(dtype)
            # dims that are multiples of 64 run the MFMA grouped GEMM (fp32 accumulation, one rounding of the hidden activation and of the
            # output); the others run upstream's own ATen ops, which round after every op -- each against the oracle form that states it
            mfma = dtype != torch.float32 and M % 64 == 0 and H % 64 == 0
            ffn = oracle.expert_ffn(enc, w1, b1, w2, b2, accum_fp32=mfma)
            yo = oracle.fast_decode(ffn.to(scores.dtype), crit, post).to(dtype)
            assert y.dtype == in_dtype, f"output dtype {y.dtype}, input {in_dtype}"
            if in_dtype == dtype or in_dtype == torch.float32:
                _close(y.view(T, -1).to(dtype), yo, dtype, vs_lowprec_reference=not mfma)     # (widening the result back is exact)
            else:
                _close(y.view(T, -1), yo.to(in_dtype), in_dtype, vs_lowprec_reference=True)   # fp32 experts, output narrowed to bf16
            if cf > 0 and case % 4 == 0:
                # every fourth capturable case: the forward as a HIP graph -- replays on the example and on another batch must carry the
                # eager forward's bits (dropless routing reads its capacity back to the host and refuses capture, impls/graph.py)
                from tutel_amd.impls.graph import GraphedForward
                with torch.no_grad():
                    gf = GraphedForward(layer, xd)
                    assert torch.equal(gf(xd), y), "graph replay vs eager, the captured batch"
                    x2 = torch.roll(xd, 1, 0)
                    want2 = layer(x2)
                    assert torch.equal(gf(x2), want2) and torch.equal(gf(x2), want2), "graph replay vs eager, another batch (twice)"
        except Exception as ex:  # noqa: BLE001
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300].replace("\n", " "))
            if verbose:
                print("FAIL", bad[-1], flush=True)
        if verbose and (case + 1) % 50 == 0:
            print(f"{case + 1} layer cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


def run_training_fuzz(oracle, n_cases, seed, verbose=False):
    """random TRAINING steps (forward + backward through the layer: dispatch / combine backward kernels, gate gradients, the MFMA data-gradient
    GEMMs where the dims allow, ATen elsewhere): output and ALL gradients of the 16-bit layer against the same layer in fp32 (same
    routing: fp32 gate on the same rounded inputs).  The bar is the reference's own: the same step with the experts on upstream's ATen
    op sequence (experts/ffn.py::_TRAIN_FUSED = False) is measured against fp32 too, and the kernels may not be further from fp32
    than three times that distance (or the fixed bars of test_training_forward_and_data_gradients_on_the_mfma_gemm, whichever is larger;
    over 1 000 soaked cases the ratio stayed below 2 except once, 2.2 on an fp16 gate-weight gradient that is a near-cancelling sum:
    fp16 steps with pre-scored buckets sit at 1 - 2e-2 relative on dx on EITHER path)."""
    from test_layer_gpu import make_layer
    from tutel_amd.experts import ffn
    rnd = random.Random(seed)
    bad, t0 = [], time.time()
    for case in range(n_cases):
        E = rnd.choice([1, 2, 4, 8, 8, 16, 32, 64])
        T = rnd.choice([64, 100, 256, 500, 1024, 2048])
        k = min(E, rnd.choice([1, 2, 2, 3]))
        M = rnd.choice([64, 128, 256, 256, 512, 100])
        H = rnd.choice([64, 128, 256, 384, 512, 72])
        cf = rnd.choice([1.0, 1.0, 2.0, 0.5, 1.25])
        dtype = rnd.choice([torch.bfloat16, torch.float16])
        norm, post = rnd.random() < 0.7, rnd.random() < 0.7
        if case % 7 == 3:
            cf = (0.0, -1.5)[(case // 7) % 2]     # dropless / clamped dropless: the capacity follows the largest expert load
        if cf > 0 and k * int(cf * ((T + E - 1) // E)) == 0:
            continue
        tag = f"train case {case}: T={T} M={M} H={H} E={E} k={k} cf={cf} {dtype} norm={norm} post={post}"
        fused_was = ffn._TRAIN_FUSED
        try:
            x, *weights = oracle.make_problem(T, M, H, E, dtype=dtype, seed=seed * 37 + case)

            def run(dt, fused):
                ffn._TRAIN_FUSED = fused
                layer = make_layer(M, H, E, k, cf, dt, [t.to(dt) for t in weights], gate={"fp32_gate": True}, normalize_gate=norm, is_postscore=post).train()
                xin = x.to(dt).cuda().requires_grad_(True)
                y = layer(xin)
                loss = (y.float() * torch.linspace(-1, 1, M, device="cuda")).sum() + y.l_aux.float()
                loss.backward()
                ex = layer.experts
                return [y.detach().float().cpu()] + [g.float().cpu() for g in (xin.grad, ex.batched_fc1_w.grad, ex.batched_fc1_bias.grad, ex.batched_fc2_w.grad,
                                                                             ex.batched_fc2_bias.grad, layer.gates[0].wg.weight.grad)], layer.dispatch_count.cpu()
            r_k, c_k = run(dtype, True)
            r_a, c_a = run(dtype, False)
            r_r, c_r = run(torch.float32, False)
            assert torch.equal(c_k, c_r) and torch.equal(c_a, c_r), "dispatch_count of the three runs"
            eps = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
            for name, a, u, b in zip(("y", "dx", "dfc1_w", "dfc1_bias", "dfc2_w", "dfc2_bias", "dgate_w"), r_k, r_a, r_r):
                mult = 4 if name == "y" else 8
                scale, nb = float(b.abs().max()), b.norm().clamp_min(1e-12)
                fro, fro_aten = float((a - b).norm() / nb), float((u - b).norm() / nb)
                assert fro <= max(mult * eps, 3 * fro_aten), f"{name}: relative Frobenius error {fro:.3e} (ATen path: {fro_aten:.3e})"
                mx, mx_aten = float((a - b).abs().max()), float((u - b).abs().max())
                assert mx <= max(12 * mult * eps * scale + 1e-6, 3 * mx_aten), f"{name}: max error {mx:.3e} (ATen path: {mx_aten:.3e}) at scale {scale:.3e}"
        except Exception as ex:  # noqa: BLE001
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300].replace("\n", " "))
            if verbose:
                print("FAIL", bad[-1], flush=True)
        finally:
            ffn._TRAIN_FUSED = fused_was
        if verbose and (case + 1) % 50 == 0:
            print(f"{case + 1} training cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


def run_ext_layer_fuzz(oracle, n_cases, seed, verbose=False):
    """random forwards of the cosine top-k gate (gates/cosine_top.py) over SwiGLU experts (experts/llama_ffn.py: silu fused into W_fc1's GEMM,
    the gating product into W_fc2's): logits against the oracle's restatement of the gate, the routing really used against the oracle's on the
    kernels' own scores element for element, y against the oracle's encode -> SwiGLU -> decode on that routing."""
    from test_layer_gpu import make_ext_layer
    from tutel_amd import ops
    rnd = random.Random(seed)
    bad, t0 = [], time.time()
    for case in range(n_cases):
        E = rnd.choice([1, 2, 4, 6, 8, 16, 32, 64])
        T = rnd.choice([1, 64, 100, 300, 1000, 1024, 4096])
        k = min(E, rnd.choice([1, 2, 2, 3]))
        M = rnd.choice([64, 128, 256, 512, 100])
        H = rnd.choice([64, 128, 256, 320, 512, 72])
        P = rnd.choice([16, 32, 64])
        cf = rnd.choice([1.0, 1.0, 2.0, 0.5, 1.25, 0.0])
        dtype = rnd.choice([torch.bfloat16, torch.float16, torch.float32])
        fp32_gate = rnd.random() < 0.6
        if (k * int(cf * ((T + E - 1) // E)) if cf > 0 else 1) == 0:
            continue
        tag = f"ext case {case}: T={T} M={M} H={H} E={E} P={P} k={k} cf={cf} {dtype} fp32_gate={fp32_gate}"
        try:
            tensors = oracle.make_problem_ext(T, M, H, E, P, dtype=dtype, seed=seed * 41 + case)
            x, pw, pb, sim, temp, w1, w2, w3 = tensors
            layer = make_ext_layer(M, H, E, P, k, cf, dtype, fp32_gate, tensors)
            layer._keep_routing, layer.last_logits = True, None
            xd = x.cuda()
            with torch.no_grad():
                y = layer(xd)
                logits = layer.gates[0](xd)
                scores = ops.gate_topk(logits.contiguous(), k, apply_softmax=True, want_scores=True)[3].cpu()
            want_logits = oracle.cosine_gate_logits(x, pw, pb, sim, temp, fp32_gate)
            ltol = dict(rtol=1e-5, atol=1e-6) if want_logits.dtype == torch.float32 else dict(rtol=2 ** -6, atol=2 ** -6)
            torch.testing.assert_close(logits.cpu(), want_logits, **ltol)
            crit, l_o = oracle.extract_critical(scores, k, cf)
            idx, loc = layer.last_routing
            assert torch.equal(idx.cpu(), torch.stack([t.to(torch.int32) for t in crit[1]])), "idx"
            assert torch.equal(loc.cpu(), torch.stack([t.to(torch.int32) for t in crit[2]])), "loc"
            assert torch.equal(layer.dispatch_count.cpu(), crit[5]), "dispatch_count"
            assert abs(float(y.l_aux) - float(l_o)) <= (1e-5 if scores.dtype == torch.float32 else 2e-2) * max(1.0, abs(float(l_o))), "l_aux"
            mfma = dtype != torch.float32 and M % 64 == 0 and H % 64 == 0
            enc = oracle.fast_encode(x.to(scores.dtype), crit).to(dtype)
            ffn = oracle.expert_llama_ffn(enc, w1, w2, w3, accum_fp32=mfma)
            yo = oracle.fast_decode(ffn.to(scores.dtype), crit).to(dtype)
            _close_at_scale(y.view(T, -1), yo, dtype)
        except Exception as ex:  # noqa: BLE001
            bad.append(tag + " :: " + (str(ex) or type(ex).__name__)[:300].replace("\n", " "))
            if verbose:
                print("FAIL", bad[-1], flush=True)
        if verbose and (case + 1) % 50 == 0:
            print(f"{case + 1} ext cases, {len(bad)} failed, {time.time() - t0:.0f} s", flush=True)
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [120, pytest.param(1500, marks=pytest.mark.slow)])
def test_routing_dispatch_combine_fuzz_vs_oracle(oracle, n_cases):
    bad = run_routing_fuzz(oracle, n_cases, seed=6060)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [120, pytest.param(1200, marks=pytest.mark.slow)])
def test_grouped_gemm_fuzz_vs_fp32_reference_and_across_kernels(n_cases):
    bad = run_gemm_fuzz(n_cases, seed=6061)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [30, pytest.param(600, marks=pytest.mark.slow)])
def test_layer_forward_fuzz_vs_oracle(oracle, n_cases):
    bad = run_layer_fuzz(oracle, n_cases, seed=6062)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [30, pytest.param(400, marks=pytest.mark.slow)])
def test_training_step_fuzz_vs_fp32_layer(oracle, n_cases):
    bad = run_training_fuzz(oracle, n_cases, seed=6063)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [40, pytest.param(400, marks=pytest.mark.slow)])
def test_cosine_gate_swiglu_expert_layer_fuzz_vs_oracle(oracle, n_cases):
    bad = run_ext_layer_fuzz(oracle, n_cases, seed=6064)
    assert not bad, "\n".join(bad[:20])


if __name__ == "__main__":
    from oracle import moe_oracle
    moe_oracle._lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 6060
    what = sys.argv[3] if len(sys.argv) > 3 else "routing"
    runners = {"routing": lambda: run_routing_fuzz(moe_oracle, n, sd, verbose=True), "gemm": lambda: run_gemm_fuzz(n, sd, verbose=True),
               "layer": lambda: run_layer_fuzz(moe_oracle, n, sd, verbose=True), "train": lambda: run_training_fuzz(moe_oracle, n, sd, verbose=True),
               "ext": lambda: run_ext_layer_fuzz(moe_oracle, n, sd, verbose=True)}
    assert what in runners, f"what: one of {sorted(runners)}"
    failed = runners[what]()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"r6_{what}_fuzz_{sd}.json"), "w") as f:
        json.dump(dict(source="tests/test_fuzz_gpu.py", cases=n, seed=sd, failed=failed), f, indent=1)
    print("cases", n, "failed", len(failed))
    sys.exit(1 if failed else 0)
