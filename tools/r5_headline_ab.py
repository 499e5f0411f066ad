"""Dev tool (round 5): A/B at the headline shape (BASELINE configs[1]) on one MI355X, random operands, interleaved arms.

  1. the gate: library projection (F.linear -> hipBLASLt) + top-k kernel   vs   split-K MFMA projection (csrc/gate_proj.hip) +
     the top-k kernel adding the partial sums;
  2. the whole forward (HIP-graph replay) with the projection outside (F.linear) / inside the native call;
  3. the whole forward with fc1's weights being warmed into the memory-side cache from a second stream while the routing kernels run
     (tutel_amd_cache_warm): contiguous first N MB, and the first-round experts' weights (the XCD-aware work order starts experts
     8 x + {0..3} on XCD x).

    python tools/r5_headline_ab.py [sections]        -> gpurun_out/r5_headline_ab.json      sections: any of g f w (default all)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import _lib, ops  # noqa: E402


def graph_time(fn, n=20, reps=9):
    """us per call of fn: n calls captured into one HIP graph, replayed between two events; (median, min) of reps"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / n)
    torch.cuda.current_stream().wait_stream(s)
    return round(sorted(ts)[len(ts) // 2], 2), round(min(ts), 2)


def main():
    sections = sys.argv[1] if len(sys.argv) > 1 else "gfslwt"
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    C = k * (T // E)
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(0)
    x = torch.randn([T, M], generator=g).to(dt).to(dev)
    wg = (torch.randn([E, M], generator=g) / M ** 0.5).to(dt).to(dev)
    res = {"shape": dict(T=T, M=M, H=H, E=E, k=k, C=C), "gate": {}, "forward": {}, "warm": {}}

    if "g" in sections:
        ws = ops.routing_workspace(T, E, k, dev)
        smap = torch.empty([E * C], dtype=torch.int32, device=dev)
        S = ops.gate_proj_splits(T, M, E, dt)
        part = torch.empty([S, T, E], dtype=torch.float32, device=dev)

        def lib_gate():
            return ops.gate_topk(torch.nn.functional.linear(x, wg), k, apply_softmax=True, ws=ws, clear=smap)

        def native_gate():
            ops.gate_proj(x, wg, partials=part)
            return ops.gate_topk_partials(part, dt, k, ws=ws, clear=smap)
        lg = torch.nn.functional.linear(x, wg)
        for rnd in range(2):
            for name, fn in (("library projection + top-k", lib_gate), ("library projection alone", lambda: torch.nn.functional.linear(x, wg)),
                             ("top-k on logits alone", lambda: ops.gate_topk(lg, k, apply_softmax=True, ws=ws, clear=smap)),
                             ("split-K projection + top-k on partials", native_gate), ("split-K projection alone", lambda: ops.gate_proj(x, wg, partials=part)),
                             ("top-k on partials alone", lambda: ops.gate_topk_partials(part, dt, k, ws=ws, clear=smap))):
                res["gate"].setdefault(name, []).append(graph_time(fn))
        res["gate"]["splits"] = S
        i1, g1 = lib_gate()[:2]
        i2, g2 = native_gate()[:2]
        torch.cuda.synchronize()
        res["gate"]["assignments_differing_from_the_library_logits"] = int((i1 != i2).sum())
        print(json.dumps(res["gate"]), flush=True)

    if any(c in sections for c in "fwtsl"):
        from tutel import moe
        from tutel_amd.impls import moe_layer as ML
        torch.set_default_dtype(dt)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)}).to(dev).eval()
        torch.set_default_dtype(torch.float32)

        def fwd():
            with torch.no_grad():
                return layer(x)

    if "f" in sections:
        for rnd in range(3):
            for native in (False, True):
                ML._NATIVE_GATE = native
                res["forward"].setdefault("gate projection %s" % ("inside the native call (split-K)" if native else "F.linear (library)"), []).append(graph_time(fwd, n=10))
        ML._NATIVE_GATE = True
        print(json.dumps(res["forward"]), flush=True)

    if "s" in sections:
        # fc1's fused fast_encode: slot-map entries through the scalar cache, waited for after the first weight pieces went out (1)
        # vs four vector loads in front of the first DMA (0, rounds 1-4)
        res["scalar_gather"] = {}
        ops.set_option(_lib.OPT_FUSED_LOCATION, 0)   # (with the fused location on, the ring kernel takes its rows from the in-kernel scan)
        for rnd in range(3):
            for mode in (0, 1):
                ops.set_option(_lib.OPT_GEMM_GATHER, mode)
                res["scalar_gather"].setdefault("forward, slot map via %s" % ("scalar loads after the weight issue" if mode else "vector loads first"), []).append(graph_time(fwd, n=10))
        ops.set_option(_lib.OPT_GEMM_GATHER, -1)
        import ctypes
        for rnd in range(2):
            for mode in (0, 1):
                ops.set_option(_lib.OPT_GEMM_GATHER, mode)
                _lib.lib().tutel_amd_stage_timing(2)
                for _ in range(60):
                    fwd()
                torch.cuda.synchronize()
                tot, cnt = (ctypes.c_double * 16)(), (ctypes.c_int * 16)()
                _lib.lib().tutel_amd_stage_report(tot, cnt, 16)
                _lib.lib().tutel_amd_stage_timing(0)
                res["scalar_gather"].setdefault("eager fc1 / fc2 us, mode %d" % mode, []).append([round(tot[3] / max(cnt[3], 1), 2), round(tot[4] / max(cnt[4], 1), 2)])
        ops.set_option(_lib.OPT_GEMM_GATHER, -1)
        ops.set_option(_lib.OPT_FUSED_LOCATION, -1)
        print(json.dumps(res["scalar_gather"], indent=0), flush=True)

    if "l" in sections:
        # locations inside the first expert GEMM (no location launch) vs the location kernel
        res["fused_location"] = {}
        for rnd in range(3):
            for mode in (0, -1):
                ops.set_option(_lib.OPT_FUSED_LOCATION, mode)
                layer.__dict__.pop("_ep_workspaces", None)
                res["fused_location"].setdefault("forward, %s" % ("locations inside fc1" if mode else "location kernel"), []).append(graph_time(fwd, n=10))
        import ctypes
        for rnd in range(2):
            for mode in (0, -1):
                ops.set_option(_lib.OPT_FUSED_LOCATION, mode)
                layer.__dict__.pop("_ep_workspaces", None)
                fwd()
                ops.stage_timing(1)
                for _ in range(60):
                    fwd()
                torch.cuda.synchronize()
                rep = ops.stage_report()
                ops.stage_timing(0)
                res["fused_location"].setdefault("eager us per launch, %s" % ("fused" if mode else "unfused"), []).append(
                    {name: round(tot / max(cnt, 1), 2) for name, (tot, cnt) in rep.items() if cnt})
        ops.set_option(_lib.OPT_FUSED_LOCATION, -1)
        print(json.dumps(res["fused_location"], indent=0), flush=True)

    if "w" in sections:
        w1 = layer.experts.fused_params(dt)[0]          # [E, H, M]: fc1's weights as the GEMM streams them
        per_e = H * M * 2
        side = torch.cuda.Stream(priority=-1)

        def warmed(chunks, blocks):
            def f():
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.cache_warm(w1, *chunks, blocks=blocks)
                y = fwd()
                cur.wait_stream(side)
                return y
            return f
        arms = [("no warm-up", None), ("fork / join only (one 16-byte chunk)", ((16, 1, 0), 1))]
        for mb in (32, 64, 96, 128, 192):
            arms.append((f"first {mb} MB, contiguous", ((mb << 20, 1, 0), 256)))
        for epx in (1, 2, 3, 4):   # experts per XCD group: the first-round experts 8 x + {0 .. epx - 1}
            arms.append((f"first-round experts, {epx} per XCD ({8 * epx * per_e >> 20} MB)", ((epx * per_e, 8, 8 * per_e), 256)))
        arms.append(("first-round experts, 2 per XCD, 512 blocks", ((2 * per_e, 8, 8 * per_e), 512)))
        arms.append(("first-round experts, 2 per XCD, 128 blocks", ((2 * per_e, 8, 8 * per_e), 128)))
        for native in (True, False):
            ML._NATIVE_GATE = native
            for rnd in range(2):
                for name, spec in arms:
                    fn = fwd if spec is None else warmed(*spec)
                    res["warm"].setdefault(("native gate | " if native else "library gate | ") + name, []).append(graph_time(fn, n=10))
            print(json.dumps({k_: v for k_, v in res["warm"].items() if k_.startswith("native" if native else "library")}, indent=0), flush=True)
        ML._NATIVE_GATE = True

    if "t" in sections:
        # what the warm-up does to fc1 / fc2 THEMSELVES: eager forwards, the library's own event pairs around the two GEMM launches
        import ctypes
        w1 = layer.experts.fused_params(dt)[0]
        per_e = H * M * 2
        side = torch.cuda.Stream(priority=-1)
        res["gemm_us_with_warm_up"] = {}
        res["warm_kernel_alone_us"] = {f"{mb} MB": graph_time(lambda: ops.cache_warm(w1, mb << 20, 1, 0, blocks=256)) for mb in (32, 64, 128)}
        print(json.dumps(res["warm_kernel_alone_us"]), flush=True)
        for rnd in range(2):
            for name, spec in (("no warm-up", None), ("first-round experts, 1 per XCD (64 MB)", (per_e, 8, 8 * per_e)),
                               ("first-round experts, 2 per XCD (128 MB)", (2 * per_e, 8, 8 * per_e)), ("first 96 MB, contiguous", (96 << 20, 1, 0))):
                _lib.lib().tutel_amd_stage_timing(2)
                for _ in range(60):
                    cur = torch.cuda.current_stream()
                    if spec is not None:
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            ops.cache_warm(w1, *spec, blocks=256)
                    fwd()
                    if spec is not None:
                        cur.wait_stream(side)
                torch.cuda.synchronize()
                tot, cnt = (ctypes.c_double * 16)(), (ctypes.c_int * 16)()
                _lib.lib().tutel_amd_stage_report(tot, cnt, 16)
                _lib.lib().tutel_amd_stage_timing(0)
                res["gemm_us_with_warm_up"].setdefault(name, []).append({"fc1": round(tot[3] / max(cnt[3], 1), 2), "fc2": round(tot[4] / max(cnt[4], 1), 2)})
        print(json.dumps(res["gemm_us_with_warm_up"], indent=0), flush=True)

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r5_headline_ab.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
