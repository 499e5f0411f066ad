#!/usr/bin/env python3
"""Dev tool (round 4): is ONE grouped-GEMM launch over all local experts slower than TWO concurrent launches over half of them each
(two streams of different priority, half-chip 256 x 256 grids side by side)?  The degree-2 pipeline's stage GEMMs suggested so
(profiles/r04_ipc_rank_pipeline_probe.json).  fc1 + fc2 pairs on random operands, graph-replayed (no host gaps), us per pair.

    python tools/r4_split_gemm_probe.py   -> gpurun_out/r4_split_gemm_probe.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import _lib, ops  # noqa: E402


def graph_time(fn, n=10, reps=7):
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
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    side = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    out = {}
    for name, E, R, M, H, dt in (("EP-8 rank, headline dims", 8, 1024, 2048, 2048, torch.bfloat16), ("configs[3] rank", 8, 1024, 4096, 4096, torch.bfloat16),
                                 ("EP-4 rank", 16, 512, 2048, 2048, torch.bfloat16), ("EP-2 rank", 32, 256, 2048, 2048, torch.bfloat16)):
        a = torch.randn([E, R, M], generator=g).to(dt).to(dev)
        w1 = (torch.randn([E, H, M], generator=g) / M ** 0.5).to(dt).to(dev)
        w2 = (torch.randn([E, M, H], generator=g) / H ** 0.5).to(dt).to(dev)
        b1 = torch.randn([E, H], generator=g).to(dt).to(dev)
        b2 = torch.randn([E, M], generator=g).to(dt).to(dev)
        hid = torch.empty([E, R, H], dtype=dt, device=dev)
        y = torch.empty([E, R, M], dtype=dt, device=dev)
        h = E // 2

        def whole():
            ops.expert_gemm(a, w1, b1, True, act="relu", out=hid, d_layout=(R * H, 0, R, H))
            ops.expert_gemm(hid, w2, b2, True, out=y, d_layout=(R * M, 0, R, M))

        def halves(hint):
            cur = torch.cuda.current_stream()
            for i, st in enumerate(side):
                st.wait_stream(cur)
                sl = slice(i * h, (i + 1) * h)
                with torch.cuda.stream(st):
                    if hint:
                        ops.set_option(_lib.OPT_GEMM_TILE, 4)   # the 256 x 256 ping-pong kernel on its half-chip grid
                    ops.expert_gemm(a[sl], w1[sl], b1[sl], True, act="relu", out=hid[sl], d_layout=(R * H, 0, R, H))
                    ops.expert_gemm(hid[sl], w2[sl], b2[sl], True, out=y[sl], d_layout=(R * M, 0, R, M))
                    ops.set_option(_lib.OPT_GEMM_TILE, -1)
            for st in side:
                cur.wait_stream(st)
        whole()
        ref = y.clone()
        halves(True)
        torch.cuda.synchronize()
        same = bool(torch.equal(ref, y))
        flops = 2 * 2.0 * E * R * M * H
        r = {}
        for rep in range(2):
            for key, fn in (("one launch per GEMM", whole), ("two half launches, auto tile", lambda: halves(False)), ("two half launches, 256x256 ping-pong", lambda: halves(True))):
                med, best = graph_time(fn)
                r.setdefault(key, []).append({"pair_us": med, "min": best, "TFLOPs": round(flops / med * 1e-6, 1)})
        r["bit_identical"] = same
        out[f"{name}: {E} x {R} rows, {M} x {H}"] = r
        print(name, json.dumps(r), flush=True)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "r4_split_gemm_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
