#!/usr/bin/env python3
"""Round-3 A/B at the headline shape (BASELINE configs[1]) on one MI355X: the forward replayed from a HIP graph for every
combination of the fused routing kernel (TUTEL_OPT_ROUTING) and the fast_decode launch shape (TUTEL_OPT_DECODE), steady state
(`--reps` replays between two events, several rounds interleaved so that clock / temperature drift hits every variant alike),
plus the per-kernel averages of an eager pass with events around every launch.  Prints one JSON object.

    python tools/r3_headline_ab.py [--reps 300] [--rounds 3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_layer  # noqa: E402
from tutel_amd import _lib, ops  # noqa: E402
from tutel_amd.impls.graph import GraphedForward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    layer = build_layer(M, H, E, k, 0, 1, torch.bfloat16, False).to(dev).eval()
    torch.manual_seed(0)
    x = torch.randn([16, T // 16, M]).to(torch.bfloat16).to(dev)
    combos = [(r, d) for r in (0, 1) for d in (0, 1, 2, 3)]
    graphs, ref = {}, None
    with torch.no_grad():
        for _ in range(50):
            layer(x)
        for r, d in combos:
            ops.set_option(_lib.OPT_ROUTING, r)
            ops.set_option(_lib.OPT_DECODE, d)
            g = GraphedForward(layer, x)
            y = g(g.static_in).clone()
            if ref is None:
                ref = y
            assert torch.equal(y, ref), (r, d)   # every variant returns the same bits
            graphs[(r, d)] = g
    res = {f"routing{r}_decode{d}": [] for r, d in combos}
    for _ in range(args.rounds):
        for (r, d), g in graphs.items():
            for _ in range(20):
                g(g.static_in)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(args.reps):
                g(g.static_in)
            b.record()
            torch.cuda.synchronize()
            res[f"routing{r}_decode{d}"].append(round(a.elapsed_time(b) * 1e3 / args.reps, 2))
    out = {"us_per_forward_graph_replay": res, "best": min(res, key=lambda n: min(res[n]))}
    # per-kernel averages, eager, events around every launch (each event record drains the queue: durations, not the step)
    stages = {}
    with torch.no_grad():
        for r, d in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3)):
            ops.set_option(_lib.OPT_ROUTING, r)
            ops.set_option(_lib.OPT_DECODE, d)
            for _ in range(5):
                layer(x)
            ops.stage_timing(1)
            for _ in range(40):
                layer(x)
            torch.cuda.synchronize()
            ops.stage_timing(0)
            rep = ops.stage_report()
            stages[f"routing{r}_decode{d}"] = {n: round(t / max(c, 1), 2) for n, (t, c) in rep.items() if c}
    out["avg_us_per_launch_eager_events"] = stages
    ops.set_option(_lib.OPT_ROUTING, -1)
    ops.set_option(_lib.OPT_DECODE, -1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
