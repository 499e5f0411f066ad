#!/usr/bin/env python3
"""Round-3 A/B: the overlapped expert-parallel pipeline at the per-rank shape of the metric's 8-GPU point (8 local experts x 1024
rows, M = H = 2048, bf16), forced onto one rank through the library's REAL 1-rank RCCL communicator (ncclAllToAll = a device copy,
so the collectives occupy the GPU like a co-running kernel would), replayed from a HIP graph:
    degree 1 | degree 2 / 4 with every stage's GEMMs on ONE side stream | ... alternating between TWO side streams
Bit equality with degree 1 asserted.  Prints one JSON object (us per forward, rounds interleaved)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from tutel import moe
    from tutel_amd import _lib, ops
    from tutel_amd.impls import ep_native
    from tutel_amd.impls.graph import GraphedForward
    T, M, H, E, k = 4096, int(os.environ.get("M", 2048)), int(os.environ.get("H", 2048)), 8, 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}).cuda().eval()
    torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device=dev).bfloat16()
    ep_native._FORCE_COMM = True
    variants = [("degree 1", 1, -1), ("degree 2, one side stream", 2, 1), ("degree 2, two side streams", 2, 2),
                ("degree 4, one side stream", 4, 1), ("degree 4, two side streams", 4, 2)]
    if os.environ.get("ONLY"):   # one variant per process: several captured graphs on ONE communicator were seen to hang at replay
        variants = [variants[0], variants[int(os.environ["ONLY"])]]
    graphs, ref = {}, None
    eager = {n: [] for n, _, _ in variants}
    with torch.no_grad():
        for rnd in range(2):     # eager first (GPU-bound here: ~0.3 ms of device work against ~0.16 ms of host enqueue per forward)
            for name, degree, streams in variants:
                ops.set_option(_lib.OPT_EP_STREAMS, streams)
                for _ in range(10):
                    y = layer(x, a2a_ffn_overlap_degree=degree)
                if ref is None:
                    ref = y.clone()
                assert torch.equal(y, ref), name
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(100):
                    layer(x, a2a_ffn_overlap_degree=degree)
                b.record()
                torch.cuda.synchronize()
                eager[name].append(round(a.elapsed_time(b) * 1e3 / 100, 2))
                print("eager", name, eager[name], file=sys.stderr, flush=True)
        print(json.dumps({"eager_us_per_forward": eager}), flush=True)
        for name, degree, streams in (variants[1:] if os.environ.get("ONLY") else variants):
            ops.set_option(_lib.OPT_EP_STREAMS, streams)
            print("capturing", name, file=sys.stderr, flush=True)
            g = GraphedForward(layer, x, a2a_ffn_overlap_degree=degree)
            y = g(g.static_in).clone()
            torch.cuda.synchronize()
            assert torch.equal(y, ref), name
            graphs[name] = g
    res = {n: [] for n in graphs}
    for _ in range(3):
        for name, g in graphs.items():
            print("replaying", name, file=sys.stderr, flush=True)
            for _ in range(10):
                g(g.static_in)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(200):
                g(g.static_in)
            b.record()
            torch.cuda.synchronize()
            res[name].append(round(a.elapsed_time(b) * 1e3 / 200, 2))
    ops.set_option(_lib.OPT_EP_STREAMS, -1)
    print(json.dumps({"shape": f"8 local experts x 1024 rows, M = {M}, H = {H}, bf16, 1-rank RCCL communicator", "us_per_forward": res}))
    ep_native.destroy_all()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
