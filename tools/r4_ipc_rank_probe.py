#!/usr/bin/env python3
"""Dev tool (round 4): ONE rank's pipeline of an 8-way expert-parallel run over the IPC transport, alone on the GPU -- a world-size-1
IPC communicator (the peer is the rank itself: fast_encode and the fc2 epilogue store into its own segment through the peer table,
flags are signalled and awaited exactly as between ranks), 8 local experts x 1024 rows, M = H = 2048 (T = 4096, E = 8, top-2).
Per overlap degree and stage-grid choice: wall per forward eager and HIP-graph replayed, host enqueue time, and -- under
`rocprofv3 --kernel-trace` with TRACE=1 -- nothing but a few forwards for the timeline.

    python tools/r4_ipc_rank_probe.py            -> gpurun_out/r4_ipc_rank_probe.json"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(step):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    host = []
    for _ in range(10):
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        host.append((time.perf_counter() - t0) / 8)
        torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return round(1e3 * sorted(host)[len(host) // 2], 4), round(1e3 * (time.perf_counter() - t0) / n, 4)


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29561"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.cuda.set_device(0)
    from tutel import moe
    from tutel_amd import _lib, ops
    from tutel_amd.impls import ep_native as EN
    from tutel_amd.impls.graph import GraphedForward
    EN.HOSTED, EN.TRANSPORT, EN._FORCE_COMM = False, "ipc", True
    T, M, H, E, k = 4096, int(os.environ.get("M", 2048)), int(os.environ.get("M", 2048)), 8, 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}).cuda().eval()
    torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device="cuda").bfloat16()
    if os.environ.get("TRACE"):
        ops.set_option(_lib.OPT_EP_STAGE_GRID, int(os.environ.get("STAGE_GRID", "-1")))
        with torch.no_grad():
            for _ in range(15):
                layer(x, a2a_ffn_overlap_degree=int(os.environ.get("DEGREE", "2")))
        torch.cuda.synchronize()
        dist.destroy_process_group()
        return
    out, ref = {"shape": dict(T=T, M=M, H=H, E_loc=E, rows_per_expert=2 * T // E), "modes": {}}, None
    for degree in (1, 2, 4):
        for grid in ((-1,) if degree == 1 else (1, 0)):
            ops.set_option(_lib.OPT_EP_STAGE_GRID, grid)
            layer.__dict__.pop("_ep_workspaces", None)
            with torch.no_grad():
                y = layer(x, a2a_ffn_overlap_degree=degree)
                torch.cuda.synchronize()
                comm = EN.communicator(layer.group, x.device)
                assert comm is not None and comm.ipc and comm.world == 1
                ref = y.clone() if ref is None else ref
                assert torch.equal(y, ref), (degree, grid)     # the degrees keep the rows per launch here: same bits
                host, wall = measure(lambda: layer(x, a2a_ffn_overlap_degree=degree))
                gf = GraphedForward(layer, x, a2a_ffn_overlap_degree=degree)
                assert torch.equal(gf(x), ref)
                ghost, gwall = measure(lambda: gf(x))
            name = f"degree {degree}" + ("" if degree == 1 else (", half-chip stage grids" if grid else ", full stage grids"))
            out["modes"][name] = {"eager_host_ms": host, "eager_wall_ms": wall, "graph_host_ms": ghost, "graph_wall_ms": gwall}
            print(name, out["modes"][name], flush=True)
            del gf
    ops.set_option(_lib.OPT_EP_STAGE_GRID, -1)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "r4_ipc_rank_probe.json"), "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
