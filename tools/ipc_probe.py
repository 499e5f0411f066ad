"""Dev tool: the expert-parallel forward over the IPC transport with W rank processes sharing cuda:0 -- host enqueue time per
forward (bursts of 8 on an idle queue, host clock stopped before the synchronize), wall per forward (200 back to back) and the
same replayed from a HIP graph, per overlap degree; next to the host-staged exchange for reference.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ipc_probe.py

Per-rank shape: the EP-8 rank problem at the headline dims (8 local experts x 1024 rows, M = H = 2048) reproduced with W ranks.
With the ranks on ONE GPU the wall time is W ranks' device work on one device; the host number is what carries over."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(step, sync):
    for _ in range(20):
        step()
    sync()
    host = []
    for _ in range(10):
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        host.append((time.perf_counter() - t0) / 8)
        sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    wall = (time.perf_counter() - t0) / n
    return 1e3 * sorted(host)[len(host) // 2], 1e3 * wall


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tutel import moe
    from tutel_amd.impls import ep_native as EN
    from tutel_amd.impls.graph import GraphedForward
    M = H = int(os.environ.get("M", 2048))
    E_loc, k = int(os.environ.get("E_LOC", 8)), 2
    T = int(os.environ.get("T", 4096))
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}, seeds=(1, rank + 1, 1)).cuda().eval()
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(rank)
    x = torch.randn([T, M], device="cuda").bfloat16()

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
    out = {"world": world, "T": T, "M": M, "E_loc": E_loc, "rows_per_expert": world * k * ((T + E_loc * world - 1) // (E_loc * world)), "modes": {}}
    ref = {}
    for transport in ("ipc", "hosted"):
        EN.HOSTED, EN.TRANSPORT = transport == "hosted", ("ipc" if transport == "ipc" else "rccl")
        EN._comms.clear()
        layer.__dict__.pop("_ep_workspaces", None)
        for degree in (1, 2, 4):
            with torch.no_grad():
                y = layer(x, a2a_ffn_overlap_degree=degree)
                sync()
                if degree in ref:
                    assert torch.equal(y, ref[degree]), (transport, degree)
                ref.setdefault(degree, y.clone())
                host, wall = measure(lambda: layer(x, a2a_ffn_overlap_degree=degree), sync)
            out["modes"][f"{transport} eager degree {degree}"] = {"host_ms": round(host, 4), "wall_ms": round(wall, 4)}
            if transport == "ipc":
                gf = GraphedForward(layer, x, a2a_ffn_overlap_degree=degree)
                assert torch.equal(gf(x), ref[degree])
                host, wall = measure(lambda: gf(x), sync)
                out["modes"][f"ipc graph degree {degree}"] = {"host_ms": round(host, 4), "wall_ms": round(wall, 4)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
