"""Dev tool: host enqueue time per forward (loop without synchronisation until the end, host clock stopped
before the final synchronize) vs GPU time, for the plain single-rank path and for the overlapped
expert-parallel routine forced onto one rank through a real RCCL group (the code path of N > 1)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from tutel import moe
    from tutel_amd.impls import moe_layer as ML, overlap as OV
    T, M, H, E, k = 4096, 2048, 2048, int(os.environ.get("E", 8)), 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}).cuda().eval()
    torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device="cuda").bfloat16()
    for name, force, rccl, degree in (("plain", False, False, 1), ("overlap x2 (forced, copies)", True, False, 2),
                                      ("overlap x2 (forced, RCCL)", True, True, 2), ("overlap x4 (forced, RCCL)", True, True, 4)):
        ML._FORCE_OVERLAP = force
        OV._FORCE_RCCL = rccl
        with torch.no_grad():
            for _ in range(20):
                layer(x, a2a_ffn_overlap_degree=degree)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                layer(x, a2a_ffn_overlap_degree=degree)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{name:28s} host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
        if os.environ.get("PROFILE") and degree == 2:
            import cProfile, pstats
            pr = cProfile.Profile()
            with torch.no_grad():
                pr.enable()
                for _ in range(100):
                    layer(x, a2a_ffn_overlap_degree=degree)
                pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    # the same forward replayed from a HIP graph (RCCL collectives and both streams captured)
    from tutel_amd.impls.graph import GraphedForward
    ML._FORCE_OVERLAP = OV._FORCE_RCCL = True
    with torch.no_grad():
        ref = layer(x, a2a_ffn_overlap_degree=2)
    try:
        gf = GraphedForward(layer, x, a2a_ffn_overlap_degree=2)
        y = gf(x)
        torch.cuda.synchronize()
        print("graph == eager:", bool(torch.equal(y, ref)), flush=True)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            gf(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{'overlap x2 graph replay':28s} host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
    except Exception as ex:  # noqa: BLE001
        print("graph capture failed:", type(ex).__name__, str(ex)[:300], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
