"""Dev tool: host enqueue time per forward (loop without synchronisation until the end, host clock stopped
before the final synchronize) vs GPU time, for the plain single-rank path and for the overlapped
expert-parallel routine forced onto one rank through a real RCCL group (the code path of N > 1)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(step):
    """host cost per step from short bursts on an EMPTY queue (a long back-to-back loop fills the HIP queue and then
    measures the GPU, not the host); wall per step from 200 steps back to back"""
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
    wall = (time.perf_counter() - t0) / n
    return 1e3 * sorted(host)[len(host) // 2], 1e3 * wall


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from tutel import moe
    from tutel_amd.impls import moe_layer as ML, overlap as OV, ep_native as EN
    T, M, H, E, k = 4096, 2048, 2048, int(os.environ.get("E", 8)), 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}).cuda().eval()
    torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device="cuda").bfloat16()
    for name, native, force, rccl, degree in (
            ("python: plain", False, False, False, 1), ("python: overlap x2 (forced, RCCL)", False, True, True, 2),
            ("native: plain (fused encode)", True, False, False, 1), ("native: degree 1 (own RCCL comm)", True, False, True, 1),
            ("native: degree 2 (own RCCL comm)", True, False, True, 2), ("native: degree 4 (own RCCL comm)", True, False, True, 4)):
        EN.ENABLED, EN._FORCE_COMM = native, native and rccl
        ML._FORCE_OVERLAP = force
        OV._FORCE_RCCL = rccl and not native
        with torch.no_grad():
            host, wall = measure(lambda: layer(x, a2a_ffn_overlap_degree=degree))
        print(f"{name:36s} host enqueue {host:.3f} ms/step (bursts of 8 on an idle queue), wall {wall:.3f} ms/step (200 back to back)", flush=True)
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
    ML._FORCE_OVERLAP = OV._FORCE_RCCL = False
    EN.ENABLED = EN._FORCE_COMM = True
    with torch.no_grad():
        ref = layer(x, a2a_ffn_overlap_degree=2)
    try:
        gf = GraphedForward(layer, x, a2a_ffn_overlap_degree=2)
        y = gf(x)
        torch.cuda.synchronize()
        print("graph == eager:", bool(torch.equal(y, ref)), flush=True)
        host, wall = measure(lambda: gf(x))
        print(f"{'native: degree 2, HIP-graph replay':36s} host enqueue {host:.3f} ms/step, wall {wall:.3f} ms/step", flush=True)
    except Exception as ex:  # noqa: BLE001
        print("graph capture failed:", type(ex).__name__, str(ex)[:300], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
