#!/usr/bin/env python3
"""Round 5 A/B: the split-K form of the 256 x 256 ping-pong kernel (TUTEL_OPT_GEMM_SPLITK) on the launches it is made for --
96 .. 191 tiles, half the chip -- against the grids those shapes took before (unsplit 128-block grid, 256 x 128 ring), interleaved,
alternating two weight sets (weights from HBM), plus one rank's degree-2 pipeline of an 8-way run with the split forced / off.

    python tools/r5_splitk_probe.py  -> gpurun_out/r5_splitk_probe.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench(fn, iters=40):
    for i in range(8):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    from tutel_amd import _lib, ops
    dev = torch.device("cuda", 0)
    out = {"gemm": {}, "pipeline": {}}
    g = torch.Generator(device=dev).manual_seed(3)
    for name, (E, R, N, K), dtype in (("stage 4 x 1024 x 2048 x 2048 bf16", (4, 1024, 2048, 2048), torch.bfloat16),
                                      ("stage 4 x 1024 x 4096 x 4096 bf16 (configs[3], degree 2: 256 tiles, not split)", (4, 1024, 4096, 4096), torch.bfloat16),
                                      ("6 x 1024 x 2048 x 2048 bf16 (192 tiles: not split)", (6, 1024, 2048, 2048), torch.bfloat16),
                                      ("3 x 1024 x 2048 x 2048 fp16 (96 tiles)", (3, 1024, 2048, 2048), torch.float16),
                                      ("5 x 1024 x 2048 x 4096 bf16 (160 tiles)", (5, 1024, 2048, 4096), torch.bfloat16)):
        a = torch.randn([E, R, K], device=dev, generator=g).to(dtype)
        ws = [(torch.randn([E, N, K], device=dev, generator=g) * 0.03).to(dtype) for _ in range(2)]
        b = torch.randn([E, N], device=dev, generator=g).to(dtype)
        flops = 2.0 * E * R * N * K
        res = {}
        for rep in range(2):
            for label, sk, tile in (("auto(before: unsplit choices)", 0, -1), ("split-K", 1, -1), ("unsplit 256x256 ping-pong", 0, 4), ("256x128 ring", 0, 3)):
                ops.set_option(_lib.OPT_GEMM_SPLITK, sk)
                ops.set_option(_lib.OPT_GEMM_TILE, tile)
                us = bench(lambda i: ops.expert_gemm(a, ws[i & 1], b, True, act="relu"))
                res.setdefault(label, []).append(round(us, 2))
        ops.set_option(_lib.OPT_GEMM_SPLITK, -1)
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
        out["gemm"][name] = {k: {"us": v, "tflops": round(flops / min(v) * 1e-6, 1)} for k, v in res.items()}
        print(name, out["gemm"][name], flush=True)
        del a, ws, b
    # one rank's pipeline of an 8-way run (world-size-1 IPC communicator), degree 1 / 2, split forced / off / automatic
    import torch.distributed as dist
    import bench as B
    from tutel_amd.impls import ep_native as EN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(B.free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    EN.set_transport("ipc", hosted=False)
    EN._FORCE_COMM = True
    for label, sk in (("automatic", -1), ("split forced", 1), ("split off", 0), ("automatic (again)", -1)):
        ops.set_option(_lib.OPT_GEMM_SPLITK, sk)
        t = B.rank_pipeline_probe(8, 2048, 2048, 4096, 2, torch.bfloat16, dev, iters=100)
        out["pipeline"][label] = {"degree1_ms": round(t[1], 4), "degree2_ms": round(t[2], 4)}
        print(label, out["pipeline"][label], flush=True)
    ops.set_option(_lib.OPT_GEMM_SPLITK, -1)
    EN.destroy_all()
    dist.destroy_process_group()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "r5_splitk_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
