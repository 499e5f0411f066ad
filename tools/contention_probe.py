#!/usr/bin/env python3
"""Round-3 probe (VERDICT r2 "next" 2b): the stage GEMMs of the overlapped expert-parallel pipeline with a CO-RUNNING kernel on
another stream.  On a real 8-GPU run RCCL's all-to-all kernels occupy compute units while the side stream runs the GEMMs; the
stage GEMM grids are sized to fill 256 CUs with one workgroup each, so stolen CUs could turn one wave of workgroups into two.

For each GEMM shape (one pipeline stage and the whole rank of an 8-way expert-parallel run at the headline dims; the 4096^2 rank
shapes of BASELINE configs[3]) and each kernel the library would pick / could pick:
    alone                      back-to-back launches, events around them
    + N CUs pinned             a synthetic kernel holding 16 / 32 / 64 whole CUs (160 KB LDS each) on a second stream
    + 1-rank RCCL all-to-all   the library's own communicator moving 32 MiB per call on a second stream (a device copy)
Prints one JSON object; copy into profiles/.

    python tools/contention_probe.py            (builds tools/scratch/libcu_pin.so with hipcc on first use)
"""
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tutel_amd import _lib, ops  # noqa: E402


def pin_lib():
    so = os.path.join(ROOT, "tools", "scratch", "libcu_pin.so")
    src = os.path.join(ROOT, "tools", "scratch", "cu_pin.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
    L = ctypes.CDLL(so)
    L.cu_pin.restype, L.cu_pin.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    L.cu_stamp.restype, L.cu_stamp.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]
    return L


STAMPS = None   # int64[2] on the device: wall-clock ticks (100 MHz) right before / after the timed GEMMs, on their stream
PINLIB = None


def time_gemm(fn, iters, before=None, after=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        if before:
            before()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        PINLIB.cu_stamp(STAMPS.data_ptr(), ops._stream())
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        PINLIB.cu_stamp(STAMPS.data_ptr() + 8, ops._stream())
        torch.cuda.synchronize()
        if after:
            after()
        best.append(s.elapsed_time(e) * 1e3 / iters)
    return round(min(best), 2), round(sorted(best)[1], 2)


def overlap(sink, n):
    """fraction of the last timed GEMM region during which all n pinning blocks were resident (device wall clock)"""
    st = STAMPS.cpu().tolist()
    iv = sink[:2 * n].cpu().view(n, 2)
    lo, hi = int(iv[:, 0].max()), int(iv[:, 1].min())
    if os.environ.get("PROBE_DEBUG"):
        print("debug: gemm region", (st[1] - st[0]) / 100.0, "us; pin starts (rel. to region start, us)", (int(iv[:, 0].min()) - st[0]) / 100.0, (lo - st[0]) / 100.0,
              "pin ends", (hi - st[0]) / 100.0, (int(iv[:, 1].max()) - st[0]) / 100.0, file=sys.stderr)
    return round(max(0, min(hi, st[1]) - max(lo, st[0])) / max(1, st[1] - st[0]), 3)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
    from tutel_amd.impls import ep_native
    comm = ep_native.communicator(None, dev)   # the library's own RCCL communicator (1 rank: ncclAllToAll is a device copy)
    global STAMPS, PINLIB
    P = PINLIB = pin_lib()
    STAMPS = torch.zeros([2], dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(priority=-1)   # high priority: never multiplexed onto the default stream's hardware queue (profiles/r03_stream_queues.txt)
    sink = torch.zeros([512], dtype=torch.int64, device=dev)
    a2a_src = torch.randn([16 * 1024 * 1024], device=dev).bfloat16()
    a2a_dst = torch.empty_like(a2a_src)
    g = torch.Generator(device=dev).manual_seed(1)
    out = {"note": "us per launch: [min, median] of 3 rounds; pinned CUs hold 160 KB of LDS each for the whole timed region", "shapes": {}}
    shapes = [("stage 4x1024 K=N=2048 (N=8 headline, degree 2)", 4, 1024, 2048, 2048), ("rank 8x1024 K=N=2048 (N=8 headline, degree 1)", 8, 1024, 2048, 2048),
              ("stage 4x1024 K=N=4096 (configs[3], degree 2)", 4, 1024, 4096, 4096), ("rank 8x1024 K=N=4096 (configs[3], degree 1)", 8, 1024, 4096, 4096)]
    # which stream carries what: "gemm on default" = the GEMMs on torch's default stream and the co-runner on a non-blocking side
    # stream; "gemm on side" = the product's arrangement (collectives on the caller's stream, stage GEMMs on the library's side stream)
    arrangement = os.environ.get("PROBE_GEMM_STREAM", "default")
    out["gemm_stream"] = arrangement
    gemm_stream = torch.cuda.current_stream() if arrangement == "default" else torch.cuda.Stream(priority=-1)
    co_stream = side if arrangement == "default" else torch.cuda.default_stream()
    torch.cuda.set_stream(gemm_stream)
    if os.environ.get("PROBE_DEBUG"):
        shapes = shapes[:1]
    for name, El, R, K, N in shapes:
        a = torch.randn([El, R, K], device=dev, generator=g).bfloat16()
        w = (torch.randn([El, N, K], device=dev, generator=g) * 0.03).bfloat16()
        b = torch.randn([El, N], device=dev, generator=g).bfloat16()
        flops = 2.0 * El * R * K * N
        iters = max(4, int(600e-6 / (flops / 900e12)))   # ~0.6 ms of GEMMs per timed region
        res = {}
        for kname, opt in (("auto", -1), ("256x256 ping-pong", 4), ("256x128 ring", 3)):
            ops.set_option(_lib.OPT_GEMM_TILE, opt)
            fn = lambda: ops.expert_gemm(a, w, b, True, act="relu")
            r = {"alone": time_gemm(fn, iters)}
            for n in (16, 32, 64):
                def before(n=n):
                    with torch.cuda.stream(co_stream):
                        assert P.cu_pin(n, 3000.0, sink.data_ptr(), ops._stream()) == 0
                    torch.cuda._sleep(200000)   # let the pinning blocks become resident before the GEMMs are queued
                r[f"{n} CUs pinned"] = time_gemm(fn, iters, before=before, after=torch.cuda.synchronize)
                r[f"{n} CUs pinned: overlap of the pin with the timed region"] = overlap(sink, n)

            def before_a2a():
                with torch.cuda.stream(co_stream):
                    for _ in range(40):
                        comm.all_to_all(a2a_dst, a2a_src)
            r["1-rank RCCL all-to-all (32 MiB copies) co-running"] = time_gemm(fn, iters, before=before_a2a, after=torch.cuda.synchronize)
            r["tflops_alone"] = round(flops / r["alone"][0] * 1e-6, 1)
            res[kname] = r
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
        out["shapes"][name] = res
    print(json.dumps(out, indent=1))
    ep_native.destroy_all()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
